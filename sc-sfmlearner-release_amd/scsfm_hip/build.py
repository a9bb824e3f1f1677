"""Build libscsfm_hip.so (gfx950) in-tree with hipcc.

    python -m scsfm_hip.build        (from sc-sfmlearner-release_amd/)

The shared object is plain HIP + a C ABI (include/scsfm_hip.h); it does not link against torch.
It is written next to this file so that it travels with the source tree to the GPU box.

Safe for N processes at once (torchrun: every rank calls ``_lib.get()`` lazily, and ``*.so`` is git-ignored, so a fresh
clone on an 8-GPU node has no library): the build runs under an exclusive ``flock`` on ``libscsfm_hip.so.lock``, the
compiler writes to a name unique to the process and the result is moved into place atomically; a process that gets the
lock after another one has built finds a binary whose compiled-in source id matches and does not compile again.
"""
from __future__ import annotations

import contextlib
import fcntl
import glob
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIB = os.path.join(HERE, "libscsfm_hip.so")
LOCK = LIB + ".lock"
LOG = LIB + ".buildlog"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-munsafe-fp-atomics",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         # the SLP vectoriser pairs unrelated scalar fp32 chains into v_pk_* with 4 v_mov per packed op (+5 % VALU
         # in the tiled pass); the 2-wide math that pays is written with vector types in csrc/scsfm_ssim.h
         "-fno-slp-vectorize"]
# the id a binary carries is embedded behind this marker, so that it can be read without loading the library
ID_MARKER = b"scsfm-source-id:"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + \
        [os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "scsfm_hip.h")]


def source_id(extra=()):
    """First 16 hex digits of the sha256 over everything the library is built from: every source file (names and
    contents, sorted) and the compiler flags / target architecture -- including any ``extra`` flags (-D tuning knobs)
    of a non-default build, so that a tuning variant can never carry the default library's id (bench.py ties PMC
    counters to a library by this id).  It is compiled into the binary (scsfm_source_id) so that a loaded .so can be
    tied to the sources next to it."""
    h = hashlib.sha256()
    files = deps()
    if any("SCSFM_WITH_MARCH" in e or "variants" in e for e in extra):
        # tuning builds compile the experimental kernels of variants/src/ in: an edit there must change the id too (round-5
        # advisor finding: PMC counters could be attributed to a stale variant binary)
        vsrc = os.path.join(os.path.dirname(os.path.dirname(HERE)), "variants", "src")
        files = files + sorted(p for p in glob.glob(os.path.join(vsrc, "*")) if os.path.isfile(p))
    for path in files:
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    h.update(" ".join(FLAGS).encode())
    if extra:
        h.update(b"\0extra:" + " ".join(extra).encode())
    return h.hexdigest()[:16]


def binary_source_id(path=LIB):
    """The source id compiled into the shared object at ``path``, read from the file (no dlopen: a stale or foreign
    binary may lack symbols the loader insists on).  None if there is no such file or it carries no id."""
    try:
        blob = open(path, "rb").read()
    except OSError:
        return None
    i = blob.find(ID_MARKER)
    if i < 0:
        return None
    j = i + len(ID_MARKER)
    return blob[j:j + 16].decode("ascii", "replace")


def is_stale():
    """True when the in-tree library is missing or was built from other sources / flags than the tree's."""
    return binary_source_id(LIB) != source_id()


@contextlib.contextmanager
def _build_lock():
    fd = os.open(LOCK, os.O_CREAT | os.O_RDWR, 0o644)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        yield
    finally:
        try:
            fcntl.flock(fd, fcntl.LOCK_UN)
        finally:
            os.close(fd)


def build(force=False, verbose=True, extra=()):
    """Compile every .hip file under csrc/ into one shared object.  Raises on failure.  Returns the library's path.
    Without ``force`` nothing is compiled when the binary in place already carries the tree's source id -- also when
    that became true while this process was waiting for the lock (N ranks asking at once: one compiles).  ``extra``:
    further compiler flags (tuning knobs); the binary then carries source_id(extra), which differs from the tree's
    default id, so the loader treats it as stale for the default configuration and rebuilds on the next plain get()."""
    extra = tuple(extra)
    want = source_id(extra)
    if not force and binary_source_id(LIB) == want:
        return LIB
    with _build_lock():
        # (another process may have built while this one waited for the lock)
        if not force and binary_source_id(LIB) == want:
            return LIB
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            raise RuntimeError("hipcc not found: libscsfm_hip.so cannot be built on this machine")
        fd, tmp = tempfile.mkstemp(prefix="libscsfm_hip.", suffix=f".{os.getpid()}.tmp", dir=HERE)
        os.close(fd)
        try:
            cmd = [hipcc, *FLAGS, f'-DSCSFM_SOURCE_ID="{want}"', *extra, "-o", tmp, *sources()]
            if verbose:
                print("[scsfm_hip.build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            os.chmod(tmp, 0o755)
            os.replace(tmp, LIB)
            with open(LOG, "a") as f:  # who built what (tests/test_build_race.py counts the lines)
                f.write(f"{want} {os.getpid()}\n")
        finally:
            if os.path.exists(tmp):
                os.unlink(tmp)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
