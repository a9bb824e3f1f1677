"""Build libscsfm_hip.so (gfx950) in-tree with hipcc.

    python -m scsfm_hip.build        (from sc-sfmlearner-release_amd/)

The shared object is plain HIP + a C ABI (include/scsfm_hip.h); it does not link against torch.
It is written next to this file so that it travels with the source tree to the GPU box.
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIB = os.path.join(HERE, "libscsfm_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-munsafe-fp-atomics",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         # the SLP vectoriser pairs unrelated scalar fp32 chains into v_pk_* with 4 v_mov per packed op (+5 % VALU
         # in the tiled pass); the 2-wide math that pays is written with vector types in csrc/scsfm_ssim.h
         "-fno-slp-vectorize"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + \
        [os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "scsfm_hip.h")]


def source_id():
    """First 16 hex digits of the sha256 over every file the library is built from (names and contents, sorted).  It is
    compiled into the binary (scsfm_source_id) so that a loaded .so can be tied to the sources next to it."""
    h = hashlib.sha256()
    for path in deps():
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in deps())


def build(force=False, verbose=True, extra=()):
    """Compile every .hip file under csrc/ into one shared object.  Raises on failure."""
    if not force and not is_stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libscsfm_hip.so cannot be built on this machine")
    tmp = LIB + ".tmp"
    cmd = [hipcc, *FLAGS, f'-DSCSFM_SOURCE_ID="{source_id()}"', *extra, "-o", tmp, *sources()]
    if verbose:
        print("[scsfm_hip.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
