"""ctypes binding of the C ABI declared in include/scsfm_hip.h.

``get()`` returns the product library, libscsfm_hip.so (hipcc, gfx950), and nothing else: if the
shared object is missing it is built in-tree with hipcc, and if that is impossible the call raises
-- there is no CPU or eager fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes
import os
import re
import threading

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so) while libscsfm_hip.so is linked
# against the one under /opt/rocm.  Both carry the same SONAME, so whichever is loaded first serves
# the whole process: importing torch first makes the kernels launch on the very runtime that owns
# torch's streams and allocations.  (Loaded the other way round, the first launch fails with
# hipErrorNoDevice.)
import torch  # noqa: F401  (must precede ctypes.CDLL below)

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "scsfm_hip.h")
# SCSFM_HIP_LIB points at a library built elsewhere (tuning variants, a system-wide install)
LIB_PATH = os.environ.get("SCSFM_HIP_LIB") or os.path.join(HERE, "libscsfm_hip.so")

ABI_VERSION = 9  # include/scsfm_hip.h

_CTYPES = {"int": ctypes.c_int, "unsigned": ctypes.c_uint, "size_t": ctypes.c_size_t, "double": ctypes.c_double}
_DECL = re.compile(r"^(int|size_t)\s+(scsfm_\w+)\s*\(([^)]*)\)\s*;", re.M | re.S)


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function the header declares."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for ret, name, args in _DECL.findall(text):
        argtypes = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPES[a.split()[0]])
        out[name] = (_CTYPES[ret], argtypes)
    return out


class ScsfmError(RuntimeError):
    pass


class CLib:
    """A loaded implementation of the C ABI.  Every declared symbol must be present."""

    def __init__(self, path):
        self.path = path
        self._dll = ctypes.CDLL(path)
        self.decls = parse_header()
        self._fn = {}
        for name, (ret, argtypes) in self.decls.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise ScsfmError(f"{path} does not export {name} (declared in {HEADER})") from e
            fn.restype = ret
            fn.argtypes = argtypes
            self._fn[name] = fn
        if self._dll.scsfm_abi_version() != ABI_VERSION:
            raise ScsfmError(f"{path}: ABI version mismatch")

    def source_id(self):
        """The source hash compiled into the binary (scsfm_hip/build.py: source_id)."""
        buf = ctypes.create_string_buffer(64)
        self._fn["scsfm_source_id"](buf, 64)
        return buf.value.decode()

    def call(self, name, *args):
        """Invoke an int-returning entry point; raise on a non-zero status."""
        rc = self._fn[name](*args)
        if rc != 0:
            kind = "rejected argument" if rc == -1 else "hipError_t"
            raise ScsfmError(f"{name} failed with status {rc} ({kind})")

    def size(self, name, *args):
        return int(self._fn[name](*args))


_lock = threading.Lock()
_lib = None


def get() -> CLib:
    """The HIP library (singleton).  Ties the binary to the sources next to it BEFORE loading it: the source id compiled
    into libscsfm_hip.so is read from the file (scsfm_hip.build.binary_source_id -- a stale binary may lack symbols or
    carry another ABI number, which the strict loader below would report as an error instead of rebuilding), and a
    missing or stale library is built with hipcc under a file lock, so that the N ranks of a torchrun job that all
    arrive here at once compile once.  Without hipcc a stale library is refused.  A library named by SCSFM_HIP_LIB (a
    tuning variant, a system-wide install) is taken as it is."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                from . import build as _build
                own = "SCSFM_HIP_LIB" not in os.environ
                if own and _build.is_stale():
                    have = _build.binary_source_id(LIB_PATH)
                    try:
                        _build.build()
                    except Exception as e:
                        what = f"was built from other sources ({have}) than the tree's ({_build.source_id()})" if have \
                            else "is missing (or carries no source id)"
                        raise ScsfmError(f"{LIB_PATH} {what} and cannot be built here: {e}") from e
                lib = CLib(LIB_PATH)
                if own and lib.source_id() != _build.source_id():
                    raise ScsfmError(f"{LIB_PATH}: its source id {lib.source_id()} is not the tree's {_build.source_id()}")
                _lib = lib
    return _lib
