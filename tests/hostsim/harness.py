"""Loads the host-simulation build of the kernels (tests/hostsim/build.py) behind the same ctypes
binding class the product uses, for CPU-only tests.  Never imported by the product."""
import functools

from scsfm_hip._lib import CLib

from . import build as _build


@functools.lru_cache(maxsize=1)
def lib():
    return CLib(_build.build())
