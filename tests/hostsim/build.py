"""Compile the product's HIP sources, unchanged, against tests/hostsim/hip/hip_runtime.h with g++.
Result: tests/hostsim/_build/libscsfm_hostsim.so exporting the same C ABI (include/scsfm_hip.h)
with HOST pointers.  CPU-only CI uses it to run the kernels' logic against the oracle; it is test
infrastructure and is never loaded by the product."""
from __future__ import annotations

import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "sc-sfmlearner-release_amd", "csrc")
VSRC = os.path.join(ROOT, "variants", "src")  # experimental kernels that are not product sources (column march, staged forward warp)
# HOSTSIM_EXTRA="-DSCSFM_X=1 ...": a tuning variant of the kernels under the same tests (its objects go to their own
# directory, so that the default build is not disturbed)
EXTRA = os.environ.get("HOSTSIM_EXTRA", "").split()
OUT = os.path.join(HERE, "_build" + ("_" + "".join(c if c.isalnum() else "_" for c in "".join(EXTRA)) if EXTRA else ""))
LIB = os.path.join(OUT, "libscsfm_hostsim.so")


def build(force=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(VSRC, "*")) + [os.path.join(HERE, "hip", "hip_runtime.h"),
                                                           os.path.join(ROOT, "include", "scsfm_hip.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + ".o")
        objs.append(o)
        procs.append(subprocess.Popen(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "c++", "-I", HERE,
                                       "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas",
                                       "-DSCSFM_WITH_MARCH", "-I", VSRC, "-I", CSRC,  # the experimental variants (variants/src/) stay testable here
                                       *EXTRA,
                                       "-c", s, "-o", o]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hostsim build failed")
    subprocess.run(["g++", "-shared", "-o", LIB + ".tmp", *objs], check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
