#!/bin/bash
# Build tuning variants of libscsfm_hip.so into variants/*.so (git-ignored; travel to the GPU box).  The experimental
# kernels that are not product sources live in variants/src/ (tracked) and are compiled in with -DSCSFM_WITH_MARCH.
#   bash tools/build_variants.sh NAME "-DSCSFM_WIN_W=80 -DSCSFM_WIN_H=24" [NAME2 "flags2" ...]
# Use one with  SCSFM_HIP_LIB=$PWD/variants/NAME.so python tools/ablate_tail.py
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc -fno-slp-vectorize \
    -Wno-unused-function -Iinclude -Ivariants/src -Isc-sfmlearner-release_amd/csrc -DSCSFM_WITH_MARCH $2 -o variants/$1.so sc-sfmlearner-release_amd/csrc/*.hip &
  shift 2
done
wait
ls -la variants
