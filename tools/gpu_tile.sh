#!/bin/bash
# GPU session: the tile kernel's variants (kernel-only time, smooth and iid depth)
set +e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-t}; shift
for v in "$@"; do
  echo "=== $v"; SCSFM_HIP_LIB=$R/variants/$v.so timeout 300 python tools/march_sweep.py --rows 64 --iters 40 2>&1 | tail -n 1 | tee $O/sweep_${TAG}_$v.json
done
