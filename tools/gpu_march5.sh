#!/bin/bash
set +e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-m5}; shift
for v in "$@"; do
  echo "=== tile $v"; SCSFM_HIP_LIB=$R/variants/$v.so timeout 300 python tools/march_sweep.py --rows 64 2>&1 | tail -n 1 | tee $O/sweep_${TAG}_${v}_tile.json
  echo "=== march $v"; SCSFM_SPEC_KERNEL=march SCSFM_HIP_LIB=$R/variants/$v.so timeout 300 python tools/march_sweep.py --rows ${SWEEP_ROWS:-32,64,128} 2>&1 | tail -n 1 | tee $O/sweep_${TAG}_${v}_march.json
done
