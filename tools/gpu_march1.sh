#!/bin/bash
# GPU session: parity tests, then the march kernel's segment-height / occupancy sweep.
set +e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-m1}
echo "=== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -n 6 $O/pytest_gpu_$TAG.log
echo "=== sweep default (4 blocks/CU)"; timeout 300 python tools/march_sweep.py --switches 2>&1 | tail -n 1 | tee $O/sweep_${TAG}_b4.json
for v in b3 b2; do
  echo "=== sweep $v"; SCSFM_HIP_LIB=$R/variants/$v.so timeout 300 python tools/march_sweep.py 2>&1 | tail -n 1 | tee $O/sweep_${TAG}_$v.json
done
echo "=== quick bench"; bash tools/gpu_quick.sh 2>&1 | tail -n 3
