#!/bin/bash
# GPU session: switches on the 3-blocks-per-CU build, probe variants, SQ counters.
set +e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-m2}
echo "=== sweep b3 with switches"; SCSFM_HIP_LIB=$R/variants/b3.so timeout 300 python tools/march_sweep.py --rows 64 --switches 2>&1 | tail -n 1 | tee $O/sweep_${TAG}_b3.json
for v in b3park b3stag; do
  echo "=== sweep $v"; SCSFM_HIP_LIB=$R/variants/$v.so timeout 300 python tools/march_sweep.py --rows 64,256 2>&1 | tail -n 1 | tee $O/sweep_${TAG}_$v.json
done
export SCSFM_HIP_LIB=$R/variants/b3.so
bash tools/gpu_sq.sh $TAG > $O/sq_$TAG.txt 2>&1; tail -n 30 $O/sq_$TAG.txt
