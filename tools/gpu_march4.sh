#!/bin/bash
set +e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-m4}; shift
for v in "$@"; do
  case $v in
    *time) echo "=== timing $v"; SCSFM_HIP_LIB=$R/variants/$v.so SCSFM_MARCH_ROWS=${ROWS:-64} timeout 300 python tools/march_timing.py 2>&1 | tail -n 1 | tee $O/timing_${TAG}_$v.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('chunk1','chunk4'):
    if k in d: print(k, d[k]['total'], d[k]['stages'])
print(d.get('wg_life_mean_cycles'))";;
    *) echo "=== sweep $v"; SCSFM_HIP_LIB=$R/variants/$v.so timeout 300 python tools/march_sweep.py --rows ${SWEEP_ROWS:-32,64,128} 2>&1 | tail -n 1 | tee $O/sweep_${TAG}_$v.json;;
  esac
done
