#!/bin/bash
# Round 5, session f: re-speculation pass variants (compile-time flags, 3 / 4 workgroups per CU) against the two passes.
set +e
export TMPDIR=/tmp MIOPEN_FIND_MODE=FAST
O=$PWD/gpurun_out; mkdir -p $O
: > $O/r05f_variants.jsonl
for RUN in "1 rs4" "1 rs3" "0 rs4" "1 rs4" "1 rs3" "0 rs4"; do
  set -- $RUN
  echo "=== SCSFM_RESPEC=$1 $2"; SCSFM_RESPEC=$1 VARIANT_EXTRA=1 SCSFM_HIP_LIB=$PWD/variants/$2.so timeout 300 python tools/variant_check.py --depths smooth 2>&1 | tail -n 1 | sed "s/^{/{\"respec\": $1, /" | tee -a $O/r05f_variants.jsonl | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'respec', d['respec'], 'spec', d['us'], 'plain fwd', d['us_plain_fwd'], 'fallback bwd', d['us_fallback_bwd'], 'gd_abs', d['check']['smooth']['gd_abs'])"
done
