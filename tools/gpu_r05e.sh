#!/bin/bash
# Round 5, session e: the re-speculation pass of the backward against the two separate passes (SCSFM_RESPEC=0), the plain
# forward at 5 workgroups per CU, and the parity tests that exercise a failed speculation.
set +e
export TMPDIR=/tmp MIOPEN_FIND_MODE=FAST
O=$PWD/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_input_gradients.py -q -x -k "weights or speculat or fp64 or hint or oracle or repeated or scales" > $O/r05e_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r05e_pytest.log
: > $O/r05e_variants.jsonl
for RUN in "1 base5" "0 base5" "1 fb5g1" "1 base5" "0 base5" "1 fb5g1"; do
  set -- $RUN
  echo "=== SCSFM_RESPEC=$1 $2"; SCSFM_RESPEC=$1 VARIANT_EXTRA=1 SCSFM_HIP_LIB=$PWD/variants/$2.so timeout 300 python tools/variant_check.py --depths smooth 2>&1 | tail -n 1 | sed "s/^{/{\"respec\": $1, /" | tee -a $O/r05e_variants.jsonl | cut -c1-700
done
