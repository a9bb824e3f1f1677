#!/bin/bash
# Round 5, session c: the fixed-point window guard + wrap detector on hardware; A/B of the kernel alone against session a.
set +e
export TMPDIR=/tmp MIOPEN_FIND_MODE=FAST
O=$PWD/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "compressive or large_gradients or scene" > $O/r05c_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r05c_pytest.log
for D in smooth iid; do DEPTH=$D timeout 300 python tools/rt_vs_ct.py 2>&1 | tail -n 1 | tee $O/r05c_rt_vs_ct_$D.json; done
