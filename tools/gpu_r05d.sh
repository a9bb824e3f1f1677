#!/bin/bash
# Round 5, session d: the whole GPU suite at the current tree + the gradient dumps tools/diag_gates.py analyses off the box.
set +e
export TMPDIR=/tmp MIOPEN_FIND_MODE=FAST
O=$PWD/gpurun_out; mkdir -p $O
timeout 300 python tools/diag_gates.py dump 4 iid > $O/r05d_diag.log 2>&1; tail -1 $O/r05d_diag.log
timeout 2400 python -m pytest tests -m gpu -q > $O/r05d_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r05d_pytest_gpu.log
