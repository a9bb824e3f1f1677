#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench.py -m gpu -q -x -s -k "entrywise or other_loss_weights or one_rank_over_rccl or hot_path or single_gpu_line or two_ranks" > gpurun_out/pytest_new.log 2>&1; echo "rc=$?"
grep -E "^\[|backward ms|passed|failed|Error|assert" gpurun_out/pytest_new.log | tail -n 40
