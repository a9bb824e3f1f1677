#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench.py tests/test_augment.py -m gpu -q -s -k "entrywise or other_loss_weights or one_rank_over_rccl or single_gpu_line or two_ranks or baseline_size_fixture or reference_fixture" > gpurun_out/pytest_new.log 2>&1; echo "rc=$?"
grep -E "^\[|backward ms|passed|failed|^E  |g_tgt_depth:|g_ref" gpurun_out/pytest_new.log | cut -c1-330 | tail -n 50
