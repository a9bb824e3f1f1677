#!/bin/bash
# Round 5, session h: the GPU suite at the final tree + the default bench line (traffic now quoted from pmc_latest.json).
set +e
export TMPDIR=/tmp MIOPEN_FIND_MODE=FAST
O=$PWD/gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/r05h_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r05h_pytest_gpu.log
timeout 900 python bench.py > $O/r05h_bench.json 2> $O/r05h_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/r05h_bench.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['warp_loss_ms_per_step'], d['roofline'], d['cpu_baseline']['value'], d['cpu_baseline']['kind'])"
