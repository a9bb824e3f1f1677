#!/bin/bash
# Round 5, session b: the scene depth law (parity cases + bench leg + ablation), the launcher-less bench.
set +e
export TMPDIR=/tmp MIOPEN_FIND_MODE=FAST
O=$PWD/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -s -k "scene" > $O/r05b_pytest_scene.log 2>&1; echo "pytest scene rc=$?"; grep -a "scene\|passed\|failed" $O/r05b_pytest_scene.log | tail -12
timeout 600 python -m pytest tests/test_gpu_bench.py -q -x -k "without_a_launcher" > $O/r05b_pytest_launch.log 2>&1; echo "pytest launcher rc=$?"; tail -3 $O/r05b_pytest_launch.log
for D in smooth scene iid; do
  timeout 600 python bench.py --loss-steps 30 --loss-warmup 5 --depth $D --cpu-seconds 0 --e2e 0 > $O/r05b_bench_$D.json 2>> $O/r05b_bench.err; cut -c1-600 $O/r05b_bench_$D.json
done
timeout 600 python tools/ablate_tail.py --depths smooth,scene,iid --rounds 2 2>&1 | tail -n 1 > $O/r05b_ablation.json; cut -c1-900 $O/r05b_ablation.json
