set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
python __graft_entry__.py --smoke 2>&1 | tail -12
python tools/diag_pose.py 2>&1 | tail -30
python bench.py --steps 30 --warmup 5 --cpu-seconds 0 2>/dev/null | tee gpurun_out/bench_quick.json
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
