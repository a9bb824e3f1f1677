set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 50 --warmup 10 --cpu-seconds 0 2>/dev/null | tee gpurun_out/bench_quick.json
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
