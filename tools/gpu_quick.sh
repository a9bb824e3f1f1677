set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --loss-steps 50 --loss-warmup 10 --cpu-seconds 0 --e2e 0 2>/dev/null | tee gpurun_out/bench_quick.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['warp_loss']['kernel_us'], d['roofline'])"
