# round 6, session c: the smooth loss riding in the speculative forward (SCSFM_SMOOTH_RIDE=1, the default) against the
# stand-alone smooth forward (=0), same library, alternating processes: loss-path step, in-step kernel time, per-stage times
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r06c_ride.jsonl; : > $O
for r in 0 1 0 1; do
  echo "=== ride $r"
  SCSFM_SMOOTH_RIDE=$r timeout 600 python bench.py --e2e 0 --cpu-seconds 0 --other-laws 1 2> gpurun_out/r06c_err_$r.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
w=d['warp_loss']
print(json.dumps({'ride': $r, 'graph_ms': d['warp_loss_ms_per_step'], 'eager_ms': w['eager_ms_per_step'], 'single': w['single_autograd_node'], 'spec_in_step_us': d['roofline']['avg_launch_us'], 'spec_b2b_us': d['roofline']['back_to_back_launch_us'], 'kernel_us': w['kernel_us'], 'other': d['roofline_other_depth_laws'], 'losses': w['losses']}))" | tee -a $O
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -n 8 | tee gpurun_out/r06c_parity.txt
