# round 6, session g: the reference's call structure with the smooth loss as a third output of the pair losses' autograd node
# (its gradient arrives with theirs: no smooth backward launch, no second gradient per depth map): loss-path step, then the suite
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r06g_ride.jsonl; : > $O
for r in 1 0 1 0; do
  SCSFM_SMOOTH_RIDE=$r timeout 600 python bench.py --e2e 0 --cpu-seconds 0 --other-laws 0 2> gpurun_out/r06g_err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
w=d['warp_loss']
print(json.dumps({'ride': $r, 'graph_ms': d['warp_loss_ms_per_step'], 'eager_ms': w['eager_ms_per_step'], 'single': w['single_autograd_node'], 'spec_in_step_us': d['roofline']['avg_launch_us'], 'kernel_us': w['kernel_us']}))" | tee -a $O
done
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -n 30 > gpurun_out/r06g_pytest.txt
tail -n 6 gpurun_out/r06g_pytest.txt
