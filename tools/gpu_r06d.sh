# round 6, session d: where the tile evaluates its target frame's smooth loss (behind the flush = product / in front of the
# warp phase), with and without the per-tile bound of the scatter cells' unit; kernel alone with and without the smooth loss
# riding (us / us_ride), alternating processes; then the product's bench line with the ride off / on
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/variants_r06d.jsonl; : > $O
for v in r6epi r6p0 r6nobound r6epi r6p0 r6nobound; do
  echo "=== $v"; SCSFM_HIP_LIB=$PWD/variants/$v.so timeout 400 python tools/variant_check.py --depths smooth,scene 2>&1 | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); d.pop('check',None); print(json.dumps(d))" | tee -a $O
done
O=gpurun_out/r06d_ride.jsonl; : > $O
for r in 0 1 0 1; do
  echo "=== ride $r"
  SCSFM_SMOOTH_RIDE=$r timeout 600 python bench.py --e2e 0 --cpu-seconds 0 --other-laws 1 2> gpurun_out/r06d_err_$r.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
w=d['warp_loss']
print(json.dumps({'ride': $r, 'graph_ms': d['warp_loss_ms_per_step'], 'eager_ms': w['eager_ms_per_step'], 'single': w['single_autograd_node'], 'spec_in_step_us': d['roofline']['avg_launch_us'], 'spec_b2b_us': d['roofline']['back_to_back_launch_us'], 'kernel_us': w['kernel_us'], 'other': d['roofline_other_depth_laws'], 'losses': w['losses']}))" | tee -a $O
done
