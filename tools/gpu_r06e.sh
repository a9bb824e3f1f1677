# round 6, session e: the loss-path step with the smooth loss riding (product: behind the flush; r6p0: in front of the warp
# phase) against the stand-alone smooth forward, alternating processes; then the FETCH_SIZE / WRITE_SIZE calibration
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
O=gpurun_out/r06e_ride.jsonl; : > $O
run() {  # lib-or-empty ride tag
  SCSFM_HIP_LIB=$1 SCSFM_SMOOTH_RIDE=$2 timeout 600 python bench.py --e2e 0 --cpu-seconds 0 --other-laws 1 2> gpurun_out/r06e_err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
w=d['warp_loss']
print(json.dumps({'tag': '$3', 'graph_ms': d['warp_loss_ms_per_step'], 'eager_ms': w['eager_ms_per_step'], 'single': w['single_autograd_node'], 'spec_in_step_us': d['roofline']['avg_launch_us'], 'kernel_us': w['kernel_us'], 'other': d['roofline_other_depth_laws']}))" | tee -a $O
}
for i in 1 2; do
  run $R/variants/r6epi.so 0 epi_ride0
  run $R/variants/r6epi.so 1 epi_ride1
  run $R/variants/r6p0.so 1 p0_ride1
done
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/prof_calib -o calib_$C -- $R/tools/ubench/fetch_calib > $R/gpurun_out/calib_$C.log 2>&1; echo "calib $C rc=$?"
done
cd $R
python tools/pmc_calib_summary.py gpurun_out/prof_calib 1073741824 --json gpurun_out/r06_fetch_calibration.json
