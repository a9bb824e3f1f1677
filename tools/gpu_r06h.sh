# round 6, session h: the full record of the tree as it stands (tools/gpu_round.sh r06: smoke, GPU suite, bench on the four
# configurations and the three depth laws, rocprofv3 kernel trace, FETCH_SIZE / WRITE_SIZE and SQ counter passes), then the
# counters' calibration on known-bytes kernels
set +e
export TMPDIR=/tmp
R=$PWD
bash tools/gpu_round.sh r06
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/prof_calib -o calib_$C -- $R/tools/ubench/fetch_calib > $R/gpurun_out/calib_$C.log 2>&1; echo "calib $C rc=$?"
done
cd $R
python tools/pmc_calib_summary.py gpurun_out/prof_calib 1073741824 --json gpurun_out/r06_fetch_calibration.json | tail -8
# the driver's default command on the same box, after the counters of THIS library are in place (profiles/pmc_latest.json is
# tied to the source id: bench.py then quotes roofline.traffic) -- what profiles/r06_bench_default.json holds
cp gpurun_out/pmc_r06.json profiles/pmc_latest.json
cp gpurun_out/r06_fetch_calibration.json profiles/r06_fetch_calibration.json
python tools/issue_bound.py > /dev/null 2>&1
timeout 900 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_default.json').read().strip().split('\n')[-1])
r=d['roofline']
print('default bench: value', d['value'], 'loss ms', d['warp_loss_ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], 'issue', r['issue_bound_us'], r['frac_of_issue_bound'])
print('cpu', {k: d['cpu_baseline'][k] for k in ('ms_per_step','min_ms_per_step','block_medians_ms','last_two_blocks_differ_by')})
PY
