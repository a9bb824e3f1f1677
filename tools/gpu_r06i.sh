# round 6, session i: the GPU suite at the final tree (with the live-counter test) and the driver's default bench command,
# whose roofline.traffic is now measured by the run itself (bench.py --pmc-live)
set +e
export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
R=$PWD; O=$R/gpurun_out; mkdir -p $O
echo "=== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_r06i.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu_r06i.log
echo "=== default bench"; S=$(date +%s); timeout 1200 python bench.py > $O/r06i_bench_default.json 2> $O/r06i_bench_default.err; echo "bench rc=$? in $(( $(date +%s) - S )) s"
tail -3 $O/r06i_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06i_bench_default.json').read().strip().split('\n')[-1])
r=d['roofline']
print('default bench: value', d['value'], 'loss ms', d['warp_loss_ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], r['traffic_is'], r['traffic_source'][:200])
print(json.dumps(r['traffic_detail']))
print('cpu', {k: d['cpu_baseline'][k] for k in ('ms_per_step','min_ms_per_step','block_medians_ms','last_two_blocks_differ_by')})
PY
