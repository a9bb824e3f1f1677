# round 6, session f: the GPU suite on the tree as it stands (smooth loss riding, per-tile window bound, new tests), the
# default bench line
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -n 60 > gpurun_out/r06f_pytest.txt
tail -n 45 gpurun_out/r06f_pytest.txt
timeout 600 python bench.py > gpurun_out/r06f_bench.json 2> gpurun_out/r06f_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06f_bench.json').read().strip().split('\n')[-1])
print(json.dumps({k: d[k] for k in ('value','ms_per_step','warp_loss_ms_per_step','roofline')}, indent=None)[:3000])
print(json.dumps({k: d['cpu_baseline'][k] for k in ('value','ms_per_step','min_ms_per_step','block_medians_ms','last_two_blocks_differ_by','cores')}))
PY
