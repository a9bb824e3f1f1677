set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
python bench.py --steps 50 --warmup 10 --cpu-seconds 0 2>/dev/null | tee gpurun_out/bench_quick.json
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_q -o trace -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --kernel-iters 3 > $R/gpurun_out/rocprof_q.log 2>&1; cd $R
python tools/rocprof_summary.py gpurun_out/prof_q/trace_results.db | grep -v "rocclr" | head -22
