#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> gpurun_out/prof_$TAG/ + a printed summary.
TAG=${1:-quick}
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o trace -- python $R/bench.py --pmc-live 0 --loss-steps 20 --loss-warmup 5 --cpu-seconds 0 --e2e 0 --graph 0 --kernel-iters 3 > $O/rocprof_$TAG.log 2>&1
cd $R
python tools/rocprof_summary.py $O/prof_$TAG/trace_results.db | head -24 | cut -c1-150
