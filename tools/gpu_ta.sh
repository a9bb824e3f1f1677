#!/bin/bash
# Texture-addresser / cache counter passes of a short eager bench run (counters only with --kernel-trace).
#   gpurun -- 'bash tools/gpu_ta.sh TAG'
TAG=${1:-ta}
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O/prof_$TAG
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TA_[A-Z_a-z0-9]*\|TCP_[A-Z_a-z0-9]*\|TCC_HIT[A-Z_a-z0-9]*\|TCC_MISS[A-Z_a-z0-9]*\|GRBM_GUI_ACTIVE\|GRBM_COUNT\)\b" | sort -u | tr '\n' ' ' | cut -c1-1500 > $O/counters_$TAG.txt
B="python $R/bench.py --pmc-live 0 --loss-steps 3 --loss-warmup 1 --cpu-seconds 0 --e2e 0 --graph 0 --kernel-iters 2"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/prof_$TAG -o ta1 -- $B > $O/rocprof_${TAG}_ta1.log 2>&1; echo "rc=$?"
timeout 600 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --kernel-trace -d $O/prof_$TAG -o ta2 -- $B > $O/rocprof_${TAG}_ta2.log 2>&1; echo "rc=$?"
cd $R
head -c 1500 $O/counters_$TAG.txt; echo
python tools/sq_summary.py $O/prof_$TAG/ta*_results.db 2>&1 | head -20
tail -3 $O/rocprof_${TAG}_ta1.log; tail -3 $O/rocprof_${TAG}_ta2.log
