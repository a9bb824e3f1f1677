#!/bin/bash
# SQ counter passes (wave cycles / waits / active cycles; instruction counts) of a short eager bench run.
# Counters only with --kernel-trace (no other trace domain).   gpurun -- 'bash tools/gpu_sq.sh TAG'
TAG=${1:-sq}
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O/prof_$TAG
cd /tmp
B="python $R/bench.py --pmc-live 0 --loss-steps 3 --loss-warmup 1 --cpu-seconds 0 --e2e 0 --graph 0 --kernel-iters 2"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $O/prof_$TAG -o sq_cycles -- $B > $O/rocprof_${TAG}_cycles.log 2>&1; echo "rc=$?"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/prof_$TAG -o sq_insts -- $B > $O/rocprof_${TAG}_insts.log 2>&1; echo "rc=$?"
cd $R
python tools/sq_summary.py $O/prof_$TAG/sq_*_results.db
