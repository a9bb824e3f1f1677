#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench (smooth + iid), rocprofv3 kernel trace, PMC passes.
# Usage (from the build container):  gpurun --timeout 1800 -- 'bash tools/gpu_round.sh TAG'
set +e
TAG=${1:-rXX}
export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
R=$PWD
O=$R/gpurun_out
mkdir -p $O
echo "=== smoke"; python __graft_entry__.py --smoke > $O/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke_$TAG.log
echo "=== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu_$TAG.log
echo "=== bench (+ input pipeline leg)"; timeout 1200 python bench.py --steps 50 --warmup 10 --data loader > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?"; cut -c1-2600 $O/bench_$TAG.json; tail -2 $O/bench_$TAG.err
echo "=== bench over RCCL at world size 1 (process group on nccl, DistributedDataParallel, default and exact mode)"
for EX in 0 1; do MASTER_PORT=$((29600 + EX)) timeout 600 python bench.py --steps 10 --warmup 3 --loss-steps 10 --loss-warmup 3 --cpu-seconds 0 --force-dist nccl --exact $EX --graph 2 > $O/bench_${TAG}_rccl1_exact$EX.json 2>> $O/bench_$TAG.err; cut -c1-700 $O/bench_${TAG}_rccl1_exact$EX.json; done
echo "=== bench iid / scene (the hot path alone on the other two depth laws)"
for D in iid scene; do timeout 600 python bench.py --loss-steps 30 --loss-warmup 5 --depth $D --cpu-seconds 0 --e2e 0 > $O/bench_${TAG}_$D.json 2>> $O/bench_$TAG.err; cut -c1-400 $O/bench_${TAG}_$D.json; done
echo "=== bench --channels-last 1 (nets in NHWC; reported, not the headline)"
timeout 900 python bench.py --steps 20 --warmup 5 --loss-steps 10 --loss-warmup 3 --cpu-seconds 0 --channels-last 1 > $O/bench_${TAG}_channels_last.json 2>> $O/bench_$TAG.err; cut -c1-300 $O/bench_${TAG}_channels_last.json
echo "=== bench configs[3] (ResNet50 encoder, batch 8 per GPU) and configs[4] (NYU 256x320, sequence length 5, batch 16): whole training step + loss path"
timeout 900 python bench.py --steps 10 --warmup 3 --resnet-layers 50 --batch 8 --loss-steps 30 --loss-warmup 5 --cpu-seconds 0 > $O/bench_${TAG}_cfg3.json 2>> $O/bench_$TAG.err; cut -c1-420 $O/bench_${TAG}_cfg3.json
timeout 900 python bench.py --steps 10 --warmup 3 --dataset nyu --height 256 --width 320 --n-ref 4 --batch 16 --loss-steps 30 --loss-warmup 5 --cpu-seconds 0 > $O/bench_${TAG}_cfg4.json 2>> $O/bench_$TAG.err; cut -c1-420 $O/bench_${TAG}_cfg4.json
cd /tmp
# (the headline law only -- --other-laws 0 -- and enough eager steps that the dominant kernel has >= 60 launches in the trace)
echo "=== rocprof kernel trace"; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o trace -- python $R/bench.py --pmc-live 0 --loss-steps 60 --loss-warmup 5 --cpu-seconds 0 --e2e 0 --graph 0 --kernel-iters 3 --other-laws 0 > $O/rocprof_$TAG.log 2>&1; echo "rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "=== rocprof pmc $C"; timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/prof_$TAG -o pmc_$C -- python $R/bench.py --pmc-live 0 --loss-steps 3 --loss-warmup 1 --cpu-seconds 0 --e2e 0 --graph 0 --kernel-iters 2 --other-laws 0 > $O/rocprof_${TAG}_$C.log 2>&1; echo "rc=$?"
done
echo "=== rocprof pmc, iid and scene depth"
for D in iid scene; do for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/prof_$TAG -o pmc_${D}_$C -- python $R/bench.py --pmc-live 0 --loss-steps 3 --loss-warmup 1 --cpu-seconds 0 --e2e 0 --graph 0 --kernel-iters 2 --other-laws 0 --depth $D > $O/rocprof_${TAG}_${D}_$C.log 2>&1; echo "rc=$?"
done; done
echo "=== rocprof kernel trace, scene depth"; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o trace_scene -- python $R/bench.py --pmc-live 0 --loss-steps 20 --loss-warmup 5 --cpu-seconds 0 --e2e 0 --graph 0 --kernel-iters 3 --other-laws 0 --depth scene > $O/rocprof_${TAG}_scene.log 2>&1; echo "rc=$?"
cd $R
# the counters, tied to the library they were collected on (bench.py quotes profiles/pmc_latest.json only for that library)
SID=$(python -c "import sys; sys.path.insert(0, 'sc-sfmlearner-release_amd'); from scsfm_hip import _lib; print(_lib.get().source_id())" 2>/dev/null | tail -n 1)
python tools/pmc_summary.py $O/prof_$TAG --json $O/pmc_$TAG.json --source-id "$SID" > $O/pmc_$TAG.txt 2>&1
python tools/pmc_summary.py $O/prof_$TAG --prefix pmc_iid_ --json $O/pmc_${TAG}_iid.json --source-id "$SID" > $O/pmc_${TAG}_iid.txt 2>&1
python tools/pmc_summary.py $O/prof_$TAG --prefix pmc_scene_ --json $O/pmc_${TAG}_scene.json --source-id "$SID" > $O/pmc_${TAG}_scene.txt 2>&1
python tools/rocprof_summary.py $O/prof_$TAG/trace_scene_results.db | grep "pair_fwd_spec_kernel" | head -1 | tee $O/scene_kernel_$TAG.txt
head -4 $O/pmc_$TAG.txt
python tools/rocprof_summary.py $O/prof_$TAG/trace_results.db | grep -v "at::native\|rocclr" | head -20
bash tools/gpu_sq.sh $TAG > $O/sq_$TAG.txt 2>&1; tail -14 $O/sq_$TAG.txt
echo "=== library launches per single-node step (20 eager steps of compute_total_loss + backward under rocprofv3)"
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o sn -- python $R/tools/step_launches.py --steps 20 > /dev/null 2>&1)
python tools/rocprof_summary.py $O/prof_$TAG/sn_results.db | grep "scsfm::" | head -14 | tee $O/single_node_launches_$TAG.txt
echo "=== stage timeline of the tile kernel (PROBE_TIMING build) and the column-march variant for the record"
if [ -f variants/t4time.so ]; then SCSFM_HIP_LIB=$R/variants/t4time.so timeout 300 python tools/march_timing.py 2>&1 | tail -n 1 | tee $O/timing_${TAG}_tile.json; fi
if [ -f variants/m3.so ]; then
  SCSFM_HIP_LIB=$R/variants/m3.so timeout 300 python tools/march_sweep.py --rows 64 2>&1 | tail -n 1 | tee $O/sweep_${TAG}_tile.json
  SCSFM_SPEC_KERNEL=march SCSFM_HIP_LIB=$R/variants/m3.so timeout 300 python tools/march_sweep.py --rows 32,64,128 2>&1 | tail -n 1 | tee $O/sweep_${TAG}_march.json
fi
