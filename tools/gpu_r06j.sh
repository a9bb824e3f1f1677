# round 6, session j: the finalize launch with the pair losses and the frames' smooth records in workgroups of their own
# (variants/r6fin1.so = the tree) against the serial one (variants/r6fin0.so = library 502d90a96d5f6b11), alternating
# processes on one box: loss-path step (graph replay, eager, one autograd node), then the finalize kernel's own duration
# under rocprofv3 for both
set +e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
: > $O/r06j_ab.jsonl
for ROUND in 1 2; do for V in r6fin1 r6fin3; do
  SCSFM_HIP_LIB=$R/variants/$V.so timeout 300 python bench.py --e2e 0 --cpu-seconds 0 --other-laws 0 --pmc-live 0 --loss-steps 100 --loss-warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); w=d['warp_loss']
print(json.dumps({'lib':'$V','graph_ms':d['warp_loss_ms_per_step'],'eager_ms':w['eager_ms_per_step'],'single':w['single_autograd_node'],'spec_in_step_us':d['roofline']['avg_launch_us'],'pairs_fwd_spec_with_smooth':w['kernel_us'].get('pairs_fwd_spec_with_smooth'),'spec_kernel_only_with_smooth':w['kernel_us'].get('spec_kernel_only_with_smooth'),'lib_id':d['library']['source_id_in_binary']}))" | tee -a $O/r06j_ab.jsonl
done; done
cd /tmp
for V in r6fin1 r6fin3; do
  SCSFM_HIP_LIB=$R/variants/$V.so timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_r06j -o trace_$V -- python $R/bench.py --pmc-live 0 --loss-steps 40 --loss-warmup 5 --cpu-seconds 0 --e2e 0 --graph 0 --kernel-iters 2 --other-laws 0 > $O/r06j_rocprof_$V.log 2>&1
  echo "== $V"; python $R/tools/rocprof_summary.py $O/prof_r06j/trace_${V}_results.db | grep "pair_finalize_kernel" | head -2
done
