set +e
export TMPDIR=/tmp
for v in c1 s8 s16 c1 s8 s16; do
echo "== $v"; SCSFM_HIP_LIB=$PWD/variants/$v.so python bench.py --loss-steps 50 --loss-warmup 10 --cpu-seconds 0 --e2e 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['warp_loss']['kernel_us']; print(d['warp_loss_ms_per_step'], d['warp_loss']['single_autograd_node']['ms_per_step'], {x:k[x] for x in ('pairs_fwd_spec','spec_kernel_only','pairs_bwd_after_spec','smooth_fwd','smooth_bwd')})"
done
