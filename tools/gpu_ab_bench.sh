#!/bin/bash
# GPU session: bench.py's per-stage timings (kernel_us) over a list of variants/*.so, alternating
#   gpurun -- 'bash tools/gpu_ab_bench.sh a b a b'
set +e
export TMPDIR=/tmp
for v in "$@"; do
echo "== $v"; SCSFM_HIP_LIB=$PWD/variants/$v.so python bench.py --loss-steps 50 --loss-warmup 10 --cpu-seconds 0 --e2e 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=d['warp_loss']['kernel_us']; print(d['warp_loss_ms_per_step'], d['warp_loss']['single_autograd_node']['ms_per_step'], k)"
done
