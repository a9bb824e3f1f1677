#!/bin/bash
# Round 5, session a: the A/B round 4 ended on (rt vs ct instantiation), the ablation on both bases.
set +e
export TMPDIR=/tmp
O=$PWD/gpurun_out; mkdir -p $O
for D in smooth iid; do DEPTH=$D timeout 300 python tools/rt_vs_ct.py 2>&1 | tail -n 1 | tee $O/r05_rt_vs_ct_$D.json; done
timeout 600 python tools/ablate_tail.py --depths smooth,iid --rounds 3 2>&1 | tail -n 1 | tee $O/r05_ablation.json
