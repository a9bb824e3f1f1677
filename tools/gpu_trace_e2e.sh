#!/bin/bash
# rocprofv3 kernel trace of a few WHOLE training steps (nets + loss path + optimizer) -> top kernels by total time.
TAG=${1:-e2e}
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o trace -- python $R/bench.py --pmc-live 0 --steps 5 --warmup 2 --loss-steps 2 --loss-warmup 1 --cpu-seconds 0 --graph 0 --kernel-iters 1 > $O/rocprof_$TAG.log 2>&1
cd $R
python - $O/prof_$TAG/trace_results.db <<'P' | tee $O/e2e_kernels_$TAG.txt
import sqlite3, sys, re
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, count(*), sum(duration)/1e3, avg(duration)/1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# {len(rows)} distinct kernels, {sum(r[1] for r in rows)} launches, {tot/1e3:.1f} ms of kernel time in the trace")
for name, n, t, a in rows[:45]:
    short = " ".join(name.split())[:150]
    print(f"{n:7d} {t:11.1f} us {a:9.1f} us {100*t/tot:5.1f}%  {short}")
P
