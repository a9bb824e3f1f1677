#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`) as text:
per-kernel calls / total / average / share, plus grid, workgroup, VGPR, LDS of each kernel.
    python tools/rocprof_summary.py gpurun_out/prof_r01/r01_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print(f"# rocprofv3 kernel-trace summary of {path}")
    print(f"# {'calls':>6} {'total_us':>12} {'avg_us':>10} {'share%':>7}  kernel")
    for name, calls, tot, avg, pct in rows:
        short = name.split("(")[0].replace("void ", "")
        print(f"  {calls:6d} {tot:12.1f} {avg:10.2f} {pct:7.2f}  {short}")
    print("\n# launch geometry / resources (one row per distinct kernel)")
    q = ("select name, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, "
         "scratch_size, count(*), avg(duration), min(duration), max(duration) from kernels group by name, grid_x, grid_y, grid_z "
         "order by sum(duration) desc")
    for r in cur.execute(q).fetchall():
        short = r[0].split("(")[0].replace("void ", "")
        # median and a trimmed mean beside the plain average: one cold first launch (tens of ms: code object load) would
        # otherwise dominate the average of a few dozen launches
        durs = sorted(d for (d,) in db.execute("select duration from kernels where name = ? and grid_x = ? and grid_y = ? and grid_z = ?",
                                               (r[0], r[1], r[2], r[3])))
        med = durs[len(durs) // 2]
        k = max(1, len(durs) // 20)
        core = durs[:len(durs) - k] if len(durs) > 4 else durs  # (without the slowest 5 %: the cold launches)
        print(f"  {short}\n      grid=({r[1]},{r[2]},{r[3]}) wg={r[4]} vgpr={r[5]} agpr={r[6]} sgpr={r[7]} lds={r[8]} "
              f"scratch={r[9]} n={r[10]} avg={r[11] / 1e3:.2f}us min={r[12] / 1e3:.2f}us max={r[13] / 1e3:.2f}us "
              f"median={med / 1e3:.2f}us avg_without_slowest_5pct={sum(core) / len(core) / 1e3:.2f}us")


if __name__ == "__main__":
    main(sys.argv[1])
