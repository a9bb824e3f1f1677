#!/usr/bin/env python3
"""Per-kernel FETCH_SIZE / WRITE_SIZE (KiB per dispatch) from rocprofv3 --pmc runs (rocpd databases).
    python tools/pmc_summary.py gpurun_out/prof_TAG   ->  text;  --json FILE also writes a machine-readable summary;
    --prefix pmc_iid_ reads the passes of another workload (tools/gpu_round.sh: iid depth);
    --source-id ID records which library (scsfm_source_id) the passes ran on"""
import json
import os
import sqlite3
import sys


def main(d, json_out=None, prefix="pmc_", source_id=None):
    res = {}
    if source_id:  # which library the counters were collected on: bench.py only quotes them for that very library
        res["_library_source_id"] = source_id
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(d, f"{prefix}{c}_results.db")
        if not os.path.exists(path):
            continue
        cur = sqlite3.connect(path).cursor()
        q = ("select k.name, k.grid_z, avg(p.counter_value), count(*), avg(p.duration) from pmc_events p join kernels k "
             "on p.dispatch_id = k.dispatch_id where p.counter_name = ? group by k.name, k.grid_z order by sum(p.duration) desc")
        try:
            rows = cur.execute(q, (c,)).fetchall()
        except sqlite3.Error:
            q = ("select name, 0, avg(counter_value), count(*), avg(duration) from pmc_events where counter_name = ? "
                 "group by name order by sum(duration) desc")
            rows = cur.execute(q, (c,)).fetchall()
        print(f"# {c}: KiB per dispatch (rocprofv3 --pmc {c} --kernel-trace)")
        for name, gz, val, n, dur in rows:
            if "scsfm" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            print(f"  {short:62s} grid_z={gz:3d} n={n:3d} avg={val:12.1f} KiB  dur={dur / 1e3:8.1f} us")
            res.setdefault(f"{short}|gz{gz}", {})[c] = val
    if json_out:
        json.dump(res, open(json_out, "w"), indent=1)


if __name__ == "__main__":
    a = sys.argv[1:]
    pre, sid = "pmc_", None
    if "--prefix" in a:
        i = a.index("--prefix")
        pre = a[i + 1]
        del a[i:i + 2]
    if "--source-id" in a:
        i = a.index("--source-id")
        sid = a[i + 1]
        del a[i:i + 2]
    main(a[0], a[2] if len(a) > 2 and a[1] == "--json" else None, pre, sid)
