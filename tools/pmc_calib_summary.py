#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of tools/ubench/fetch_calib's kernels (known bytes per kernel) -> the counters' calibration.
    python tools/pmc_calib_summary.py DIR BYTES [--json FILE]     (DIR holds calib_FETCH_SIZE_results.db, calib_WRITE_SIZE_results.db)"""
import json
import os
import sqlite3
import sys


def main(d, nbytes, json_out=None):
    res = {"_what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB per dispatch x 1024) of kernels that touch a buffer of "
                    "`bytes` bytes exactly once (tools/ubench/fetch_calib.hip), as a fraction of the known byte count",
           "bytes": nbytes, "kernels": {}}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(d, f"calib_{c}_results.db")
        if not os.path.exists(path):
            continue
        cur = sqlite3.connect(path).cursor()
        q = ("select k.name, avg(p.counter_value), count(*), avg(p.duration) from pmc_events p join kernels k "
             "on p.dispatch_id = k.dispatch_id where p.counter_name = ? group by k.name order by k.name")
        for name, val, n, dur in cur.execute(q, (c,)).fetchall():
            short = name.split("(")[0].replace("void ", "")
            if "kernel" not in short:
                continue
            e = res["kernels"].setdefault(short, {})
            e[c + "_bytes"] = round(val * 1024)
            e[c + "_over_known"] = round(val * 1024 / nbytes, 4)
            e["us"] = round(dur / 1e3, 1)
            e["known_GBs"] = round(nbytes / (dur * 1e-9) / 1e9, 1)
            e["n"] = n
    for k, v in res["kernels"].items():
        print(k, v)
    if json_out:
        json.dump(res, open(json_out, "w"), indent=1)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0], int(a[1]), a[3] if len(a) > 3 and a[2] == "--json" else None)
