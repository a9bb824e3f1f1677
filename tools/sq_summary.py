#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 --pmc counters (rocpd databases), plus the usual ratios when the SQ
wave-cycle counters are present.
    python tools/sq_summary.py gpurun_out/prof_TAG/sq_*_results.db"""
import collections
import sqlite3
import sys


def main(paths):
    vals = collections.defaultdict(dict)  # kernel -> counter -> avg per dispatch
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        q = ("select k.name, k.grid_z, p.counter_name, avg(p.counter_value), count(*) from pmc_events p join kernels k "
             "on p.dispatch_id = k.dispatch_id group by k.name, k.grid_z, p.counter_name")
        for name, gz, c, v, n in cur.execute(q).fetchall():
            if "scsfm" not in name:
                continue
            short = name.split("(")[0].replace("void ", "").replace("scsfm::", "")
            vals[f"{short} grid_z={gz}"][c] = v
    print("# SQ counters, averages per dispatch (rocprofv3 --pmc ... --kernel-trace; the SQ counters are sampled on one shader engine)")
    for k, c in sorted(vals.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_INSTS_VALU", 0))):
        parts = []
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            pct = lambda n: f"{100 * c[n] / wc:.0f}%" if n in c else "-"
            parts.append(f"wait_any {pct('SQ_WAIT_ANY')} wait_inst {pct('SQ_WAIT_INST_ANY')} active {pct('SQ_ACTIVE_INST_ANY')} "
                         f"(valu {pct('SQ_ACTIVE_INST_VALU')} lds {pct('SQ_ACTIVE_INST_LDS')})")
        rest = {n: v for n, v in c.items() if not n.startswith("SQ_WAIT") and not n.startswith("SQ_ACTIVE") and n != "SQ_WAVE_CYCLES"}
        if rest:
            parts.append(" ".join(f"{n.replace('SQ_', '').lower()} {v:.3g}" for n, v in sorted(rest.items())))
        print(f"  {k}: " + " | ".join(parts))


if __name__ == "__main__":
    main(sys.argv[1:])
