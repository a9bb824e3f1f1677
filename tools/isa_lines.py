#!/usr/bin/env python3
"""Where a kernel's vector instructions come from: static VALU cost (tools/isa_cost.py weights) per source line of a
listing compiled with -gline-tables-only (the .loc directives name the innermost inlined function's line).
    hipcc ... -gline-tables-only -S --cuda-device-only -o pair_g.s csrc/scsfm_pair.hip
    python tools/isa_lines.py pair_g.s KERNEL-SUBSTRING [--ranges FILE:LO-HI=label,...] [--top N]"""
import collections
import re
import sys

from isa_cost import weight


def main(path, filt, ranges, top):
    lines = open(path).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = m.group(2).split("/")[-1]
        else:
            m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"', l)
            if m:
                files[int(m.group(1))] = m.group(2).split("/")[-1]
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for k, (i, name) in enumerate(starts):
        if filt not in name:
            continue
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        cur = ("?", 0)
        cost = collections.Counter()
        cnt = collections.Counter()
        mem = collections.Counter()
        for l in lines[i:end]:
            m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
            if m:
                cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
                continue
            if l.startswith("\t.") or not l.startswith("\t"):
                continue
            t = l.strip()
            if t.startswith(";"):
                continue
            op = t.split()[0]
            if op == "s_endpgm":
                break
            if op.startswith("v_"):
                cost[cur] += weight(op)
                cnt[cur] += 1
            elif op.startswith(("ds_", "global_", "flat_", "buffer_")):
                mem[cur] += 1
        total = sum(cost.values())
        print(f"{name[:110]}\n  VALU n={sum(cnt.values())} cost={total:.0f}")
        if ranges:
            agg = collections.Counter()
            aggn = collections.Counter()
            aggm = collections.Counter()
            for (f, ln), c in cost.items():
                lab = next((lab for (rf, lo, hi, lab) in ranges if rf == f and lo <= ln <= hi), f"{f}:other")
                agg[lab] += c
                aggn[lab] += cnt[(f, ln)]
            for (f, ln), c in mem.items():
                lab = next((lab for (rf, lo, hi, lab) in ranges if rf == f and lo <= ln <= hi), f"{f}:other")
                aggm[lab] += c
            for lab, c in sorted(agg.items(), key=lambda kv: -kv[1]):
                print(f"    {lab:40s} n={aggn[lab]:5d} cost={c:7.0f} ({100 * c / total:4.1f} %)  mem={aggm[lab]}")
        else:
            for (f, ln), c in sorted(cost.items(), key=lambda kv: -kv[1])[:top]:
                print(f"    {f}:{ln:<5d} n={cnt[(f, ln)]:5d} cost={c:7.0f} ({100 * c / total:4.1f} %) mem={mem[(f, ln)]}")


if __name__ == "__main__":
    a = sys.argv[1:]
    ranges, top = [], 60
    if "--ranges" in a:
        j = a.index("--ranges")
        for spec in a[j + 1].split(","):
            loc, lab = spec.split("=")
            f, r = loc.split(":")
            lo, hi = r.split("-")
            ranges.append((f, int(lo), int(hi), lab))
        del a[j:j + 2]
    if "--top" in a:
        j = a.index("--top")
        top = int(a[j + 1])
        del a[j:j + 2]
    main(a[0], a[1] if len(a) > 1 else "", ranges, top)
