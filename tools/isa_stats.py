#!/usr/bin/env python3
"""Static instruction mix of the kernels in a gfx950 .s file (hipcc -save-temps).
    python tools/isa_stats.py file.s [name-substring]"""
import collections
import re
import sys


def main(path, filt=""):
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for k, (i, name) in enumerate(starts):
        if filt not in name:
            continue
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        body = []
        for l in lines[i:end]:
            if l.startswith("\t.") or not l.startswith("\t"):
                if "s_endpgm" in l:
                    break
                continue
            t = l.strip()
            if t.startswith(";"):
                continue
            body.append(t.split()[0])
            if body[-1] == "s_endpgm":
                break
        c = collections.Counter(body)
        g = collections.Counter()
        for op, n in c.items():
            if op.startswith("v_"):
                g["valu"] += n
                if any(x in op for x in ("rcp", "sqrt", "rsq", "exp", "log", "sin", "cos", "div_")):
                    g["valu_trans/div"] += n
            elif op.startswith("s_waitcnt"):
                g["waitcnt"] += n
            elif op.startswith("s_barrier"):
                g["barrier"] += n
            elif op.startswith("s_"):
                g["salu"] += n
            elif op.startswith("global_load"):
                g["gload"] += n
            elif op.startswith("global_atomic"):
                g["gatomic"] += n
            elif op.startswith("global_store"):
                g["gstore"] += n
            elif op.startswith("ds_"):
                g["lds"] += n
            else:
                g[op] += n
        print(f"{name[:90]}\n   total={sum(c.values())} {dict(g)}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
