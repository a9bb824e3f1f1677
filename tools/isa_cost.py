#!/usr/bin/env python3
"""Static VALU issue cost of the kernels in a gfx950 .s file, weighted with the issue intervals measured on
MI355X by tools/ubench/rates.hip (gpurun_out/ubench_rates2.txt; unit = one full-rate wave64 instruction,
~1.05 ns per instruction per SIMD at >= 2 waves per SIMD):
    full rate (1)   : v_add/sub/mul/fmac/fma(1.1)_f32, v_mov, v_and/or/xor, v_add/sub_u32, v_lshrrev, v_ashrrev,
                      also with abs / neg / clamp modifiers
    half rate (1.8) : v_min/max/med3 (f32, i32, u32), v_cmp_*, v_cndmask, v_cvt_*, v_floor/trunc/rndne/fract,
                      v_lshlrev, v_lshl_add, v_add3, v_lshl_or, v_and_or, v_bfe, v_mul_*24, v_mad_*24, v_mul_lo,
                      every v_pk_* (two lanes: no gain), fp64, DPP, v_readlane, v_add_co
    quarter (3.4)   : v_rcp / v_rsq / v_sqrt / v_exp / v_log
    python tools/isa_cost.py file.s [name-substring] [-v]"""
import collections
import re
import sys

FULL = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_mov_b32", "v_and_b32", "v_or_b32",
        "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_fmaak_f32",
        "v_fmamk_f32", "v_not_b32", "v_accvgpr")
FMA = ("v_fma_f32",)
QUARTER = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def weight(op):
    base = op
    for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
        base = base.replace(suf, "")
    if "_dpp" in op or "_sdwa" in op:
        return 1.8
    if any(base.startswith(q) for q in QUARTER):
        return 3.4
    if base in FMA:
        return 1.12
    if base in FULL:
        return 1.0
    return 1.8


def main(path, filt="", verbose=False):
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for k, (i, name) in enumerate(starts):
        if filt not in name:
            continue
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        ops = collections.Counter()
        other = collections.Counter()
        for l in lines[i:end]:
            if l.startswith("\t.") or not l.startswith("\t"):
                continue
            t = l.strip()
            if t.startswith(";"):
                continue
            op = t.split()[0]
            if op == "s_endpgm":
                break
            if op.startswith("v_"):
                ops[op] += 1
            else:
                key = ("ds" if op.startswith("ds_") else "gload" if op.startswith("global_load") else
                       "gstore" if op.startswith("global_store") else "gatomic" if op.startswith("global_atomic") else
                       "waitcnt" if op.startswith("s_waitcnt") else "barrier" if op == "s_barrier" else
                       "salu" if op.startswith("s_") else op)
                other[key] += 1
        n = sum(ops.values())
        cost = sum(weight(o) * c for o, c in ops.items())
        cls = collections.Counter()
        for o, c in ops.items():
            w = weight(o)
            cls["full" if w <= 1.0 else "fma" if w < 1.5 else "half" if w < 3 else "quarter"] += c
        print(f"{name[:100]}\n   VALU n={n} cost={cost:.0f} units  {dict(cls)}  other={dict(other)}")
        if verbose:
            for o, c in sorted(ops.items(), key=lambda kv: -weight(kv[0]) * kv[1])[:45]:
                print(f"      {o:28s} n={c:5d} w={weight(o):.2f} cost={weight(o) * c:7.0f}")


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if x != "-v"]
    main(a[0], a[1] if len(a) > 1 else "", "-v" in sys.argv)
