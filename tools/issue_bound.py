#!/usr/bin/env python3
"""Static vector-issue cost of the dominant kernel of THIS tree -> profiles/issue_cost_latest.json, tied to the library's
source id (bench.py quotes it in `roofline.issue_bound_us` only for the library it was computed on).

    python tools/issue_bound.py            (needs hipcc; cross-compiles gfx950 without a GPU)

units = sum over the kernel's static vector instructions of their issue weight (tools/isa_cost.py: 1 = a full-rate wave64
instruction; weights measured on MI355X, profiles/r02_ubench_valu_rates*.txt); one unit issues in ~1.0 ns per SIMD at >= 2
waves per SIMD.  The listing contains uniformly skipped code (the wide scatter window; the smooth-loss block of the
pair-directions that do not carry a frame), so the figure is an upper bound of what a wave issues."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "sc-sfmlearner-release_amd")]
from isa_cost import weight  # noqa: E402

KERNEL = "_ZN5scsfm20pair_fwd_spec_kernelIfLb1ELj7ELb0ELb0E"
NS_PER_UNIT = 1.0


def main():
    from scsfm_hip import build
    out = "/tmp/issue_bound_pair.s"
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *flags, "-gline-tables-only", "-S", "--cuda-device-only", "-I",
                    os.path.join(ROOT, "include"), "-o", out, os.path.join(build.CSRC, "scsfm_pair.hip")], check=True,
                   capture_output=True)
    lines = open(out).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l) or re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = m.group(2).split("/")[-1]
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    cur, units, n, smooth_units = None, 0.0, 0, 0.0
    for l in lines[start:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = files.get(int(m.group(1)))
            continue
        if not l.startswith("\t") or l.startswith("\t.") or l.strip().startswith(";"):
            continue
        op = l.strip().split()[0]
        if op.startswith("v_"):
            w = weight(op)
            units += w
            n += 1
            if cur == "scsfm_smooth_math.h":
                smooth_units += w
    res = {"_library_source_id": build.source_id(), "kernel": "pair_fwd_spec_kernel<float,true,7u,false,false>",
           "static_valu_instructions": n, "issue_units_per_thread": round(units, 1),
           "of_which_attributed_to_scsfm_smooth_math_h": round(smooth_units, 1),
           "ns_per_unit_per_simd": NS_PER_UNIT,
           "_note": "static upper bound (uniformly skipped branches included); issue_bound_us = units x waves per SIMD per launch x ns_per_unit"}
    path = os.path.join(ROOT, "profiles", "issue_cost_latest.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
