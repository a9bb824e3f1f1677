#!/usr/bin/env python3
"""The speculative forward alone (SCSFM_DEBUG_KERNEL_ONLY) over segment heights of its column march (SCSFM_MARCH_ROWS)
and depth distributions; optionally with the profiling switches of include/scsfm_hip.h.  Timings only.

    SCSFM_HIP_LIB=variants/b3.so python tools/march_sweep.py [--rows 32,64,128,256] [--depths smooth,iid] [--switches]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sc-sfmlearner-release_amd"))

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--n-ref", type=int, default=2)
    ap.add_argument("--dataset", default="kitti")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rows", default="32,64,128,256")
    ap.add_argument("--depths", default="smooth,iid")
    ap.add_argument("--switches", action="store_true")
    a = ap.parse_args()
    from scsfm_hip import _lib, capi
    lib = _lib.get()
    dev = torch.device("cuda:0")
    out = {"lib": os.path.basename(lib.path), "us": {}}
    for depth in a.depths.split(","):
        a.depth = depth
        x, _ = bench.make_inputs(a, 0, dev)
        det = lambda t: t.detach()
        tgt, K, refs = x["tgt_img"], x["K"], x["ref_imgs"]
        tds, rds = [det(x["tgt_depth"][0])], [[det(r[0])] for r in x["ref_depths"]]
        ps, pis = [det(p) for p in x["poses"]], [det(p) for p in x["poses_inv"]]
        fl = capi.make_flags(1, 1, 1, "zeros")
        _, _, _, ws = capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=(1.0, 0.5))
        for rows in a.rows.split(","):
            os.environ["SCSFM_MARCH_ROWS"] = rows
            cases = {"full": 0}
            if a.switches:
                cases.update({"no_lds_scatter(X1)": 1024, "no_flush(X5)": 32768, "no_tail_pixels(X4)": 8192})
            for name, extra in cases.items():
                # (a switch selects the runtime-flag instantiation: compare switches with "rt", not with "full")
                fn = lambda: capi.photo_geometry_fwd(lib, fl | extra | 16384, tgt, K, refs, tds, rds, ps, pis, hint=(1.0, 0.5), ws=ws)
                out["us"][f"{depth}/{rows}/{name}"] = round(bench._event_time(fn, a.iters) * 1e6, 1)
            if a.switches:
                fn = lambda: capi.photo_geometry_fwd(lib, fl | 2048 | 16384, tgt, K, refs, tds, rds, ps, pis, hint=(1.0, 0.5), ws=ws)
                out["us"][f"{depth}/{rows}/rt"] = round(bench._event_time(fn, a.iters) * 1e6, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
