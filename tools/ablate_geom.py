#!/usr/bin/env python3
"""Where does the geometry pass spend its time?  Times the backward stage of bench.py's workload
(geometry pass + pose reduce + combine, after a speculative forward) with the profiling switches of
include/scsfm_hip.h (SCSFM_DEBUG_X1..X4) that drop one ingredient each.  Results are timings only --
with a switch set the gradients are wrong by construction.

    python tools/ablate_geom.py [--batch 12 --height 256 --width 832 --n-ref 2 --depth smooth]"""
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
    ap.add_argument("--depth", default="smooth")
    ap.add_argument("--dataset", default="kitti")
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    from scsfm_hip import _lib, capi
    lib = _lib.get()
    dev = torch.device("cuda:0")
    x, _ = bench.make_inputs(a, 0, dev)
    fl = capi.make_flags(1, 1, 1, "zeros")
    det = lambda t: t.detach()
    tgt, K, refs = x["tgt_img"], x["K"], x["ref_imgs"]
    tds, rds = [det(x["tgt_depth"][0])], [[det(r[0])] for r in x["ref_depths"]]
    ps, pis = [det(p) for p in x["poses"]], [det(p) for p in x["poses_inv"]]
    one, half = torch.ones(1, device=dev), torch.full((1,), 0.5, device=dev)
    _, _, _, ws = capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=(1.0, 0.5))
    cases = {"full": 0, "no_scatter(X1)": 1024, "no_dense_store(X2)": 2048, "no_block_sum(X3)": 4096,
             "no_colour_taps(X4)": 8192, "no_flush(X5)": 32768, "X1+X4": 1024 | 8192, "X4+X5": 8192 | 32768, "X1+X3+X4": 1024 | 4096 | 8192,
             "X1+X2+X3+X4": 1024 | 2048 | 4096 | 8192}
    out = {}
    for name, extra in cases.items():
        fn = lambda: capi.photo_geometry_bwd(lib, fl | extra, tgt, K, refs, tds, rds, ps, pis, ws, one, half)
        out[name] = round(bench._event_time(fn, a.iters) * 1e6, 2)
    print(json.dumps({"workload": vars(a), "bwd_stage_us": out}))


if __name__ == "__main__":
    main()
