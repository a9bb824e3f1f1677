#!/usr/bin/env python3
"""Where does the speculative forward's geometry tail spend its time?  Times the fused kernel alone
(SCSFM_DEBUG_KERNEL_ONLY) with the profiling switches of include/scsfm_hip.h.  Timings only -- with a
switch set the results are wrong by construction.

    python tools/ablate_tail.py [--depth smooth|iid]"""
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
    ap.add_argument("--ssim", type=int, default=1)
    a = ap.parse_args()
    from scsfm_hip import _lib, capi
    lib = _lib.get()
    dev = torch.device("cuda:0")
    x, _ = bench.make_inputs(a, 0, dev)
    fl = capi.make_flags(a.ssim, 1, 1, "zeros")
    det = lambda t: t.detach()
    tgt, K, refs = x["tgt_img"], x["K"], x["ref_imgs"]
    tds, rds = [det(x["tgt_depth"][0])], [[det(r[0])] for r in x["ref_depths"]]
    ps, pis = [det(p) for p in x["poses"]], [det(p) for p in x["poses_inv"]]
    _, _, _, ws = capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=(1.0, 0.5))
    cases = {"full": 0, "no_lds_scatter(X1)": 1024, "no_flush(X5)": 32768, "no_tail_pixels(X4)": 8192,
             "X1+X4": 1024 | 8192}
    out = {}
    for name, extra in cases.items():
        fn = lambda: capi.photo_geometry_fwd(lib, fl | extra | 16384, tgt, K, refs, tds, rds, ps, pis, hint=(1.0, 0.5), ws=ws)
        out[name] = round(bench._event_time(fn, a.iters) * 1e6, 2)
    print(json.dumps({"workload": vars(a), "spec_forward_kernel_us": out}))


if __name__ == "__main__":
    main()
