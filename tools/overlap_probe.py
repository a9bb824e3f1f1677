#!/usr/bin/env python3
"""Would the smooth loss hide under the pair kernels?  The two are independent (compute_smooth_loss reads the same
depth maps and images as compute_photo_and_geometry_loss and nothing the latter writes).  Times, with HIP events on
the main stream: pairs forward then smooth forward on ONE stream, against the smooth forward forked to a side stream
before the pairs forward is launched and joined after it; the same for the two backwards.

    python tools/overlap_probe.py [--depth smooth]"""
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
    for k, v in (("batch", 12), ("height", 256), ("width", 832), ("n_ref", 2), ("iters", 50)):
        ap.add_argument("--" + k.replace("_", "-"), type=int, default=v)
    ap.add_argument("--dataset", default="kitti")
    ap.add_argument("--depth", default="smooth")
    a = ap.parse_args()
    from scsfm_hip import _lib, capi
    lib = _lib.get()
    dev = torch.device("cuda:0")
    x, _ = bench.make_inputs(a, 0, dev)
    det = lambda t: t.detach()
    tgt, K, refs = x["tgt_img"], x["K"], x["ref_imgs"]
    tds, rds = [det(x["tgt_depth"][0])], [[det(r[0])] for r in x["ref_depths"]]
    ps, pis = [det(p) for p in x["poses"]], [det(p) for p in x["poses_inv"]]
    fl = capi.make_flags(1, 1, 1, "zeros")
    frames, imgs = tds + [r[0] for r in rds], [tgt] + list(refs)
    one, half = torch.ones(1, device=dev), torch.full((1,), 0.5, device=dev)
    side = torch.cuda.Stream()
    _, _, _, ws = capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=(1.0, 0.5))
    _, sws = capi.smooth_multi_fwd(lib, frames, imgs)

    def pairs_fwd():
        capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=(1.0, 0.5), ws=ws)

    def smooth_fwd():
        capi.smooth_multi_fwd(lib, frames, imgs)

    def pairs_bwd():
        capi.photo_geometry_bwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, ws, one, half)

    def smooth_bwd():
        capi.smooth_multi_bwd(lib, frames, imgs, sws, one)

    def forked(first_on_side, then_on_main):
        def fn():
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                first_on_side()
            then_on_main()
            main.wait_stream(side)
        return fn

    def serial(a_, b_):
        def fn():
            a_(); b_()
        return fn

    out = {}
    for name, fn in (("fwd_serial", serial(pairs_fwd, smooth_fwd)), ("fwd_forked", forked(smooth_fwd, pairs_fwd)),
                     ("fwd_pairs_only", pairs_fwd), ("fwd_smooth_only", smooth_fwd),
                     ("bwd_serial", serial(pairs_bwd, smooth_bwd)), ("bwd_forked", forked(smooth_bwd, pairs_bwd)),
                     ("bwd_pairs_only", pairs_bwd), ("bwd_smooth_only", smooth_bwd)):
        out[name] = round(bench._event_time(fn, a.iters) * 1e6, 1)
    print(json.dumps({"us": out}))


if __name__ == "__main__":
    main()
