#!/usr/bin/env python3
"""Multi-scale step (--num-scales N, loss_functions.py:77-82): the coarser scales' depth maps read in place by the
pair kernels (scsfm_pair_desc::depth_shift) against materialising their nearest up-sampling with F.interpolate
under autograd (what the reference does).  Device time per step (forward + backward of the photometric / geometry
loss + smooth loss), configs[1] shape.

    python tools/scales_probe.py [--scales 4] [--steps 30]"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sc-sfmlearner-release_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--scales", type=int, default=4)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    import loss_functions as LF
    from scsfm_hip import config as hip_config, synth
    dev = torch.device("cuda:0")
    H, W = a.height, a.width
    d = synth.make_batch(a.batch, H, W, n_ref=2, seed=5, depth="smooth", num_scales=a.scales)
    cv = lambda t: t.to(dev).contiguous()
    tgt, refs, K = cv(d["tgt_img"]), [cv(r) for r in d["ref_imgs"]], cv(d["intrinsics"])
    hip_config.set_weight_hint(1.0, 0.5)
    lf = lambda t: cv(t).requires_grad_(True)
    td, rd = [lf(t) for t in d["tgt_depth"]], [[lf(t) for t in r] for r in d["ref_depths"]]
    ps, pi = [lf(p) for p in d["poses"]], [lf(p) for p in d["poses_inv"]]
    leaves = td + [t for r in rd for t in r] + ps + pi

    def step(materialise):
        up = (lambda t: F.interpolate(t, (H, W), mode="nearest") if t.shape[-1] != W else t) if materialise else (lambda t: t)
        photo, geom = LF.compute_photo_and_geometry_loss(tgt, refs, K, [up(t) for t in td],
                                                         [[up(t) for t in r] for r in rd], ps, pi, a.scales, 1, 1, 1, "zeros")
        smooth = LF.compute_smooth_loss(td, tgt, rd, refs)
        loss = photo + 0.1 * smooth + 0.5 * geom
        for t in leaves:
            t.grad = None
        loss.backward()
        return loss

    out = {}
    for name, mat in (("in_place", False), ("materialised", True)):
        for _ in range(5):
            loss = step(mat)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            step(mat)
        e1.record()
        torch.cuda.synchronize()
        out[name] = {"ms_per_step": round(e0.elapsed_time(e1) / a.steps, 4), "loss": float(loss.detach()),
                     "peak_alloc_MB": round(torch.cuda.max_memory_allocated() / 2**20, 1)}
        torch.cuda.reset_peak_memory_stats()
    out["config"] = {"batch": a.batch, "height": H, "width": W, "scales": a.scales, "n_ref": 2}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
