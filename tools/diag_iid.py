#!/usr/bin/env python3
"""How far are fp32 pose gradients from fp64 on iid inputs (independent depths 0.1 .. 100 per pixel, 4 x 256 x 832)?
Prints, per pose tensor, the per-row relative error of the HIP path and of the fp32 oracle (= the reference's own
arithmetic) against the fp64 oracle.  SEED=n selects the batch, SCSFM_HIP_LIB a library variant.  The row statistics
of tests/test_gpu_parity.py (POSE_RTOL_IID) come from this."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sc-sfmlearner-release_amd")]
import torch
import loss_functions as LF
from oracle import scsfm_oracle as O
from scsfm_hip import synth
dev = torch.device("cuda")
B, H, W, n_ref = 4, 256, 832, 2
SEED = int(os.environ.get("SEED", "17"))
d = synth.make_batch(B, H, W, n_ref=n_ref, seed=SEED, depth="iid", image="iid", dataset="kitti")
flags = (1, 1, 1, "zeros")
def run(device, fn_pg, fn_s, dtype=torch.float32):
    mv = lambda t: t.to(device=device, dtype=dtype).clone().requires_grad_(True)
    cv = lambda t: t.to(device=device, dtype=dtype)
    td = [mv(t) for t in d["tgt_depth"]]; rd = [[mv(t) for t in r] for r in d["ref_depths"]]
    ps, pi = [mv(p) for p in d["poses"]], [mv(p) for p in d["poses_inv"]]
    tgt, refs, K = cv(d["tgt_img"]), [cv(r) for r in d["ref_imgs"]], cv(d["intrinsics"])
    photo, geom = fn_pg(tgt, refs, K, td, rd, ps, pi, 1, *flags)
    smooth = fn_s(td, tgt, rd, refs)
    (photo + 0.1 * smooth + 0.5 * geom).backward()
    grads = [td[0].grad] + [r[0].grad for r in rd] + [p.grad for p in ps + pi]
    return [float(photo.detach()), float(geom.detach()), float(smooth.detach())], [g.detach().cpu().double() for g in grads]
vh, gh = run(dev, LF.compute_photo_and_geometry_loss, LF.compute_smooth_loss)
vo, go = run("cpu", O.photo_and_geometry_loss, O.smooth_loss)
v64, g64 = run("cpu", O.photo_and_geometry_loss, O.smooth_loss, torch.float64)
print("seed", SEED, "lib", os.environ.get("SCSFM_HIP_LIB", "tree"))
for i, (a, b, c) in enumerate(zip(gh, go, g64)):
    scale = float(c.abs().max())
    if i <= n_ref:
        for thr in (5e-3, 1e-3):
            print(i, "depth map: share > %.0e*scale: hip-vs-32 %.2e  hip-vs-64 %.2e  32-vs-64 %.2e" % (thr, ((a - b).abs() > thr * scale).double().mean(), ((a - c).abs() > thr * scale).double().mean(), ((b - c).abs() > thr * scale).double().mean()))
    else:
        rows = ((a - c).abs().max(dim=1).values / c.abs().max(dim=1).values)
        rows32 = ((b - c).abs().max(dim=1).values / c.abs().max(dim=1).values)
        print(i, "rows hip-64 rel:", [round(float(x), 4) for x in rows], " rows 32-64 rel:", [round(float(x), 4) for x in rows32])
        print(i, "pose: scale %.3f  max|hip-64| %.4f  max|32-64| %.4f  max|hip-32| %.4f" % (scale, float((a - c).abs().max()), float((b - c).abs().max()), float((a - b).abs().max())))
