#!/usr/bin/env python3
"""Dump the HIP fp32 depth gradients of a seeded full-size batch (on the GPU box) so that tests/test_gpu_parity.py's
entry-wise comparison against the fp64 oracle can be analysed off the box:   python tools/diag_gates.py [B] [depth]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sc-sfmlearner-release_amd"))


def main():
    import loss_functions as LF
    from scsfm_hip import synth
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    depth = sys.argv[2] if len(sys.argv) > 2 else "smooth"
    d = synth.make_batch(B, 256, 832, n_ref=2, seed=29, depth=depth, image="smooth" if depth == "smooth" else "iid", dataset="kitti")
    dev = "cuda:0"
    mv = lambda t: t.to(dev).clone().requires_grad_(True)
    cv = lambda t: t.to(dev)
    td, rd = [mv(d["tgt_depth"][0])], [[mv(r[0])] for r in d["ref_depths"]]
    ps, pi = [mv(p) for p in d["poses"]], [mv(p) for p in d["poses_inv"]]
    photo, geom = LF.compute_photo_and_geometry_loss(cv(d["tgt_img"]), [cv(r) for r in d["ref_imgs"]], cv(d["intrinsics"]), td, rd, ps,
                                                     pi, 1, 1, 1, 1, "zeros")
    (photo + 0.5 * geom).backward()
    out = os.path.join(ROOT, "gpurun_out", f"diag_gates_{B}_{depth}.npz")
    np.savez_compressed(out, photo=float(photo), geom=float(geom), **{f"g{i}": t.grad.cpu().numpy() for i, t in enumerate(td + [r[0] for r in rd])})
    print(out)


if __name__ == "__main__":
    main()
