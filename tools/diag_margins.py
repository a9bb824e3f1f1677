#!/usr/bin/env python3
"""How the entry-wise gradient judgement (tests/test_gpu_parity.py::test_depth_gradients_entrywise_away_from_the_gates)
moves with the slope-aware margin `eps_slope_px` of oracle.pairwise_gate_margins: for each of the test's four cases and
each of a list of margins, the judged share and HIP-worst / reference-fp32-worst of every depth-gradient map, plus the
quantile ratios.  A margin argument of the form slope:px:val varies eps_px and eps_val as well.  One GPU run per case (the gradients do not depend on the margin); the margins are evaluated on the host
in fp64.  Prints one JSON line per (case, margin).  Run on the GPU box:  python tools/diag_margins.py 5e-4 2.5e-4 1.2e-4 0"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sc-sfmlearner-release_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import loss_functions as LF  # noqa: E402
from oracle import scsfm_oracle as O  # noqa: E402
from scsfm_hip import synth  # noqa: E402


def unsafe_maps(d, n_ref, pad, eps_slope_px, eps_px=2e-3, eps_val=2e-4):
    c = lambda t: t.double()
    ti, K = c(d["tgt_img"]), c(d["intrinsics"])
    unsafe = [torch.zeros(ti.shape[0], ti.shape[2], ti.shape[3], dtype=torch.bool) for _ in range(1 + n_ref)]
    for i in range(n_ref):
        ri, td, rd = c(d["ref_imgs"][i]), c(d["tgt_depth"][0]), c(d["ref_depths"][i][0])
        for (a_img, b_img, a_d, b_d, pose, ia, ib) in ((ti, ri, td, rd, c(d["poses"][i]), 0, 1 + i),
                                                       (ri, ti, rd, td, c(d["poses_inv"][i]), 1 + i, 0)):
            m = O.pairwise_gate_margins(a_img, b_img, a_d, b_d, pose, K, 1, 1, 1, pad, eps_px=eps_px, eps_val=eps_val, eps_slope_px=eps_slope_px)
            dense, scatter = O.unsafe_gradient_entries(m)
            unsafe[ia] |= dense
            unsafe[ib] |= scatter
    return unsafe


def main(margins):
    dev = torch.device("cuda:0")
    H, W, n_ref = 256, 832, 2
    for B, depth, pad in [(12, "smooth", "zeros"), (4, "iid", "zeros"), (4, "smooth", "border"), (4, "scene", "zeros")]:
        d = synth.make_batch(B, H, W, n_ref=n_ref, seed=29, depth=depth, image=synth.image_law(depth), dataset="kitti")
        flags = (1, 1, 1, pad)

        def run(device, fn, dtype):
            mv = lambda t: t.to(device=device, dtype=dtype).clone().requires_grad_(True)
            cv = lambda t: t.to(device=device, dtype=dtype)
            td, rd = [mv(d["tgt_depth"][0])], [[mv(r[0])] for r in d["ref_depths"]]
            ps, pi = [mv(p) for p in d["poses"]], [mv(p) for p in d["poses_inv"]]
            photo, geom = fn(cv(d["tgt_img"]), [cv(r) for r in d["ref_imgs"]], cv(d["intrinsics"]), td, rd, ps, pi, 1, *flags)
            (photo + 0.5 * geom).backward()
            return [g.grad.detach().cpu().double() for g in td + [r[0] for r in rd]]

        gh = run(dev, LF.compute_photo_and_geometry_loss, torch.float32)
        g32 = run("cpu", O.photo_and_geometry_loss, torch.float32)
        g64 = run("cpu", O.photo_and_geometry_loss, torch.float64)
        for eps in margins:
            # a margin is either the slope margin alone, or "slope:px:val" (the other two margins as well)
            trip = [float(x) for x in str(eps).split(":")]
            unsafe = unsafe_maps(d, n_ref, pad, *trip)
            maps = []
            for a, o, c, u in zip(gh, g32, g64, unsafe):
                a, o, c = a[:, 0], o[:, 0], c[:, 0]
                keep = ~u
                scale = float(c.abs().max())
                eh, eo = ((a - c).abs() / scale)[keep], ((o - c).abs() / scale)[keep]
                q = lambda t: [float(torch.quantile(t[::3], p)) for p in (0.5, 0.99, 0.999, 0.9999)]
                qh, qo = q(eh), q(eo)
                maps.append({"judged": round(float(keep.double().mean()), 4), "worst_hip": float(eh.max()), "worst_ref32": float(eo.max()),
                             "ratio": round(float(eh.max()) / max(float(eo.max()), 1e-30), 2),
                             "quantile_ratios": [round(x / max(y, 1e-30), 2) for x, y in zip(qh, qo)]})
            print(json.dumps({"case": f"{depth}/{pad} B={B}", "eps_slope_px": eps, "maps": maps}), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or ["5e-4", "2.5e-4", "1.2e-4", "0"])
