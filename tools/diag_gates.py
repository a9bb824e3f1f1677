#!/usr/bin/env python3
"""Root-causing the worst judged entries of tests/test_gpu_parity.py::test_depth_gradients_entrywise_away_from_the_gates.

    on the GPU box:   python tools/diag_gates.py dump [B] [depth]      -> gpurun_out/diag_gates_B_depth.npz (HIP fp32 gradients)
    anywhere (CPU):   python tools/diag_gates.py analyse [B] [depth]   -> the 20 worst judged entries of every depth-gradient
                      map with the pixel, what carried it (dense term of which pair / which target pixels scatter into
                      it), every gate's margin there, the tap fractions, the unscaled magnitude of the scattered terms
                      (>= 64 = direct atomics, else the fixed-point window) and the same entry's error in the
                      reference's own fp32 arithmetic -- as JSON (profiles/r05_iid_worst_entries.json)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sc-sfmlearner-release_amd"))
H, W, N_REF, SEED = 256, 832, 2, 29


def batch(B, depth):
    from scsfm_hip import synth
    return synth.make_batch(B, H, W, n_ref=N_REF, seed=SEED, depth=depth, image=synth.image_law(depth), dataset="kitti")


def dump(B, depth):
    import loss_functions as LF
    d = batch(B, depth)
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


def analyse(B, depth, top=20):
    from oracle import scsfm_oracle as O
    d = batch(B, depth)
    hip = np.load(os.path.join(ROOT, "gpurun_out", f"diag_gates_{B}_{depth}.npz"))
    gh = [torch.from_numpy(hip[f"g{i}"]).double()[:, 0] for i in range(1 + N_REF)]

    def run(dtype):
        mv = lambda t: t.to(dtype).clone().requires_grad_(True)
        cv = lambda t: t.to(dtype)
        td, rd = [mv(d["tgt_depth"][0])], [[mv(r[0])] for r in d["ref_depths"]]
        ps, pi = [mv(p) for p in d["poses"]], [mv(p) for p in d["poses_inv"]]
        photo, geom = O.photo_and_geometry_loss(cv(d["tgt_img"]), [cv(r) for r in d["ref_imgs"]], cv(d["intrinsics"]), td, rd, ps, pi,
                                                1, 1, 1, 1, "zeros")
        (photo + 0.5 * geom).backward()
        return [t.grad.detach().double()[:, 0] for t in td + [r[0] for r in rd]]

    g32, g64 = run(torch.float32), run(torch.float64)
    c = lambda t: t.double()
    frames = [(c(d["tgt_img"]), c(d["tgt_depth"][0]))] + [(c(d["ref_imgs"][i]), c(d["ref_depths"][i][0])) for i in range(N_REF)]
    K = c(d["intrinsics"])
    # the four pair-directions: (target map index, reference map index, pose)
    pairs = []
    for i in range(N_REF):
        pairs.append((0, 1 + i, c(d["poses"][i])))
        pairs.append((1 + i, 0, c(d["poses_inv"][i])))
    info = []
    unsafe = [torch.zeros(B, H, W, dtype=torch.bool) for _ in frames]
    for (ia, ib, pose) in pairs:
        (a_img, a_d), (b_img, b_d) = frames[ia], frames[ib]
        m = O.pairwise_gate_margins(a_img, b_img, a_d, b_d, pose, K, 1, 1, 1, "zeros")
        dense, scatter = O.unsafe_gradient_entries(m)
        unsafe[ia] |= dense
        unsafe[ib] |= scatter
        # per-pixel quantities of this pair for the report: sampling position, scattered term (unscaled), diff_depth
        a_dl = a_d.clone().requires_grad_(True)
        warped, valid, pd, cd = O.inverse_warp2(b_img, a_dl, b_d, pose, K, "zeros")
        pd.retain_grad()
        dimg, dd, mask = O.pairwise_maps(a_img, b_img, a_dl, b_d, pose, K, 1, 1, 1, "zeros")
        info.append({"ia": ia, "ib": ib, "m": m, "dd": dd.detach()[:, 0], "mask": mask.detach()[:, 0], "valid": valid[:, 0],
                     "Z": cd.detach()[:, 0], "Dp": pd.detach()[:, 0]})
    report = {"workload": {"B": B, "H": H, "W": W, "depth": depth, "seed": SEED}, "maps": []}
    for i in range(1 + N_REF):
        scale = float(g64[i].abs().max())
        eh, eo = (gh[i] - g64[i]).abs() / scale, (g32[i] - g64[i]).abs() / scale
        keep = ~unsafe[i]
        ehk = eh.clone()
        ehk[~keep] = -1
        idx = torch.topk(ehk.flatten(), top).indices
        rows = []
        for j in idx.tolist():
            b, y, x = j // (H * W), (j // W) % H, j % W
            row = {"b": b, "y": y, "x": x, "hip_err": float(eh[b, y, x]), "ref_fp32_err": float(eo[b, y, x]),
                   "value64": float(g64[i][b, y, x]) / scale, "as_target_of": [], "scattered_into_by": []}
            for q, inf in enumerate(info):
                m = inf["m"]
                if inf["ia"] == i:  # the dense term of this pixel
                    row["as_target_of"].append({"pair": q, "mask": float(inf["mask"][b, y, x]), "dd": float(inf["dd"][b, y, x]),
                                                "Z": float(inf["Z"][b, y, x]), "Dp": float(inf["Dp"][b, y, x]),
                                                "hard": bool(m["hard"][b, y, x]), "soft": bool(m["soft"][b, y, x]), "own": bool(m["own"][b, y, x])})
                if inf["ib"] == i:  # target pixels of pair q whose 2x2 block covers (y, x)
                    hit = ((m["ya"][b] == y) | (m["ya"][b] == y - 1)) & ((m["xa"][b] == x) | (m["xa"][b] == x - 1)) & (inf["mask"][b] > 0)
                    ys, xs = hit.nonzero(as_tuple=True)
                    srcs = []
                    for yy, xx in zip(ys.tolist()[:12], xs.tolist()[:12]):
                        Z, Dp = float(inf["Z"][b, yy, xx]), float(inf["Dp"][b, yy, xx])
                        srcs.append({"y": yy, "x": xx, "dd": float(inf["dd"][b, yy, xx]), "Z": Z, "Dp": Dp,
                                     "unscaled_2Z_over_sum2": 2 * Z / (Z + Dp) ** 2 if Z + Dp > 0 else None,
                                     "hard": bool(m["hard"][b, yy, xx]), "soft": bool(m["soft"][b, yy, xx])})
                    row["scattered_into_by"].append({"pair": q, "n_sources": int(hit.sum()), "sources": srcs})
            rows.append(row)
        q = lambda t: [float(torch.quantile(t[keep][::3], p)) for p in (0.5, 0.99, 0.999, 0.9999)]
        report["maps"].append({"map": i, "scale": scale, "judged_share": float(keep.double().mean()),
                               "hip_quantiles": q(eh), "ref_fp32_quantiles": q(eo),
                               "hip_max": float(eh[keep].max()), "ref_fp32_max": float(eo[keep].max()), "worst": rows})
    print(json.dumps(report))


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "dump"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    depth = sys.argv[3] if len(sys.argv) > 3 else "iid"
    dump(B, depth) if mode == "dump" else analyse(B, depth)
