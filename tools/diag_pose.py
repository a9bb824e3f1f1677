#!/usr/bin/env python3
"""Diagnostic (GPU box): where do the HIP fp32 depth gradients differ most from the fp64 oracle,
and what do those pixels look like (mask margins, clamps)?  Explains pose-gradient deviations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sc-sfmlearner-release_amd")]
import torch
import loss_functions as LF
from oracle import scsfm_oracle as O
from scsfm_hip import synth

dev = torch.device("cuda:0")
d = synth.make_batch(2, 128, 416, n_ref=2, seed=17, depth="smooth")
i = 1  # second reference: the pair whose pose gradient deviated
def grads(device, dtype, fn):
    mv = lambda t: t.to(device=device, dtype=dtype).clone().requires_grad_(True)
    cv = lambda t: t.to(device=device, dtype=dtype)
    td, rd, po = mv(d["tgt_depth"][0]), mv(d["ref_depths"][i][0]), mv(d["poses"][i])
    p, g = fn(cv(d["tgt_img"]), cv(d["ref_imgs"][i]), td, rd, po, cv(d["intrinsics"]), 1, 1, 0, "zeros")
    (p + 0.5 * g).backward()
    return float(p.detach()), float(g.detach()), td.grad.detach().cpu().double(), rd.grad.detach().cpu().double(), po.grad.detach().cpu().double()
ph, gh, tdh, rdh, poh = grads(dev, torch.float32, LF.compute_pairwise_loss)
p6, g6, td6, rd6, po6 = grads("cpu", torch.float64, O.pairwise_loss)
p3, g3, td3, rd3, po3 = grads("cpu", torch.float32, O.pairwise_loss)
print("photo", ph, p3, p6, "geom", gh, g3, g6)
print("pose hip ", poh)
print("pose o32 ", po3)
print("pose o64 ", po6)
di, dd, m = O.pairwise_maps(d["tgt_img"].double(), d["ref_imgs"][i].double(), d["tgt_depth"][0].double(), d["ref_depths"][i][0].double(),
                            d["poses"][i].double(), d["intrinsics"].double(), 1, 1, 0, "zeros")
w, v, pd, cd = O.inverse_warp2(d["ref_imgs"][i].double(), d["tgt_depth"][0].double(), d["ref_depths"][i][0].double(), d["poses"][i].double(), d["intrinsics"].double())
for name, a, b in (("hip", tdh, td6), ("o32", td3, td6)):
    e = (a - b).abs().reshape(-1)
    top = torch.topk(e, 8)
    print(f"--- {name}: depth-grad scale {b.abs().max():.3e}, sum|err| {e.sum():.3e}, top errors:")
    for val, idx in zip(top.values, top.indices):
        bb, rem = divmod(int(idx), 128 * 416); y, x = divmod(rem, 416)
        print(f"   b{bb} y{y} x{x} err {val:.3e} g64 {b.reshape(-1)[idx]:.3e} depth {float(d['tgt_depth'][0][bb,0,y,x]):.3f} valid {float(v[bb,0,y,x])} "
              f"cd {float(cd[bb,0,y,x]):.4f} pd {float(pd[bb,0,y,x]):.4f} dd {float(dd[bb,0,y,x]):.4f}")
