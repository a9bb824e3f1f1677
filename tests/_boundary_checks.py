"""Checks of the reference's public `inverse_warp` names that round 1 left as stubs -- pixel2cam, cam2pixel,
cam2pixel2, legacy inverse_warp with quaternions -- shared by the CPU (hostsim) and the GPU test modules.  Each
function of the drop-in module `IW` runs on `dev` and is compared, values and gradients, with the oracle in fp64 /
fp32 on the CPU."""
import torch

from oracle import scsfm_oracle as O
from scsfm_hip import synth

REFERENCE_NAMES = {  # `from inverse_warp import *` of the reference (inverse_warp.py: every top-level name)
    "F", "cam2pixel", "cam2pixel2", "check_sizes", "division", "euler2mat", "inverse_warp", "inverse_warp2", "pixel2cam",
    "pixel_coords", "pose_vec2mat", "quat2mat", "set_id_grid", "torch",
}


def _rel(a, b):
    return float((a.detach().cpu().double() - b.detach().double()).abs().max() / (b.detach().double().abs().max() + 1e-300))


def run(IW, dev, dtype, tol):
    d = synth.make_batch(2, 40, 72, n_ref=1, seed=17)
    to = lambda t: t.detach().to(dtype).to(dev)
    leaf = lambda t: t.detach().to(dtype).to(dev).clone().requires_grad_(True)
    cleaf = lambda t: t.detach().to(dtype).clone().requires_grad_(True)
    depth, K, pose = d["tgt_depth"][0].squeeze(1).to(dtype), d["intrinsics"].to(dtype), (d["poses"][0] * 3).to(dtype)
    Kinv = O.inv3x3(K, "explicit")
    # ---- pixel2cam -------------------------------------------------------------------------------------
    dl, dc = leaf(depth), cleaf(depth)
    cam = IW.pixel2cam(dl, to(Kinv))
    cam_o = O.back_project(dc, Kinv)
    assert tuple(cam.shape) == (2, 3, 40, 72) and _rel(cam, cam_o) <= tol
    w = torch.cos(torch.arange(cam_o.numel(), dtype=dtype).reshape(cam_o.shape))
    (cam * w.to(dev)).sum().backward(); (cam_o * w).sum().backward()
    assert _rel(dl.grad, dc.grad) <= tol
    # ---- cam2pixel2 (zeros: overwrite; border: none) and cam2pixel, gradients to cam / rot / tr --------------
    P = K @ O.pose_vec2mat(pose)
    rot, tr = P[:, :, :3].contiguous(), P[:, :, 3:].contiguous()
    for pad in ("zeros", "border"):
        cl, rl, tl = (leaf(t) for t in (cam_o, rot, tr))
        co, ro, tro = (cleaf(t) for t in (cam_o, rot, tr))
        grid, z = IW.cam2pixel2(cl, rl, tl, pad)
        xn, yn, zo = O.project(co, ro, tro, pad)
        assert tuple(grid.shape) == (2, 40, 72, 2) and tuple(z.shape) == (2, 1, 40, 72)
        go = torch.stack([xn, yn], dim=-1).reshape(2, 40, 72, 2)
        # coordinates that round across +-1 differ by the overwrite: compare where both agree on it
        same = ((grid.detach().cpu() == 2) == (go.detach() == 2)).all(dim=-1)
        assert same.double().mean() >= 0.999
        assert float(((grid.detach().cpu() - go.detach()).abs().max(dim=-1)[0] * same).max()) <= tol * 4
        assert _rel(z, zo.reshape(2, 1, 40, 72)) <= tol
        wg = torch.sin(torch.arange(go.numel(), dtype=dtype).reshape(go.shape)) * same.unsqueeze(-1)
        wz = torch.cos(torch.arange(z.numel(), dtype=dtype).reshape(z.shape))
        ((grid * wg.to(dev)).sum() + (z * wz.to(dev)).sum()).backward()
        ((go * wg).sum() + (zo.reshape(2, 1, 40, 72) * wz).sum()).backward()
        assert _rel(cl.grad, co.grad) <= 50 * tol and _rel(rl.grad, ro.grad) <= 50 * tol and _rel(tl.grad, tro.grad) <= 50 * tol
    g1 = IW.cam2pixel(to(cam_o.detach()), to(rot), to(tr), "zeros")       # legacy: never overwrites
    xn, yn, _ = O.project(cam_o.detach(), rot, tr, "border")
    assert _rel(g1, torch.stack([xn, yn], dim=-1).reshape(2, 40, 72, 2)) <= tol * 4
    g2 = IW.cam2pixel(to(cam_o.detach()), None, None, "zeros")            # identity rotation, no translation
    xn, yn, _ = O.project(cam_o.detach(), torch.eye(3, dtype=dtype).expand(2, 3, 3), torch.zeros(2, 3, 1, dtype=dtype), "border")
    assert _rel(g2, torch.stack([xn, yn], dim=-1).reshape(2, 40, 72, 2)) <= tol * 4
    # ---- legacy inverse_warp with a quaternion rotation (inverse_warp.py:157-191) ---------------------------
    img = d["ref_imgs"][0].to(dtype)
    for mode in ("euler", "quat"):
        pl, po = leaf(pose), cleaf(pose)
        dl2, do2 = leaf(depth), cleaf(depth)
        wimg, valid = IW.inverse_warp(to(img), dl2, pl, to(K), mode, "zeros")
        Pq = K @ O.pose_vec2mat(po, mode)
        xn, yn, _ = O.project(O.back_project(do2, Kinv), Pq[:, :, :3], Pq[:, :, 3:], "border")
        wo = O.bilinear_sample(img, xn, yn, "zeros", "explicit")
        vo = torch.maximum(xn.abs(), yn.abs()) <= 1
        assert valid.dtype == torch.bool and (valid.cpu() != vo.reshape(valid.shape)).double().mean() <= 1e-3
        assert _rel(wimg, wo) <= 200 * tol
        wt = torch.cos(0.01 * torch.arange(wo.numel(), dtype=dtype).reshape(wo.shape))
        (wimg * wt.to(dev)).sum().backward(); (wo * wt).sum().backward()
        assert _rel(pl.grad, po.grad) <= 2e3 * tol and _rel(dl2.grad, do2.grad) <= 2e3 * tol, (mode, _rel(pl.grad, po.grad))
