"""CPU oracle for the SC-SfMLearner warp + loss hot path.  TEST INFRASTRUCTURE ONLY.

This file is a restatement, in plain eager PyTorch on the CPU, of the algorithm that the
reference implements in ``inverse_warp.py`` and ``loss_functions.py``.  It is *not* the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path (``sc-sfmlearner-release_amd/``) never
imports it and has no CPU fallback.

Pinning
-------
The reference ships no tests / golden vectors for this path, so this oracle is
pinned against the reference *itself*: ``oracle/make_golden.py`` imports the unmodified reference
modules from ``/root/reference`` in the build container, runs them on seeded inputs and commits the
outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` asserts that this restatement
reproduces every one of those fixtures (fp32, forward values and autograd gradients), and
``tests/test_oracle_vs_reference.py`` re-checks live whenever ``/root/reference`` is mounted.

Third-party arithmetic
----------------------
The reference delegates bilinear sampling, 3x3 pooling, reflection padding, 3x3 inversion and the
small matrix products to PyTorch ATen (``requirements.txt:1`` pins only ``torch>=1.5.1``; the
goldens were produced with torch 2.10.0+rocm7.0 CPU kernels).  Those algorithms are restated here
explicitly (``impl='explicit'``: gathers, shifted-slice sums, closed-form inverse) so that the HIP
kernels can be compared term by term; ``impl='aten'`` calls the very same ATen entry points as the
reference (``F.grid_sample`` ``inverse_warp.py:262,267``; ``AvgPool2d``/``ReflectionPad2d``
``loss_functions.py:17-23``; ``Tensor.inverse`` ``inverse_warp.py:253``) and is bit-identical to
the reference on CPU.

All functions work in the dtype of their inputs (fp32 or fp64).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# SSIM constants, loss_functions.py:25-26
SSIM_C1 = 0.01 ** 2
SSIM_C2 = 0.03 ** 2
# mean_on_mask gate, loss_functions.py:125
MASK_COUNT_GATE = 10000
# projection depth clamp, inverse_warp.py:211
Z_MIN = 1e-3


# --------------------------------------------------------------------------------------------
# SE(3) pose  (inverse_warp.py:77-154)
# --------------------------------------------------------------------------------------------
def rot_from_euler(angle: torch.Tensor) -> torch.Tensor:
    """[B,3] (rx,ry,rz) -> R = Rx(rx) @ Ry(ry) @ Rz(rz), [B,3,3].  inverse_warp.py:77-112."""
    rx, ry, rz = angle[:, 0], angle[:, 1], angle[:, 2]
    o = torch.zeros_like(rx)
    l = torch.ones_like(rx)

    def m(rows):
        return torch.stack([torch.stack(r, dim=1) for r in rows], dim=1)

    cz, sz = torch.cos(rz), torch.sin(rz)
    cy, sy = torch.cos(ry), torch.sin(ry)
    cx, sx = torch.cos(rx), torch.sin(rx)
    Rz = m([[cz, -sz, o], [sz, cz, o], [o, o, l]])
    Ry = m([[cy, o, sy], [o, l, o], [-sy, o, cy]])
    Rx = m([[l, o, o], [o, cx, -sx], [o, sx, cx]])
    return Rx @ Ry @ Rz


def rot_from_quat(q3: torch.Tensor) -> torch.Tensor:
    """[B,3] (x,y,z of a quaternion whose w is 1 before normalisation) -> [B,3,3].
    inverse_warp.py:115-136."""
    q = torch.cat([torch.ones_like(q3[:, :1]), q3], dim=1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [
        [w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z],
        [2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x],
        [2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z],
    ]
    return torch.stack([torch.stack(r, dim=1) for r in rows], dim=1)


def pose_vec2mat(vec: torch.Tensor, rotation_mode: str = "euler") -> torch.Tensor:
    """[B,6] (tx,ty,tz,rx,ry,rz) -> [R|t], [B,3,4].  inverse_warp.py:139-154."""
    if rotation_mode == "euler":
        R = rot_from_euler(vec[:, 3:])
    elif rotation_mode == "quat":
        R = rot_from_quat(vec[:, 3:])
    else:
        raise ValueError(rotation_mode)
    return torch.cat([R, vec[:, :3].unsqueeze(-1)], dim=2)


# --------------------------------------------------------------------------------------------
# Geometry  (inverse_warp.py:29-44, 194-227)
# --------------------------------------------------------------------------------------------
def inv3x3(K: torch.Tensor, impl: str) -> torch.Tensor:
    """inverse_warp.py:253 calls ``intrinsics.inverse()`` (ATen ``linalg_inv_ex``, LU).  The
    explicit form is the adjugate / determinant, evaluated in fp64 and rounded once."""
    if impl == "aten":
        return K.inverse()
    Kd = K.double()
    a, b, c = Kd[:, 0, 0], Kd[:, 0, 1], Kd[:, 0, 2]
    d, e, f = Kd[:, 1, 0], Kd[:, 1, 1], Kd[:, 1, 2]
    g, h, i = Kd[:, 2, 0], Kd[:, 2, 1], Kd[:, 2, 2]
    A = e * i - f * h
    B_ = -(d * i - f * g)
    C = d * h - e * g
    det = a * A + b * B_ + c * C
    adj = torch.stack([
        torch.stack([A, -(b * i - c * h), b * f - c * e], dim=1),
        torch.stack([B_, a * i - c * g, -(a * f - c * d)], dim=1),
        torch.stack([C, -(a * h - b * g), a * e - b * d], dim=1),
    ], dim=1)
    return (adj / det[:, None, None]).to(K.dtype)


def back_project(depth: torch.Tensor, Kinv: torch.Tensor) -> torch.Tensor:
    """depth [B,H,W] -> camera points [B,3,H,W]: (Kinv @ [u,v,1]) * depth.
    inverse_warp.py:8-17 (pixel grid: x = column j, y = row i) and :29-44."""
    B, H, W = depth.shape
    v, u = torch.meshgrid(torch.arange(H, dtype=depth.dtype), torch.arange(W, dtype=depth.dtype),
                          indexing="ij")
    pix = torch.stack([u, v, torch.ones_like(u)], dim=0).reshape(1, 3, -1).expand(B, 3, -1)
    rays = (Kinv @ pix).reshape(B, 3, H, W)
    return rays * depth.unsqueeze(1)


def project(cam: torch.Tensor, A: torch.Tensor, c: torch.Tensor, padding_mode: str):
    """cam [B,3,H,W], A [B,3,3], c [B,3,1] -> normalised coords xn, yn [B,H,W] and the computed
    depth Z [B,1,H,W].  inverse_warp.py:194-227 (cam2pixel2)."""
    B, _, H, W = cam.shape
    p = A @ cam.reshape(B, 3, -1) + c
    X, Y = p[:, 0], p[:, 1]
    Z = p[:, 2].clamp(min=Z_MIN)
    xn = 2 * (X / Z) / (W - 1) - 1
    yn = 2 * (Y / Z) / (H - 1) - 1
    if padding_mode == "zeros":
        # out-of-range coordinates are overwritten with the constant 2 (x and y independently);
        # the overwrite is in place on a detached mask, so no gradient reaches those entries.
        xn = torch.where((xn > 1) | (xn < -1), torch.full_like(xn, 2.0), xn)
        yn = torch.where((yn > 1) | (yn < -1), torch.full_like(yn, 2.0), yn)
    return xn.reshape(B, H, W), yn.reshape(B, H, W), Z.reshape(B, 1, H, W)


def bilinear_sample(src: torch.Tensor, xn: torch.Tensor, yn: torch.Tensor, padding_mode: str,
                    impl: str) -> torch.Tensor:
    """Bilinear sampling of src [B,C,H,W] at normalised coords ([B,H,W] each),
    ``align_corners=False``.  Restates ATen ``grid_sampler_2d`` as called at
    inverse_warp.py:262,267:

        ix = ((xn + 1) * W - 1) / 2 ;  x0 = floor(ix) ;  weights (x0+1-ix), (ix-x0)
        zeros : taps outside [0,W)x[0,H) contribute 0
        border: ix is first clipped to [0, W-1] (zero gradient where clipped or on the bound)
    """
    B, C, H, W = src.shape
    if impl == "aten":
        return F.grid_sample(src, torch.stack([xn, yn], dim=-1), padding_mode=padding_mode,
                             align_corners=False)
    ix = ((xn + 1) * W - 1) / 2
    iy = ((yn + 1) * H - 1) / 2
    if padding_mode == "border":
        ix = torch.where((ix > 0) & (ix < W - 1), ix, ix.detach().clamp(0, W - 1))
        iy = torch.where((iy > 0) & (iy < H - 1), iy, iy.detach().clamp(0, H - 1))
    x0 = ix.detach().floor()
    y0 = iy.detach().floor()
    wx1 = ix - x0
    wx0 = 1 - wx1
    wy1 = iy - y0
    wy0 = 1 - wy1
    flat = src.reshape(B, C, H * W)
    out = 0
    for dy, wy in ((0, wy0), (1, wy1)):
        for dx, wx in ((0, wx0), (1, wx1)):
            xi = x0 + dx
            yi = y0 + dy
            inb = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
            lin = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long().reshape(B, 1, -1)
            val = flat.gather(2, lin.expand(B, C, -1)).reshape(B, C, *xn.shape[1:])
            out = out + val * (wy * wx * inb.to(src.dtype)).unsqueeze(1)
    return out


def inverse_warp2(img, depth, ref_depth, pose, intrinsics, padding_mode="zeros", impl="explicit"):
    """inverse_warp.py:230-269.  Returns (projected_img, valid_mask, projected_depth,
    computed_depth)."""
    Kinv = inv3x3(intrinsics, impl)
    cam = back_project(depth.squeeze(1), Kinv)
    P = intrinsics @ pose_vec2mat(pose)
    xn, yn, Z = project(cam, P[:, :, :3], P[:, :, 3:], padding_mode)
    warped = bilinear_sample(img, xn, yn, padding_mode, impl)
    valid = (torch.maximum(xn.abs(), yn.abs()) <= 1).to(img.dtype).unsqueeze(1)
    proj_depth = bilinear_sample(ref_depth, xn, yn, padding_mode, impl)
    return warped, valid, proj_depth, Z


# --------------------------------------------------------------------------------------------
# Photometric / geometric terms  (loss_functions.py:11-45, 95-129)
# --------------------------------------------------------------------------------------------
def _reflect_pad1(x):
    """ReflectionPad2d(1): mirror without repeating the edge sample (pad[-1] = x[1])."""
    x = torch.cat([x[..., 1:2], x, x[..., -2:-1]], dim=-1)
    return torch.cat([x[..., 1:2, :], x, x[..., -2:-1, :]], dim=-2)


def _box3(xp):
    """3x3 mean, stride 1, no padding (AvgPool2d(3, 1)) of an already padded map."""
    H, W = xp.shape[-2] - 2, xp.shape[-1] - 2
    acc = 0
    for dy in range(3):
        for dx in range(3):
            acc = acc + xp[..., dy:dy + H, dx:dx + W]
    return acc / 9


def ssim_map(x, y, impl="explicit"):
    """clamp((1 - SSIM(x, y)) / 2, 0, 1) with 3x3 reflect-padded means.  loss_functions.py:28-42."""
    if impl == "aten":
        pad = torch.nn.ReflectionPad2d(1)
        pool = torch.nn.AvgPool2d(3, 1)
        xp, yp = pad(x), pad(y)
    else:
        pool = _box3
        xp, yp = _reflect_pad1(x), _reflect_pad1(y)
    mu_x, mu_y = pool(xp), pool(yp)
    sig_x = pool(xp ** 2) - mu_x ** 2
    sig_y = pool(yp ** 2) - mu_y ** 2
    sig_xy = pool(xp * yp) - mu_x * mu_y
    n = (2 * mu_x * mu_y + SSIM_C1) * (2 * sig_xy + SSIM_C2)
    d = (mu_x ** 2 + mu_y ** 2 + SSIM_C1) * (sig_x + sig_y + SSIM_C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def mean_on_mask(diff, valid_mask):
    """sum(diff*m)/sum(m) over the whole batch if sum(m expanded to diff's shape) > 10000,
    else the constant 0 (no grad).  loss_functions.py:123-129."""
    mask = valid_mask.expand_as(diff)
    if mask.sum() > MASK_COUNT_GATE:
        return (diff * mask).sum() / mask.sum()
    return torch.zeros((), dtype=diff.dtype)


def pairwise_maps(tgt_img, ref_img, tgt_depth, ref_depth, pose, K, with_ssim, with_mask,
                  with_auto_mask, padding_mode, impl="explicit"):
    """The per-pixel maps of compute_pairwise_loss (loss_functions.py:95-113): returns
    (diff_img [B,3,H,W], diff_depth [B,1,H,W], mask [B,1,H,W])."""
    warped, valid, proj_depth, comp_depth = inverse_warp2(ref_img, tgt_depth, ref_depth, pose, K,
                                                          padding_mode, impl)
    diff_img = (tgt_img - warped).abs().clamp(0, 1)
    diff_depth = ((comp_depth - proj_depth).abs() / (comp_depth + proj_depth)).clamp(0, 1)
    mask = valid
    if with_auto_mask:
        ident = (tgt_img - ref_img).abs().mean(dim=1, keepdim=True)
        mask = (diff_img.mean(dim=1, keepdim=True) < ident).to(valid.dtype) * valid
    if with_ssim:
        diff_img = 0.15 * diff_img + 0.85 * ssim_map(tgt_img, warped, impl)
    if with_mask:
        diff_img = diff_img * (1 - diff_depth)
    return diff_img, diff_depth, mask


def pairwise_loss(tgt_img, ref_img, tgt_depth, ref_depth, pose, K, with_ssim, with_mask,
                  with_auto_mask, padding_mode, impl="explicit"):
    """compute_pairwise_loss, loss_functions.py:95-119 -> (photo, geometry) scalars."""
    diff_img, diff_depth, mask = pairwise_maps(tgt_img, ref_img, tgt_depth, ref_depth, pose, K,
                                               with_ssim, with_mask, with_auto_mask, padding_mode,
                                               impl)
    return mean_on_mask(diff_img, mask), mean_on_mask(diff_depth, mask)


def photo_and_geometry_loss(tgt_img, ref_imgs, K, tgt_depth, ref_depths, poses, poses_inv,
                            max_scales, with_ssim, with_mask, with_auto_mask, padding_mode,
                            impl="explicit"):
    """compute_photo_and_geometry_loss, loss_functions.py:50-92: plain sums over refs, scales and
    both directions; scale s > 0 depths are nearest-upsampled to the image size (:81-82)."""
    photo = 0
    geom = 0
    H, W = tgt_img.shape[-2:]
    n_scales = min(len(tgt_depth), max_scales)
    for ref_img, ref_depth, pose, pose_inv in zip(ref_imgs, ref_depths, poses, poses_inv):
        for s in range(n_scales):
            dt, dr = tgt_depth[s], ref_depth[s]
            if s > 0:
                dt = F.interpolate(dt, (H, W), mode="nearest")
                dr = F.interpolate(dr, (H, W), mode="nearest")
            p1, g1 = pairwise_loss(tgt_img, ref_img, dt, dr, pose, K, with_ssim, with_mask,
                                   with_auto_mask, padding_mode, impl)
            p2, g2 = pairwise_loss(ref_img, tgt_img, dr, dt, pose_inv, K, with_ssim, with_mask,
                                   with_auto_mask, padding_mode, impl)
            photo = photo + p1 + p2
            geom = geom + g1 + g2
    return photo, geom


# --------------------------------------------------------------------------------------------
# Edge-aware smoothness  (loss_functions.py:132-159)
# --------------------------------------------------------------------------------------------
def smooth_term(depth, img):
    """One frame: mean-normalised depth, first differences weighted by exp(-mean_c |d img|)."""
    mean = depth.mean(2, True).mean(3, True)
    d = depth / (mean + 1e-7)
    gx = (d[..., :, :-1] - d[..., :, 1:]).abs()
    gy = (d[..., :-1, :] - d[..., 1:, :]).abs()
    wx = torch.exp(-(img[..., :, :-1] - img[..., :, 1:]).abs().mean(1, keepdim=True))
    wy = torch.exp(-(img[..., :-1, :] - img[..., 1:, :]).abs().mean(1, keepdim=True))
    return (gx * wx).mean() + (gy * wy).mean()


def smooth_loss(tgt_depth, tgt_img, ref_depths, ref_imgs):
    """compute_smooth_loss, loss_functions.py:132-159: scale 0 of the target and of every ref."""
    loss = smooth_term(tgt_depth[0], tgt_img)
    for rd, ri in zip(ref_depths, ref_imgs):
        loss = loss + smooth_term(rd[0], ri)
    return loss


# --------------------------------------------------------------------------------------------
# Depth metrics  (loss_functions.py:162-205)
# --------------------------------------------------------------------------------------------
@torch.no_grad()
def depth_errors(gt, pred, dataset):
    """compute_errors: [abs_diff, abs_rel, sq_rel, a1, a2, a3] averaged over the batch.
    Crop fractions / depth caps: kitti :173-178, nyu :180-185; median scaling :195."""
    B, h, w = gt.shape
    if dataset == "kitti":
        y1, y2 = int(0.40810811 * h), int(0.99189189 * h)
        x1, x2 = int(0.03594771 * w), int(0.96405229 * w)
        cap = 80
    elif dataset == "nyu":
        y1, y2 = int(0.09375 * h), int(0.98125 * h)
        x1, x2 = int(0.0640625 * w), int(0.9390625 * w)
        cap = 10
    else:
        raise ValueError(dataset)
    crop = torch.zeros(h, w, dtype=torch.bool)
    crop[y1:y2, x1:x2] = True
    tot = [0.0] * 6
    for g, p in zip(gt, pred):
        ok = (g > 0.1) & (g < cap) & crop
        g = g[ok]
        p = p[ok].clamp(1e-3, cap)
        p = p * torch.median(g) / torch.median(p)
        r = torch.max(g / p, p / g)
        vals = [(g - p).abs().mean(), ((g - p).abs() / g).mean(), ((g - p) ** 2 / g).mean(),
                (r < 1.25).float().mean(), (r < 1.25 ** 2).float().mean(),
                (r < 1.25 ** 3).float().mean()]
        tot = [t + v for t, v in zip(tot, vals)]
    return [float(t) / B for t in tot]


# --------------------------------------------------------------------------------------------
# Where an fp32 evaluation may legitimately differ from an fp64 one by more than round-off
# --------------------------------------------------------------------------------------------
def _sampling_slopes(src, xn, yn, padding_mode, delta_px=1e-3):
    """|d sampled / d ix| + |d sampled / d iy| of bilinear_sample(src) at (xn, yn), per channel [B,C,H,W]: one-sided
    differences over delta_px pixels (fp64), i.e. the slope of the bilinear patch the position lies in."""
    _, _, H, W = src.shape
    base = bilinear_sample(src, xn, yn, padding_mode, "explicit")
    dx = bilinear_sample(src, xn + 2 * delta_px / W, yn, padding_mode, "explicit")
    dy = bilinear_sample(src, xn, yn + 2 * delta_px / H, padding_mode, "explicit")
    return ((dx - base).abs() + (dy - base).abs()) / delta_px


def slope_margin_px(H, W):
    """The sampling-position uncertainty the slope-aware gate margins allow for: TWO units in the last place of the largest
    pixel coordinate of the image in fp32 (W = 832: 2 x 2^-14 = 1.2e-4 px; W = 320: 6.1e-5 px) -- what two fp32
    evaluations of the projection (the reference's chain of ATen ops, the kernels' fused form) can differ by.  Round 5
    used a constant 5e-4 px, ~8 ulp (advisor finding: wider than the error it stands for); round 6 measured the judgement
    over 5e-4, 2.5e-4, 1.2e-4, 6e-5 and 0 px on the hardware (tools/diag_margins.py, profiles/r06_margin_sensitivity.json):
    every worst-entry ratio of the twelve judged maps is IDENTICAL from 5e-4 down to 6e-5 (one ulp) and only a zero margin
    lets the flipped gates back in (iid: 4.8 / 3.6 / 6.6 x), while the judged share rises from 0.81-0.96 to 0.92-0.97 --
    so the margin was narrowed to two ulp: more entries judged under unchanged bounds.  The same sweep over the constant
    value margin eps_val (2e-4, 1e-4, 5e-5, 0: ratios identical down to 5e-5) halved it to 1e-4; the tap-switch margin eps_px
    stays at 2e-3 (1e-3 changes nothing, 5e-4 moves one iid map from 1.19 to 1.56 x)."""
    m = max(int(H), int(W)) - 1
    return 2.0 * 2.0 ** (math.floor(math.log2(max(m, 1))) - 23)


def pairwise_gate_margins(tgt_img, ref_img, tgt_depth, ref_depth, pose, K, with_ssim, with_mask, with_auto_mask,
                          padding_mode, eps_px=2e-3, eps_val=1e-4, eps_slope_px=None):
    """The path is full of discontinuous gates (inverse_warp.py:219-224,264; loss_functions.py:99,101,104-105; the clamps
    of the SSIM module :42; the tap switch of grid_sample).  A pixel whose gate is decided by less than fp32 round-off
    can come out on the other side in ANY fp32 evaluation, the reference's own included, and then differs by its full
    value.  This returns, per target pixel of one pair-direction (evaluate in fp64):

        hard  [B,H,W] bool : the warped VALUE is discontinuous here (valid / overwrite gate, Z' clamp): it reaches the
                             SSIM statistics of the 3x3 neighbours and through them the gradients of the 5x5
        soft  [B,H,W] bool : a coefficient of this pixel is discontinuous (clamps of |It - Iw|, sign at 0, auto-mask
                             comparison, SSIM clamp, depth-inconsistency sign / clamp): 3x3 of gradients
        own   [B,H,W] bool : only this pixel's own dense gradient is discontinuous (sampling position within eps of a
                             tap switch: the bilinear slope changes, values and scatter weights do not)
        xa, ya [B,H,W] long: north-west tap of the pixel (clamped into the image), for the scatter footprint

    Round 5: the margins of the VALUE gates (clamps and sign of |It - Iw|, the auto-mask comparison, sign and clamp of the
    depth inconsistency) grow with the local slope of what is sampled: a sampling position that differs by a few ulp of
    the coordinate (6e-5 px at x = 800; rcp against division, the fused projection) moves the sampled value by slope x
    that difference, and on iid inputs -- depth steps of up to 100 and colour steps of up to 4 between neighbouring
    texels -- that is 10 .. 1000 times eps_val.  The 20 worst judged entries of round 4's iid case were exactly such
    pixels (tools/diag_gates.py, profiles/r05_iid_worst_entries.json): |Z - D_p| / (Z + D_p) of 3e-4 .. 2e-3 with a
    depth slope of tens per pixel, auto-mask comparisons decided by 1e-3 with colour slopes of 2 .. 4 per pixel."""
    assert tgt_img.dtype == torch.float64
    B, _, H, W = tgt_img.shape
    if eps_slope_px is None:  # (round 6: two ulp of the largest coordinate instead of round 5's constant 5e-4 px)
        eps_slope_px = slope_margin_px(H, W)
    Kinv = inv3x3(K, "explicit")
    cam = back_project(tgt_depth.squeeze(1), Kinv)
    P = K @ pose_vec2mat(pose)
    p = P[:, :, :3] @ cam.reshape(B, 3, -1) + P[:, :, 3:]
    X, Y, Zr = p[:, 0].reshape(B, H, W), p[:, 1].reshape(B, H, W), p[:, 2].reshape(B, H, W)
    Z = Zr.clamp(min=Z_MIN)
    xn = 2 * (X / Z) / (W - 1) - 1
    yn = 2 * (Y / Z) / (H - 1) - 1
    # margins in pixels of the sampling position
    hard = ((xn.abs() - 1).abs() * (W / 2) < eps_px) | ((yn.abs() - 1).abs() * (H / 2) < eps_px)
    hard |= (Zr - Z_MIN).abs() < 1e-5
    ix = ((xn + 1) * W - 1) / 2
    iy = ((yn + 1) * H - 1) / 2
    if padding_mode == "border":
        hard |= (ix.abs() < eps_px) | ((ix - (W - 1)).abs() < eps_px) | (iy.abs() < eps_px) | ((iy - (H - 1)).abs() < eps_px)
        # (a clipped coordinate sits exactly on the border and has no gradient: nothing switches there)
        inside = (ix > 0) & (ix < W - 1) & (iy > 0) & (iy < H - 1)
        ix, iy = ix.clamp(0, W - 1), iy.clamp(0, H - 1)
    else:
        inside = (xn.abs() <= 1) & (yn.abs() <= 1)
    fx, fy = ix - ix.floor(), iy - iy.floor()
    own = inside & ((torch.minimum(fx, 1 - fx) < eps_px) | (torch.minimum(fy, 1 - fy) < eps_px))
    warped, valid, proj_depth, comp_depth = inverse_warp2(ref_img, tgt_depth, ref_depth, pose, K, padding_mode, "explicit")
    # how far the sampled colours / depth move per pixel of sampling-position error (0 where nothing is sampled)
    xs, ys, _ = project(back_project(tgt_depth.squeeze(1), Kinv), P[:, :, :3], P[:, :, 3:], padding_mode)
    slope_img = _sampling_slopes(ref_img, xs, ys, padding_mode) * eps_slope_px       # [B,3,H,W]
    slope_dep = _sampling_slopes(ref_depth, xs, ys, padding_mode)[:, 0] * eps_slope_px  # [B,H,W]
    d = tgt_img - warped
    ev = eps_val + slope_img
    soft = ((d.abs() < ev) | ((d.abs() - 1).abs() < ev)).any(dim=1)
    if with_auto_mask:
        ident = (tgt_img - ref_img).abs().mean(dim=1)
        soft |= (d.abs().clamp(0, 1).mean(dim=1) - ident).abs() < eps_val + slope_img.mean(dim=1)
    if with_ssim:
        # (1 - SSIM)/2 before its clamp
        xp, yp = _reflect_pad1(tgt_img), _reflect_pad1(warped)
        mu_x, mu_y = _box3(xp), _box3(yp)
        sig_x, sig_y = _box3(xp ** 2) - mu_x ** 2, _box3(yp ** 2) - mu_y ** 2
        sig_xy = _box3(xp * yp) - mu_x * mu_y
        raw = (1 - (2 * mu_x * mu_y + SSIM_C1) * (2 * sig_xy + SSIM_C2) / ((mu_x ** 2 + mu_y ** 2 + SSIM_C1) * (sig_x + sig_y + SSIM_C2))) / 2
        soft |= ((raw.abs() < eps_val) | ((raw - 1).abs() < eps_val)).any(dim=1)
    dd = (comp_depth - proj_depth).abs() / (comp_depth + proj_depth)
    # (a pixel that samples nothing has projected depth 0 and diff_depth == 1 exactly, in any precision: no hazard)
    # d dd / d D_p is at most 2 / (Z + D_p) in magnitude
    edd = (eps_val + 2 * slope_dep / (comp_depth + proj_depth).squeeze(1)).unsqueeze(1)
    soft |= ((dd < edd) | (((dd - 1).abs() < edd) & (proj_depth != 0))).squeeze(1)
    xa = ix.floor().clamp(0, W - 2).long()
    ya = iy.floor().clamp(0, H - 2).long()
    return {"hard": hard, "soft": soft, "own": own, "xa": xa, "ya": ya}


def _dilate(mask, r):
    return F.max_pool2d(mask.to(torch.float32).unsqueeze(1), 2 * r + 1, 1, r).squeeze(1) > 0


def unsafe_gradient_entries(m):
    """From pairwise_gate_margins: (dense, scatter) boolean maps [B,H,W] of the entries of dL/d tgt_depth and of
    dL/d ref_depth of that pair-direction whose value may depend on which side an fp32 gate fell."""
    dense = _dilate(m["hard"], 2) | _dilate(m["soft"], 1) | m["own"]
    src = _dilate(m["hard"], 1) | m["soft"]          # target pixels whose scattered VALUE may have jumped
    B, H, W = src.shape
    scatter = torch.zeros(B, H * W, dtype=torch.bool)
    b, y, x = src.nonzero(as_tuple=True)
    for dy in (0, 1):
        for dx in (0, 1):
            lin = (m["ya"][b, y, x] + dy) * W + (m["xa"][b, y, x] + dx)
            scatter[b, lin] = True
    return dense, scatter.reshape(B, H, W)
