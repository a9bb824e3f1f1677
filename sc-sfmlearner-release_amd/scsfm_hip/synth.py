"""Seeded synthetic batches with the shapes and value ranges of the reference's training step
(SURVEY.md §8d).  No dataset is reachable, so tests, ``smoke()`` and ``bench.py`` all draw their
inputs here; everything is generated with numpy's PCG64 on the host and returned as CPU fp32
tensors so that the HIP path and the CPU oracle see identical bits.

Value ranges follow the reference: images normalised with mean .45 / std .225 (train.py:92-93),
depth = 1/(10*sigmoid+0.01) (models/DispResNet.py:98, train.py:427), pose = 0.01*(...)
(models/PoseResNet.py:49), intrinsics = KITTI P_rect_02 rescaled to HxW with a +-15 % zoom/shift
per sample as custom_transforms.RandomScaleCrop (custom_transforms.py:62-84) would produce.
"""
from __future__ import annotations

import numpy as np
import torch

# KITTI raw P_rect_02 for 1242x375, and an NYUv2-like pinhole for 640x480.
_K_BASE = {
    "kitti": (721.5377, 721.5377, 609.5593, 172.854, 1242.0, 375.0),
    "nyu": (518.8579, 519.4696, 325.5824, 253.7362, 640.0, 480.0),
}


def _lowpass(rng, shape, coarse):
    """Random field on a coarse grid, bilinearly upsampled to ``shape[-2:]`` (values in [0,1))."""
    B, C, H, W = shape
    ch, cw = coarse
    g = torch.from_numpy(rng.random((B, C, ch, cw), dtype=np.float32))
    return torch.nn.functional.interpolate(g, size=(H, W), mode="bilinear", align_corners=True)


def intrinsics(rng, B, H, W, dataset="kitti", jitter=0.15):
    fx, fy, cx, cy, w0, h0 = _K_BASE[dataset]
    K = np.zeros((B, 3, 3), dtype=np.float32)
    for b in range(B):
        z = 1.0 + jitter * rng.random()
        sx, sy = W / w0 * z, H / h0 * z
        ox = rng.random() * (z - 1.0) * W
        oy = rng.random() * (z - 1.0) * H
        K[b] = [[fx * sx, 0, cx * sx - ox], [0, fy * sy, cy * sy - oy], [0, 0, 1]]
    return torch.from_numpy(K)


def _scene_objects(rng, B, h, w):
    """Fronto-parallel objects of a synthetic scene: per image a list of (mask [h, w] bool, extra inverse depth, colour
    shift [3]) in paint order (later objects occlude earlier ones).  About 44 objects per 256 x 832 image (scaled with
    the area, at least 3): rectangles and ellipses of 4 .. 40 % of the image height, anywhere in the frame."""
    n_obj = max(3, int(round(44.0 * h * w / (256.0 * 832.0))))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    scenes = []
    for _ in range(B):
        objs = []
        for _ in range(n_obj):
            cy, cx = rng.random() * h, rng.random() * w
            ry = (0.04 + 0.36 * rng.random()) * h * 0.5
            rx = ry * (0.5 + 1.5 * rng.random())
            ellipse = rng.random() < 0.5
            if ellipse:
                m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
            else:
                m = (np.abs(yy - cy) <= ry) & (np.abs(xx - cx) <= rx)
            # the object stands in front of the background: its inverse depth exceeds the background's (median under it)
            # by 0.5 .. 4.5, i.e. a disparity jump of ~5 .. 50 px at its outline for the bench's motion (f |t| ~ 8 .. 12
            # px per unit of 1/Z)
            extra = 0.5 + 4.0 * rng.random()
            colour = (rng.random(3).astype(np.float32) - 0.5) * 0.6
            objs.append((m, np.float32(extra), colour))
        scenes.append(objs)
    return scenes


def scene_edge_fraction(depth_map, jump=0.3):
    """Share of pixels on an occlusion edge of a [B,1,H,W] depth map: inverse depth differs from a 4-neighbour's by more
    than `jump` (the law's smooth background varies by < 0.1 per pixel)."""
    inv = 1.0 / depth_map
    dx = (inv[..., :, 1:] - inv[..., :, :-1]).abs() > jump
    dy = (inv[..., 1:, :] - inv[..., :-1, :]).abs() > jump
    e = torch.zeros_like(inv, dtype=torch.bool)
    e[..., :, 1:] |= dx; e[..., :, :-1] |= dx
    e[..., 1:, :] |= dy; e[..., :-1, :] |= dy
    return float(e.double().mean())


def image_law(depth):
    """The image law that goes with a depth law in the tests and in bench.py."""
    return {"smooth": "smooth", "scene": "scene"}.get(depth, "iid")


def make_batch(B, H, W, n_ref=2, seed=0, depth="smooth", image="smooth", dataset="kitti",
               pose_scale=0.01, num_scales=1):
    """Returns a dict with the argument structure of the reference's loss calls
    (train.py:259-266): ``tgt_img`` [B,3,H,W], ``ref_imgs`` n_ref x [B,3,H,W], ``intrinsics``
    [B,3,3], ``tgt_depth`` list over scales of [B,1,H/2^s,W/2^s], ``ref_depths`` list of such
    lists, ``poses`` / ``poses_inv`` lists of [B,6].

    depth: 'iid' = 1/(10*U+0.01) per pixel (stress: incoherent gathers);
           'smooth' = the same law applied to a coarse random field (realistic locality);
           'scene' = what a trained DispResNet emits on a real sequence (datasets/sequence_folders.py:57-58): the smooth
           law as background with a few dozen fronto-parallel objects per image in front of it -- piecewise-smooth
           depth, occlusion edges on ~5 % of the pixels, disparity jumps of ~5 .. 50 px across them.
    image: 'iid' = (U-0.45)/0.225 per pixel; 'smooth' = low-passed field + 10 % iid texture; 'scene' = the smooth image
           with every object of the frame's depth map painted in its own colour shift (edges coincide with the depth's;
           only with depth='scene').
    A frame's image and its depth pyramid share one object layout (scale s is the layout sampled at H >> s, W >> s).
    """
    rng = np.random.default_rng(seed)
    if image == "scene" and depth != "scene":
        raise ValueError("image='scene' needs depth='scene' (the objects are the depth map's)")

    def img(objs=None):
        if image == "iid":
            x = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32))
        else:
            x = 0.9 * _lowpass(rng, (B, 3, H, W), (max(H // 8, 2), max(W // 8, 2)))
            if image == "scene":
                xn = x.numpy()
                for b in range(B):
                    for m, _, colour in objs[b]:
                        base = xn[b][:, m].mean(axis=1, keepdims=True)  # flat object colour + a rest of the background's texture
                        xn[b][:, m] = np.clip(0.3 * xn[b][:, m] + 0.7 * base + colour[:, None] * 0.9, 0.0, 0.9)
            x = x + 0.1 * torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32))
        return ((x - 0.45) / 0.225).contiguous()

    def dep(h, w, objs=None):
        if depth == "iid":
            s = torch.from_numpy(rng.random((B, 1, h, w), dtype=np.float32))
        else:
            s = _lowpass(rng, (B, 1, h, w), (max(h // 32, 2), max(w // 32, 2)))
        inv = 10.0 * s + 0.01
        if depth == "scene":
            iv = inv.numpy()
            sy, sx = H // h, W // w
            for b in range(B):
                bg = iv[b, 0].copy()
                placed = []
                for m, extra, _ in objs[b]:
                    ms = m[::sy, ::sx][:h, :w]
                    if ms.any():  # constant depth over the object, in front of the background it covers
                        placed.append((min(float(np.median(bg[ms])) + float(extra), 10.01), ms))
                for v, ms in sorted(placed, key=lambda t: t[0]):  # far to near: nearer objects occlude
                    iv[b, 0][ms] = v
        return (1.0 / inv).contiguous()

    def dep_pyr(objs=None):
        return [dep(H >> s, W >> s, objs) for s in range(num_scales)]

    if depth == "scene":
        # one object layout per frame, drawn first so that images and depth maps below can share it
        layouts = [_scene_objects(rng, B, H, W) for _ in range(1 + n_ref)]
        use_img = image == "scene"
        out = {
            "tgt_img": img(layouts[0] if use_img else None),
            "ref_imgs": [img(layouts[1 + i] if use_img else None) for i in range(n_ref)],
            "intrinsics": intrinsics(rng, B, H, W, dataset),
            "tgt_depth": dep_pyr(layouts[0]),
            "ref_depths": [dep_pyr(layouts[1 + i]) for i in range(n_ref)],
        }
        out["poses"] = [torch.from_numpy((pose_scale * rng.standard_normal((B, 6))).astype(np.float32)) for _ in range(n_ref)]
        out["poses_inv"] = [torch.from_numpy((pose_scale * rng.standard_normal((B, 6))).astype(np.float32)) for _ in range(n_ref)]
        return out

    def pose():
        return torch.from_numpy((pose_scale * rng.standard_normal((B, 6))).astype(np.float32))

    out = {
        "tgt_img": img(),
        "ref_imgs": [img() for _ in range(n_ref)],
        "intrinsics": intrinsics(rng, B, H, W, dataset),
        "tgt_depth": dep_pyr(),
        "ref_depths": [dep_pyr() for _ in range(n_ref)],
        "poses": [pose() for _ in range(n_ref)],
        "poses_inv": [pose() for _ in range(n_ref)],
    }
    return out
