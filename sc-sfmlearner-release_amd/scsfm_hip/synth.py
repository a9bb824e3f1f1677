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


def make_batch(B, H, W, n_ref=2, seed=0, depth="smooth", image="smooth", dataset="kitti",
               pose_scale=0.01, num_scales=1):
    """Returns a dict with the argument structure of the reference's loss calls
    (train.py:259-266): ``tgt_img`` [B,3,H,W], ``ref_imgs`` n_ref x [B,3,H,W], ``intrinsics``
    [B,3,3], ``tgt_depth`` list over scales of [B,1,H/2^s,W/2^s], ``ref_depths`` list of such
    lists, ``poses`` / ``poses_inv`` lists of [B,6].

    depth: 'iid' = 1/(10*U+0.01) per pixel (stress: incoherent gathers);
           'smooth' = the same law applied to a coarse random field (realistic locality).
    image: 'iid' = (U-0.45)/0.225 per pixel; 'smooth' = low-passed field + 10 % iid texture.
    """
    rng = np.random.default_rng(seed)

    def img():
        if image == "iid":
            x = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32))
        else:
            x = 0.9 * _lowpass(rng, (B, 3, H, W), (max(H // 8, 2), max(W // 8, 2)))
            x = x + 0.1 * torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32))
        return ((x - 0.45) / 0.225).contiguous()

    def dep(h, w):
        if depth == "iid":
            s = torch.from_numpy(rng.random((B, 1, h, w), dtype=np.float32))
        else:
            s = _lowpass(rng, (B, 1, h, w), (max(h // 32, 2), max(w // 32, 2)))
        return (1.0 / (10.0 * s + 0.01)).contiguous()

    def dep_pyr():
        return [dep(H >> s, W >> s) for s in range(num_scales)]

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
