"""Host side of the device input transform (csrc/scsfm_augment.hip): the random draws of the
reference's RandomHorizontalFlip / RandomScaleCrop in the reference's order
(custom_transforms.py:46-84), the intrinsics update, and Pillow's bicubic coefficient tables
(ImagingResample: precompute_coeffs + normalize_coeffs_8bpc) for the cropped window.  The byte-level
result equals PIL's ``Image.resize`` (default filter) followed by the crop, the flip before it."""
from __future__ import annotations

import random

import numpy as np
import torch

from . import _lib, capi

PRECISION_BITS = 22  # Pillow: 32 - 8 - 2


def draw_params(n_samples, H, W):
    """One record per sample, drawn with the RNG calls of the reference transforms, in their order:
    ``random.random()`` (flip), ``np.random.uniform(1, 1.15, 2)`` (x, y scaling), then
    ``np.random.randint`` for the vertical and the horizontal crop offset."""
    recs = []
    for _ in range(n_samples):
        flip = random.random() < 0.5
        xs, ys = np.random.uniform(1, 1.15, 2)
        sh, sw = int(H * ys), int(W * xs)
        oy = np.random.randint(sh - H + 1)
        ox = np.random.randint(sw - W + 1)
        # x/y scaling stay numpy float64 scalars: the reference multiplies the float32 intrinsics by them in
        # place, which numpy evaluates in float64 before rounding back (custom_transforms.py:72-73)
        recs.append(dict(flip=flip, x_scaling=np.float64(xs), y_scaling=np.float64(ys), sw=sw, sh=sh, ox=int(ox), oy=int(oy)))
    return recs


def update_intrinsics(K, recs, W):
    """K [S,3,3] (numpy or tensor, float32) -> transformed copy (custom_transforms.py:54-56,70-83)."""
    K = np.array(K, dtype=np.float32, copy=True)
    for s, r in enumerate(recs):
        if r["flip"]:
            K[s, 0, 2] = W - K[s, 0, 2]
        K[s, 0] *= np.float64(r["x_scaling"])
        K[s, 1] *= np.float64(r["y_scaling"])
        K[s, 0, 2] -= r["ox"]
        K[s, 1, 2] -= r["oy"]
    return K


def _bicubic(x):
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def axis_table(in_size, out_size, start, length):
    """Rows {first source index, tap count, 5 integer taps, 0} for output positions start .. start+length-1
    of an axis resized in_size -> out_size (up-scaling: support 2, at most 5 taps); the identity entry when
    the size does not change (Pillow skips that pass)."""
    tab = np.zeros((length, 8), dtype=np.int32)
    xx = np.arange(start, start + length)
    if out_size == in_size:
        tab[:, 0], tab[:, 1], tab[:, 2] = xx, 1, 1 << PRECISION_BITS
        return tab
    assert out_size > in_size, "RandomScaleCrop only zooms in"
    scale = in_size / out_size
    support = 2.0  # bicubic support x max(scale, 1)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    cnt = xmax - xmin
    assert cnt.max() <= 5
    k = np.zeros((length, 5))
    ww = np.zeros(length)
    for i in range(5):
        w = np.where(i < cnt, _bicubic(i + xmin - center + 0.5), 0.0)
        k[:, i] = w
        ww = ww + w  # sequential double sum, as the C loop
    k = np.where(ww[:, None] != 0.0, k / ww[:, None], k)
    ki = np.where(k < 0, (-0.5 + k * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + k * (1 << PRECISION_BITS)).astype(np.int64))
    tab[:, 0], tab[:, 1], tab[:, 2:7] = xmin, cnt, ki
    return tab


def byte_lut(mean=0.45, std=0.225):
    """fl32(fl32(fl32(v / 255) - mean) / std): ArrayToTensor + Normalize on one byte."""
    v = np.arange(256, dtype=np.float32) / np.float32(255)
    return ((v - np.float32(mean)) / np.float32(std)).astype(np.float32)


def tables(recs, H, W):
    params = np.zeros((len(recs), 8), dtype=np.int32)
    htab = np.zeros((len(recs), W, 8), dtype=np.int32)
    vtab = np.zeros((len(recs), H, 8), dtype=np.int32)
    for s, r in enumerate(recs):
        params[s, :5] = [int(r["flip"]), r["sw"], r["sh"], r["ox"], r["oy"]]
        htab[s] = axis_table(W, r["sw"], r["ox"], W)
        vtab[s] = axis_table(H, r["sh"], r["oy"], H)
    return params, htab, vtab


def augment(frames_u8, recs, lib=None, mean=0.45, std=0.225):
    """frames_u8: uint8 [S, T, H, W, 3] on the device -> fp32 [T, S, 3, H, W] (frame-major: out[t] is the
    contiguous batch of frame t), flipped / zoomed / cropped per sample record and normalised.
    ``lib`` defaults to the HIP library."""
    lib = lib or _lib.get()
    S, T, H, W, C = frames_u8.shape
    assert C == 3 and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and len(recs) == S
    dev = frames_u8.device
    params, htab, vtab = tables(recs, H, W)
    to = lambda a: torch.from_numpy(a).to(dev)
    p, h, v, lut = to(params), to(htab), to(vtab), to(byte_lut(mean, std))
    out = torch.empty(T, S, 3, H, W, dtype=torch.float32, device=dev)
    lib.call("scsfm_augment_u8_f32", S * T, T, H, W, frames_u8.data_ptr(), p.data_ptr(), h.data_ptr(), v.data_ptr(),
             lut.data_ptr(), out.data_ptr(), capi._stream(frames_u8))
    return out
