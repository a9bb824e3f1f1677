"""Tensor-level wrappers of the C ABI: allocate outputs, pass raw pointers + the current stream.

Device-agnostic on purpose: the product (``ops.py``) calls these with the HIP library and CUDA
tensors; the CPU-only CI calls them with tests/hostsim's build of the same sources and host
tensors.  Nothing here computes anything.
"""
from __future__ import annotations

import torch

from . import _lib

WITH_SSIM, WITH_MASK, WITH_AUTO_MASK, PAD_BORDER = 1, 2, 4, 8
ROT = {"euler": 0, "quat": 1}


def make_flags(with_ssim=False, with_mask=False, with_auto_mask=False, padding_mode="zeros"):
    if padding_mode not in ("zeros", "border"):
        raise ValueError(f"padding_mode must be 'zeros' or 'border', got {padding_mode!r}")
    # the reference compares the int flags with `== True` (loss_functions.py:103,107,111)
    return ((WITH_SSIM if with_ssim == True else 0) | (WITH_MASK if with_mask == True else 0) |  # noqa: E712
            (WITH_AUTO_MASK if with_auto_mask == True else 0) | (PAD_BORDER if padding_mode == "border" else 0))  # noqa: E712


def _suffix(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError(f"unsupported dtype {t.dtype}")


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk(*ts):
    dev, dt = ts[0].device, ts[0].dtype
    for t in ts:
        if t is None:
            continue
        if t.device != dev or t.dtype != dt:
            raise TypeError("all tensors of one call must share device and dtype")
        if not t.is_contiguous():
            raise ValueError("tensors must be contiguous")


def _ws(lib, fn, like, *dims):
    n = lib.size(fn, *dims)
    return torch.empty(n, dtype=torch.uint8, device=like.device)


# -- inverse_warp2 -----------------------------------------------------------------------------
def warp_fwd(lib, img, depth, ref_depth, pose, K, flags):
    _chk(img, depth, ref_depth, pose, K)
    B, _, H, W = img.shape
    ws = _ws(lib, "scsfm_warp_ws_bytes", img, B)
    o_img = torch.empty_like(img)
    o_valid, o_pd, o_cd = (torch.empty_like(depth) for _ in range(3))
    lib.call(f"scsfm_warp_fwd_{_suffix(img)}", B, H, W, _p(img), _p(depth), _p(ref_depth), _p(pose), _p(K), flags,
             _p(ws), _p(o_img), _p(o_valid), _p(o_pd), _p(o_cd), _stream(img))
    return o_img, o_valid, o_pd, o_cd


def warp_bwd(lib, img, depth, ref_depth, pose, K, flags, g_img, g_pd, g_cd):
    _chk(img, depth, ref_depth, pose, K, g_img, g_pd, g_cd)
    B, _, H, W = img.shape
    ws = _ws(lib, "scsfm_warp_ws_bytes", img, B)
    g_depth = torch.zeros_like(depth)
    g_ref = torch.zeros_like(ref_depth)
    g_pose = torch.empty_like(pose)
    lib.call(f"scsfm_warp_bwd_{_suffix(img)}", B, H, W, _p(img), _p(depth), _p(ref_depth), _p(pose), _p(K), flags,
             _p(ws), _p(g_img), _p(g_pd), _p(g_cd), _p(g_depth), _p(g_ref), _p(g_pose), _stream(img))
    return g_depth, g_ref, g_pose


# -- pose_vec2mat ------------------------------------------------------------------------------
def pose_fwd(lib, vec, mode):
    _chk(vec)
    mat = torch.empty(vec.shape[0], 3, 4, dtype=vec.dtype, device=vec.device)
    lib.call(f"scsfm_pose_vec2mat_fwd_{_suffix(vec)}", vec.shape[0], _p(vec), ROT[mode], _p(mat), _stream(vec))
    return mat


def pose_bwd(lib, vec, mode, g_mat):
    _chk(vec, g_mat)
    g_vec = torch.empty_like(vec)
    lib.call(f"scsfm_pose_vec2mat_bwd_{_suffix(vec)}", vec.shape[0], _p(vec), ROT[mode], _p(g_mat), _p(g_vec),
             _stream(vec))
    return g_vec


# -- compute_pairwise_loss ---------------------------------------------------------------------
def pair_fwd(lib, tgt_img, ref_img, tgt_depth, ref_depth, pose, K, flags):
    """-> (out[4] = {photo, geom, sum_mask, 0}, ws).  ``ws`` must be handed to pair_bwd."""
    _chk(tgt_img, ref_img, tgt_depth, ref_depth, pose, K)
    B, _, H, W = tgt_img.shape
    ws = _ws(lib, "scsfm_pair_ws_bytes", tgt_img, B, H, W)
    out = torch.empty(4, dtype=tgt_img.dtype, device=tgt_img.device)
    lib.call(f"scsfm_pair_fwd_{_suffix(tgt_img)}", B, H, W, _p(tgt_img), _p(ref_img), _p(tgt_depth), _p(ref_depth),
             _p(pose), _p(K), flags, _p(ws), _p(out), _stream(tgt_img))
    return out, ws


def pair_bwd(lib, tgt_img, ref_img, tgt_depth, ref_depth, pose, K, flags, ws, g_photo, g_geom,
             g_tgt_depth=None, g_ref_depth=None):
    """Accumulates into g_tgt_depth / g_ref_depth (allocated zeroed when None); returns them and
    g_pose [B,6]."""
    _chk(tgt_img, ref_img, tgt_depth, ref_depth, pose, K, g_photo, g_geom, g_tgt_depth, g_ref_depth)
    B, _, H, W = tgt_img.shape
    if g_tgt_depth is None:
        g_tgt_depth = torch.zeros_like(tgt_depth)
    if g_ref_depth is None:
        g_ref_depth = torch.zeros_like(ref_depth)
    g_pose = torch.empty_like(pose)
    lib.call(f"scsfm_pair_bwd_{_suffix(tgt_img)}", B, H, W, _p(tgt_img), _p(ref_img), _p(tgt_depth), _p(ref_depth),
             _p(pose), _p(K), flags, _p(ws), _p(g_photo), _p(g_geom), _p(g_tgt_depth), _p(g_ref_depth), _p(g_pose),
             _stream(tgt_img))
    return g_tgt_depth, g_ref_depth, g_pose


# -- get_smooth_loss ---------------------------------------------------------------------------
def smooth_fwd(lib, depth, img, out=None):
    _chk(depth, img)
    B, _, H, W = img.shape
    ws = _ws(lib, "scsfm_smooth_ws_bytes", img, B, H, W)
    if out is None:
        out = torch.empty(1, dtype=img.dtype, device=img.device)
    lib.call(f"scsfm_smooth_fwd_{_suffix(img)}", B, H, W, _p(depth), _p(img), _p(ws), _p(out), _stream(img))
    return out, ws


def smooth_bwd(lib, depth, img, ws, g_loss, g_depth=None):
    _chk(depth, img, g_loss, g_depth)
    B, _, H, W = img.shape
    if g_depth is None:
        g_depth = torch.zeros_like(depth)
    lib.call(f"scsfm_smooth_bwd_{_suffix(img)}", B, H, W, _p(depth), _p(img), _p(ws), _p(g_loss), _p(g_depth),
             _stream(img))
    return g_depth
