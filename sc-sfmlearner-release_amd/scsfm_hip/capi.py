"""Tensor-level wrappers of the C ABI: allocate outputs, pass raw pointers + the current stream.

Device-agnostic on purpose: the product (``ops.py``) calls these with the HIP library and CUDA
tensors; the CPU-only CI calls them with tests/hostsim's build of the same sources and host
tensors.  Nothing here computes anything.
"""
from __future__ import annotations

import torch

from . import _lib

WITH_SSIM, WITH_MASK, WITH_AUTO_MASK, PAD_BORDER, LEGACY_GRID = 1, 2, 4, 8, 16
ROT_QUAT_FLAG, C2P_OVERWRITE = 32, 64
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


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    """Raw handle of torch's current stream on t's device (0 = the host build of the test suite)."""
    if not t.is_cuda:
        return 0
    if _raw_stream is not None:  # skips building a torch.cuda.Stream object (~10 us per call)
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


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


def check_sizes(t, name, expected):
    """Shape guard with the reference's message (inverse_warp.py:20-26); raises AssertionError."""
    if tuple(t.shape) == expected:
        return
    ok = t.dim() == len(expected) and all(t.size(i) == e for i, e in enumerate(expected) if e is not None)
    assert ok, "wrong size for {}, expected {}, got  {}".format(
        name, "x".join("?" if e is None else str(e) for e in expected), list(t.size()))


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


def warp_bwd(lib, img, depth, ref_depth, pose, K, flags, g_img, g_pd, g_cd, need_img=False, need_K=False):
    _chk(img, depth, ref_depth, pose, K, g_img, g_pd, g_cd)
    B, _, H, W = img.shape
    ws = _ws(lib, "scsfm_warp_ws_bytes", img, B)
    g_depth = torch.zeros_like(depth)
    g_ref = torch.zeros_like(ref_depth)
    g_pose = torch.empty_like(pose)
    lib.call(f"scsfm_warp_bwd_{_suffix(img)}", B, H, W, _p(img), _p(depth), _p(ref_depth), _p(pose), _p(K), flags,
             _p(ws), _p(g_img), _p(g_pd), _p(g_cd), _p(g_depth), _p(g_ref), _p(g_pose), _stream(img))
    if not (need_img or need_K):
        return g_depth, g_ref, g_pose
    # the gradients of the data inputs (the reference's autograd reaches them): the bilinear splat of g_img into the
    # sampled image, dL/d intrinsics from the sums the backward left in ws
    g_src = torch.zeros_like(img) if (need_img and g_img is not None) else None
    g_K = torch.empty_like(K) if need_K else None
    lib.call(f"scsfm_warp_bwd_inputs_{_suffix(img)}", B, H, W, _p(depth), _p(pose), _p(K), flags, _p(ws), _p(g_img),
             _p(g_src), _p(g_K), _stream(img))
    return g_depth, g_ref, g_pose, g_src, g_K


# -- pixel2cam / cam2pixel / cam2pixel2 ----------------------------------------------------------
def pixel2cam_fwd(lib, depth, Kinv):
    """depth [B,H,W], intrinsics_inv [B,3,3] -> cam [B,3,H,W] (inverse_warp.py:29-44)."""
    _chk(depth, Kinv)
    B, H, W = depth.shape
    cam = torch.empty((B, 3, H, W), dtype=depth.dtype, device=depth.device)
    lib.call(f"scsfm_pixel2cam_fwd_{_suffix(depth)}", B, H, W, _p(depth), _p(Kinv), _p(cam), _stream(depth))
    return cam


def pixel2cam_bwd(lib, Kinv, g_cam):
    _chk(Kinv, g_cam)
    B, _, H, W = g_cam.shape
    g_depth = torch.empty((B, H, W), dtype=g_cam.dtype, device=g_cam.device)
    lib.call(f"scsfm_pixel2cam_bwd_{_suffix(g_cam)}", B, H, W, _p(Kinv), _p(g_cam), _p(g_depth), _stream(g_cam))
    return g_depth


def cam2pixel_fwd(lib, cam, rot, tr, flags, want_z):
    """cam [B,3,H,W], rot [B,3,3] | None, tr [B,3,1] | None -> grid [B,H,W,2] (+ z [B,1,H,W])
    (inverse_warp.py:47-74, 194-227)."""
    _chk(cam, rot, tr)
    B, _, H, W = cam.shape
    grid = torch.empty((B, H, W, 2), dtype=cam.dtype, device=cam.device)
    z = torch.empty((B, 1, H, W), dtype=cam.dtype, device=cam.device) if want_z else None
    lib.call(f"scsfm_cam2pixel_fwd_{_suffix(cam)}", B, H, W, _p(cam), _p(rot), _p(tr), flags, _p(grid), _p(z), _stream(cam))
    return grid, z


def cam2pixel_bwd(lib, cam, rot, tr, flags, g_grid, g_z):
    """-> g_cam [B,3,H,W], g_rot [B,3,3], g_tr [B,3,1] (the latter two reduced in fp64 on the device)."""
    _chk(cam, rot, tr, g_grid, g_z)
    B, _, H, W = cam.shape
    g_cam = torch.empty_like(cam)
    acc = torch.empty((B, 12), dtype=torch.float64, device=cam.device)
    lib.call(f"scsfm_cam2pixel_bwd_{_suffix(cam)}", B, H, W, _p(cam), _p(rot), _p(tr), flags, _p(g_grid), _p(g_z), _p(g_cam),
             _p(acc), _stream(cam))
    return g_cam, acc[:, :9].reshape(B, 3, 3).to(cam.dtype), acc[:, 9:].reshape(B, 3, 1).to(cam.dtype)


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
def pair_fwd_into(lib, tgt_img, ref_img, tgt_depth, ref_depth, pose, K, flags, out):
    """Writes {photo, geom, S_photo, S_geom, S_mask, 0, 0, 0} into ``out`` (8 contiguous elements); returns the
    workspace that must be handed to pair_bwd."""
    _chk(tgt_img, ref_img, tgt_depth, ref_depth, pose, K, out)
    B, _, H, W = tgt_img.shape
    check_sizes(ref_img, "ref_img", (B, 3, H, W))
    check_sizes(tgt_depth, "tgt_depth", (B, 1, H, W))
    check_sizes(ref_depth, "ref_depth", (B, 1, H, W))
    check_sizes(pose, "pose", (B, 6))
    check_sizes(K, "intrinsics", (B, 3, 3))
    ws = _ws(lib, "scsfm_pair_ws_bytes", tgt_img, B, H, W)
    lib.call(f"scsfm_pair_fwd_{_suffix(tgt_img)}", B, H, W, _p(tgt_img), _p(ref_img), _p(tgt_depth), _p(ref_depth),
             _p(pose), _p(K), flags, _p(ws), _p(out), _stream(tgt_img))
    return ws


def pair_fwd(lib, tgt_img, ref_img, tgt_depth, ref_depth, pose, K, flags):
    """-> (out[8] = {photo, geom, S_photo, S_geom, S_mask, 0, 0, 0}, ws)."""
    out = torch.empty(8, dtype=tgt_img.dtype, device=tgt_img.device)
    ws = pair_fwd_into(lib, tgt_img, ref_img, tgt_depth, ref_depth, pose, K, flags, out)
    return out, ws


def pair_refinalize(lib, shape, ws, out):
    """Exact data-parallel mode: out[2..4] hold globally reduced sums; rewrites out[0..1] and the
    backward coefficients inside ``ws``."""
    B, H, W = shape
    lib.call(f"scsfm_pair_refinalize_{_suffix(out)}", B, H, W, _p(ws), _p(out), _stream(out))


def pixel2cam_bwd_intrinsics(lib, depth, g_cam):
    """dL/d intrinsics_inv [B,3,3] of pixel2cam."""
    _chk(depth, g_cam)
    B, _, H, W = g_cam.shape
    g_kinv = torch.empty((B, 3, 3), dtype=g_cam.dtype, device=g_cam.device)
    lib.call(f"scsfm_pixel2cam_bwd_intrinsics_{_suffix(g_cam)}", B, H, W, _p(depth), _p(g_cam), _p(g_kinv), _stream(g_cam))
    return g_kinv


def pair_bwd_scratch(lib, like, B, H, W):
    """Device scratch of one pair backward (dL/d warped colours + dL/d diff_depth between its two
    kernels); contents are irrelevant between calls, so consecutive calls on a stream may share it."""
    return _ws(lib, "scsfm_pair_bwd_scratch_bytes", like, B, H, W)


def pair_bwd(lib, tgt_img, ref_img, tgt_depth, ref_depth, pose, K, flags, ws, g_photo, g_geom,
             g_tgt_depth=None, g_ref_depth=None, scratch=None, need_tgt_img=False, need_ref_img=False, need_K=False):
    """Accumulates into g_tgt_depth / g_ref_depth (allocated zeroed when None); returns them and
    g_pose [B,6]."""
    _chk(tgt_img, ref_img, tgt_depth, ref_depth, pose, K, g_photo, g_geom, g_tgt_depth, g_ref_depth)
    B, _, H, W = tgt_img.shape
    if g_tgt_depth is None:
        g_tgt_depth = torch.zeros_like(tgt_depth)
    if g_ref_depth is None:
        g_ref_depth = torch.zeros_like(ref_depth)
    g_pose = torch.empty_like(pose)
    if scratch is None:
        scratch = pair_bwd_scratch(lib, tgt_img, B, H, W)
    lib.call(f"scsfm_pair_bwd_{_suffix(tgt_img)}", B, H, W, _p(tgt_img), _p(ref_img), _p(tgt_depth), _p(ref_depth),
             _p(pose), _p(K), flags, _p(ws), _p(scratch), _p(g_photo), _p(g_geom), _p(g_tgt_depth), _p(g_ref_depth),
             _p(g_pose), _stream(tgt_img))
    if not (need_tgt_img or need_ref_img or need_K):
        return g_tgt_depth, g_ref_depth, g_pose
    d = (PairDesc * 1)()
    d[0].tgt_img, d[0].ref_img, d[0].tgt_depth, d[0].ref_depth, d[0].pose = tgt_img.data_ptr(), ref_img.data_ptr(), \
        tgt_depth.data_ptr(), ref_depth.data_ptr(), pose.data_ptr()
    d[0].ws = ws.data_ptr()
    g_ti = torch.zeros_like(tgt_img) if need_tgt_img else None
    g_ri = torch.zeros_like(ref_img) if need_ref_img else None
    g_K = torch.empty_like(K) if need_K else None
    d[0].g_tgt_img, d[0].g_ref_img = _p(g_ti) or None, _p(g_ri) or None
    lib.call(f"scsfm_pairs_bwd_inputs_{_suffix(tgt_img)}", 1, _ct.addressof(d), B, H, W, _p(K), flags, _p(g_photo),
             _p(g_geom), _p(g_K), _stream(tgt_img))
    return g_tgt_depth, g_ref_depth, g_pose, g_ti, g_ri, g_K


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


# -- SSIM layer / mean_on_mask (stand-alone public helpers) --------------------------------------
def ssim_fwd(lib, x, y):
    _chk(x, y)
    assert x.shape == y.shape and x.dim() == 4, "SSIM expects two [B,C,H,W] tensors of equal shape"
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    lib.call(f"scsfm_ssim_fwd_{_suffix(x)}", B * C, H, W, _p(x), _p(y), _p(out), _stream(x))
    return out


def ssim_bwd(lib, x, y, g_out, need_x=True, need_y=True):
    _chk(x, y, g_out)
    B, C, H, W = x.shape
    g_x = torch.empty_like(x) if need_x else None
    g_y = torch.empty_like(y) if need_y else None
    lib.call(f"scsfm_ssim_bwd_{_suffix(x)}", B * C, H, W, _p(x), _p(y), _p(g_out), _p(g_x), _p(g_y), _stream(x))
    return g_x, g_y


def masked_mean_fwd(lib, diff, mask):
    _chk(diff, mask)
    B, C = diff.shape[0], diff.shape[1]
    HW = diff[0, 0].numel()
    Cm = mask.shape[1]
    assert mask.shape[0] == B and Cm in (1, C) and mask[0, 0].numel() == HW, "mask must broadcast over channels only"
    ws = torch.empty(lib.size("scsfm_masked_mean_ws_bytes"), dtype=torch.uint8, device=diff.device)
    out = torch.empty(1, dtype=diff.dtype, device=diff.device)
    lib.call(f"scsfm_masked_mean_fwd_{_suffix(diff)}", B, C, Cm, HW, _p(diff), _p(mask), _p(ws), _p(out), _stream(diff))
    return out, ws


def masked_mean_bwd(lib, diff_shape, mask, ws, g):
    _chk(mask, g)
    B, C = diff_shape[0], diff_shape[1]
    HW = 1
    for d in diff_shape[2:]:
        HW *= d
    g_diff = torch.empty(diff_shape, dtype=mask.dtype, device=mask.device)
    lib.call(f"scsfm_masked_mean_bwd_{_suffix(mask)}", B, C, mask.shape[1], HW, _p(mask), _p(ws), _p(g), _p(g_diff),
             _stream(mask))
    return g_diff


def masked_mean_bwd_mask(lib, diff, mask_shape, ws, g):
    """dL/d mask (a floating-point mask is differentiable in the reference)."""
    _chk(diff, g)
    B, C = diff.shape[0], diff.shape[1]
    HW = diff[0, 0].numel()
    g_mask = torch.empty(mask_shape, dtype=diff.dtype, device=diff.device)
    lib.call(f"scsfm_masked_mean_bwd_mask_{_suffix(diff)}", B, C, mask_shape[1], HW, _p(diff), _p(ws), _p(g), _p(g_mask),
             _stream(diff))
    return g_mask


# -- compute_photo_and_geometry_loss: refs x scales x both directions ----------------------------
import ctypes as _ct
import functools as _ft


class PairDesc(_ct.Structure):
    """scsfm_pair_desc of include/scsfm_hip.h."""
    _fields_ = [(n, _ct.c_void_p) for n in ("tgt_img", "ref_img", "tgt_depth", "ref_depth", "pose", "ws", "out",
                                            "g_tgt_depth", "g_ref_depth", "g_pose", "gbuf", "total", "hint")] + \
               [("depth_shift", _ct.c_int), ("g_tgt_img", _ct.c_void_p), ("g_ref_img", _ct.c_void_p)] + \
               [(n, _ct.c_void_p) for n in ("smooth_ws", "smooth_edge", "smooth_out", "smooth_total")]


def depth_shift(shape, B, H, W):
    """s >= 0 if ``shape`` is [B, 1, H >> s, W >> s] with H and W multiples of 2^s -- a depth map the pair kernels
    read in place, its nearest up-sampling to (H, W) (loss_functions.py:77-82) folded into their index arithmetic --
    else None."""
    if len(shape) != 4 or shape[0] != B or shape[1] != 1:
        return None
    for s in range(9):
        if (H >> s) << s != H or (W >> s) << s != W:
            return None
        if shape[2] == H >> s and shape[3] == W >> s:
            return s
    return None


def _pair_shift(dt, dr, B, H, W):
    """Both depth maps of a pair-direction are full resolution or the same coarser scale."""
    sh = dt.shape
    if sh == dr.shape and len(sh) == 4 and sh[3] == W and sh[2] == H and sh[0] == B and sh[1] == 1:
        return 0  # (the single-scale step: no search)
    s = depth_shift(dt.shape, B, H, W)
    if s is None or tuple(dr.shape) != tuple(dt.shape):
        check_sizes(dt, "depth", (B, 1, H, W))
        check_sizes(dr, "depth", (B, 1, H, W))
    return s


@_ft.lru_cache(maxsize=64)
def _sizes(lib, B, H, W):
    return (lib.size("scsfm_pair_ws_bytes", B, H, W), lib.size("scsfm_pair_bwd_scratch_bytes", B, H, W),
            lib.size("scsfm_smooth_ws_bytes", B, H, W))


def _pair_list(tgt_img, ref_imgs, tgt_depths, ref_depths, poses, poses_inv):
    """(tgt_img, ref_img, tgt_depth, ref_depth, pose, key_tgt, key_ref) per pair-direction, in the
    reference's order: for each ref, for each scale: tgt -> ref (loss_functions.py:84), ref -> tgt (:86)."""
    pairs = []
    for i, ref in enumerate(ref_imgs):
        for s, dt in enumerate(tgt_depths):
            dr = ref_depths[i][s]
            pairs.append((tgt_img, ref, dt, dr, poses[i], ("t", s), ("r", i, s)))
            pairs.append((ref, tgt_img, dr, dt, poses_inv[i], ("r", i, s), ("t", s)))
    return pairs


DEBUG_CHECK_WINDOW = 65536  # SCSFM_DEBUG_CHECK_WINDOW (include/scsfm_hip.h)


class WindowOverflow(RuntimeError):
    """A fixed-point cell of the speculative forward's scatter window wrapped (SCSFM_CHECK_WINDOW=1 runs only)."""


def smooth_rides_along(flags, tgt_img, tgt_depths, ref_depths, hint):
    """Can the speculative forward carry the smooth loss of the step's frames (scsfm_pair_desc::smooth_ws)?  It needs a
    speculative launch (``hint`` with a non-zero photo weight) and every frame's scale-0 depth map at full resolution --
    each frame is then the target of a pair-direction at depth_shift 0."""
    if hint is None or float(hint[0]) == 0.0:
        return False
    B, _, H, W = tgt_img.shape
    return all(tuple(m.shape) == (B, 1, H, W) for m in [tgt_depths[0]] + [r[0] for r in ref_depths])


def photo_geometry_fwd(lib, flags, tgt_img, K, ref_imgs, tgt_depths, ref_depths, poses, poses_inv, group=None,
                       hint=None, ws=None, hint_dev=None, check_window=False, smooth=False, keep_edges=True, step=None):
    """All pair-directions of loss_functions.py:56-90 in ONE call into the library.  ``tgt_depths[s]``
    and ``ref_depths[i][s]`` are full-resolution maps or, for a coarser scale, [B, 1, H >> k, W >> k] maps that the
    kernels read through the nearest up-sampling's index map (`depth_shift`).  Returns (photo, geom, outs [n_pairs, 8], ws)
    where ``ws`` (one tensor, n_pairs slices) must reach photo_geometry_bwd untouched.

    ``hint`` = (w_photo, w_geom): run the speculative forward (scsfm_pair_fwd_spec) -- the backward's
    tiled pass doubles as the forward and leaves its output planes in ``ws``'s tail for the backward,
    valid if the upstream gradients later stand in that ratio (checked on the device).  ``hint_dev``: a float64[2]
    tensor on the device holding that pair (scsfm_pair_desc::hint): the kernels read it instead of the host values, and
    photo_geometry_bwd -- given the same tensor -- leaves the upstream gradients it saw in it, so that the next
    forward speculates on the weights the training loop really uses.

    ``check_window`` (debugging; config.check_window(), i.e. SCSFM_CHECK_WINDOW=1): the speculative forward runs with
    SCSFM_DEBUG_CHECK_WINDOW -- the runtime-flag instantiation, every add into the scatter window a returning atomic --
    and this call synchronises and raises WindowOverflow if a cell wrapped (the gradients it would produce are wrong).

    ``group``: a torch.distributed process group -> exact data-parallel mode: the three raw sums of
    every pair are all-reduced (one [n_pairs, 3] collective) and the masked means are re-evaluated on
    the global sums, so every rank holds the losses of the concatenated batch.

    ``smooth`` (only where smooth_rides_along() holds): the speculative forward also evaluates compute_smooth_loss
    (loss_functions.py:132-159) of the frames [tgt, refs...] at scale 0 -- each frame in the tile of the first
    pair-direction whose TARGET it is -- and the call returns two more values: the smooth loss (the sum over the frames)
    and the smooth workspace ``sws``, laid out exactly as smooth_multi_fwd leaves it (smooth_multi_bwd and
    photo_geometry_bwd(smooth=...) take it from there; ``keep_edges`` as there).  ``step`` = (w_photo, w_smooth, w_geom)
    with ``smooth``: a seventh value, tensor[4] = {w_photo photo + w_smooth smooth + w_geom geometry, photo, smooth,
    geometry}, formed by the same finalize launch (scsfm_pairs_fwd_step; not in the exact data-parallel mode)."""
    B, _, H, W = tgt_img.shape
    pairs = _pair_list(tgt_img, ref_imgs, tgt_depths, ref_depths, poses, poses_inv)
    n = len(pairs)
    every = [tgt_img, K] + list(ref_imgs) + list(tgt_depths) + [d for r in ref_depths for d in r] + list(poses) + \
        list(poses_inv)
    _chk(*every)
    check_sizes(K, "intrinsics", (B, 3, 3))
    for r in ref_imgs:
        check_sizes(r, "ref_img", (B, 3, H, W))
    shifts = [_pair_shift(dt, dr, B, H, W) for _, _, dt, dr, _, _, _ in pairs]
    for p in list(poses) + list(poses_inv):
        check_sizes(p, "pose", (B, 6))
    ws_bytes, scratch_bytes, _ = _sizes(lib, B, H, W)
    spec = hint is not None and float(hint[0]) != 0.0
    stride = ws_bytes + (scratch_bytes if spec else 0)  # per pair: workspace, then (speculative) the gbuf planes
    if ws is None:
        ws = torch.empty(n * stride, dtype=torch.uint8, device=tgt_img.device)
    outs = torch.empty(n + 1, 8, dtype=tgt_img.dtype, device=tgt_img.device)  # per pair; last row: the totals
    descs = (PairDesc * n)()
    descs[0].total = outs.data_ptr() + n * outs.element_size() * 8
    if spec and hint_dev is not None:
        assert hint_dev.dtype == torch.float64 and hint_dev.numel() == 2 and hint_dev.device == tgt_img.device
        descs[0].hint = hint_dev.data_ptr()
    wp, op, esz = ws.data_ptr(), outs.data_ptr(), outs.element_size() * 8
    sws = souts = step_out = None
    if smooth:
        assert spec and smooth_rides_along(flags, tgt_img, tgt_depths, ref_depths, hint), "smooth=True needs a speculative forward on full-resolution maps"
        assert step is None or group is None, "the step total is not formed in the exact data-parallel mode"
        nf = 1 + len(ref_imgs)
        sw_bytes = _sizes(lib, B, H, W)[2]
        plane_bytes = ((B * H * W * tgt_img.element_size() + 255) // 256) * 256 if keep_edges else 0
        sws = torch.empty(nf * (sw_bytes + plane_bytes), dtype=torch.uint8, device=tgt_img.device)
        souts = torch.empty(nf + 1, dtype=tgt_img.dtype, device=tgt_img.device)  # one loss per frame, then their sum
        descs[0].smooth_total = souts.data_ptr() + nf * souts.element_size()
    seen = set()
    for j, (ti, ri, dt, dr, po, kt, _) in enumerate(pairs):
        d = descs[j]
        d.tgt_img, d.ref_img, d.tgt_depth, d.ref_depth, d.pose = ti.data_ptr(), ri.data_ptr(), dt.data_ptr(), \
            dr.data_ptr(), po.data_ptr()
        d.ws, d.out = wp + j * stride, op + j * esz
        d.gbuf = wp + j * stride + ws_bytes if spec else None
        d.depth_shift = shifts[j]
        if smooth and kt[-1] == 0 and kt not in seen:  # the first pair-direction whose target is this frame at scale 0
            seen.add(kt)
            f = 0 if kt[0] == "t" else 1 + kt[1]  # frame order of compute_smooth_loss: target, then the references
            d.smooth_ws = sws.data_ptr() + f * sw_bytes
            d.smooth_edge = sws.data_ptr() + nf * sw_bytes + f * plane_bytes if keep_edges else None
            d.smooth_out = souts.data_ptr() + f * souts.element_size()
    if smooth:
        assert len(seen) == nf
    MAX_PAIRS_PER_LAUNCH = 8  # (csrc: kMaxPairs; scsfm_pairs_fwd_step forms the total in ONE finalize launch)
    if smooth and step is not None and n <= MAX_PAIRS_PER_LAUNCH:
        step_out = torch.empty(4, dtype=tgt_img.dtype, device=tgt_img.device)
        lib.call(f"scsfm_pairs_fwd_step_{_suffix(tgt_img)}", n, _ct.addressof(descs), B, H, W, _p(K),
                 flags | (DEBUG_CHECK_WINDOW if check_window else 0), float(step[0]), float(step[1]), float(step[2]),
                 _p(step_out), _stream(tgt_img))
    else:
        lib.call(f"scsfm_pairs_fwd_{_suffix(tgt_img)}", n, _ct.addressof(descs), B, H, W, _p(K),
                 flags | (DEBUG_CHECK_WINDOW if check_window else 0),
                 float(hint[0]) if spec else 0.0, float(hint[1]) if spec else 0.0, _stream(tgt_img))
    if check_window:
        wrapped_t = outs[:n, 7].sum().reshape(1)
        if group is not None:
            # every rank must take the same branch: a rank that raised alone would leave the others hanging in the
            # all-reduce of the pair sums below (round-5 advisor finding)
            import torch.distributed as dist
            dist.all_reduce(wrapped_t, op=dist.ReduceOp.MAX, group=group)
        wrapped = int(wrapped_t.item())  # (synchronises: a debugging mode)
        if wrapped:
            raise WindowOverflow(f"scsfm_hip: {wrapped} fixed-point cell(s) of the scatter window wrapped although the unit of every "
                                 "tile's cells follows from an upper bound of what the tile can add to one cell "
                                 "(csrc/scsfm_geom.h: win_units_of): the bound is violated -- a bug, please report the inputs; "
                                 "the depth gradients of this step are wrong")
    if group is not None:
        import torch.distributed as dist
        sums = outs[:n, 2:5].contiguous()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        outs[:n, 2:5] = sums
        for j in range(n):
            pair_refinalize(lib, (B, H, W), ws[j * stride:j * stride + ws_bytes], outs[j])
    if smooth and step is not None and step_out is None:
        # more pair-directions than one launch holds (several scales): the weighted sum in a launch of its own
        step_out = step_total(lib, outs[n, :2], souts[nf:nf + 1], *step)
    extra = () if not smooth else ((souts[nf], sws) if step is None else (souts[nf], sws, step_out))
    if flags & 16384:  # SCSFM_DEBUG_KERNEL_ONLY (bench.py): nothing was finalised
        return (None, None, outs[:n], ws) + extra
    if group is not None:
        tot = outs[:n, :2].sum(dim=0)  # the re-finalised losses
        return (tot[0], tot[1], outs[:n], ws) + extra
    # plain sums over refs, scales and directions (loss_functions.py:89-90), left in the last row by the library
    return (outs[n, 0], outs[n, 1], outs[:n], ws) + extra


def window_overflows(lib, ws, n_pairs, B, H, W, spec=True):
    """Wraps of fixed-point scatter cells counted by launches with SCSFM_DEBUG_CHECK_WINDOW since the last forward on
    ``ws`` (one int per pair-direction; reads the device: synchronises)."""
    ws_bytes, scratch_bytes, _ = _sizes(lib, B, H, W)
    stride = ws_bytes + (scratch_bytes if spec else 0)
    return [int(ws[j * stride + 256 * B + 104:j * stride + 256 * B + 108].view(torch.int32).item()) for j in range(n_pairs)]


def photo_geometry_bwd(lib, flags, tgt_img, K, ref_imgs, tgt_depths, ref_depths, poses, poses_inv, ws, g_photo,
                       g_geom, hint_dev=None, need_imgs=None, need_K=False, smooth=None, check_window=False):
    """Gradients of photo_geometry_fwd's two sums: (g_tgt_depths[s], g_ref_depths[i][s], g_poses[i],
    g_poses_inv[i]).  Each depth map's gradient buffer receives the sum over every pair-direction that
    touches it (dense as target, scattered as reference) from the library's combining kernel, which
    stores (no zero-fill); one call into the library, one shared scratch buffer.
    ``need_imgs`` (one bool per image: target, then the references) / ``need_K``: also the gradients of the data inputs
    -> two more results, ([g_tgt_img, g_ref_imgs...] with None where not wanted, g_K).
    ``smooth`` = (sws, g_smooth): the workspace of smooth_multi_fwd over the frames [tgt, refs...] at scale 0 (edge
    planes kept) and the smooth term's upstream gradient -> its gradient is added to the scale-0 depth gradients by the
    same combining pass that stores them (scsfm_pairs_bwd_smooth), no smooth_multi_bwd launch."""
    B, _, H, W = tgt_img.shape
    pairs = _pair_list(tgt_img, ref_imgs, tgt_depths, ref_depths, poses, poses_inv)
    n = len(pairs)
    ws_bytes, scratch_bytes, _ = _sizes(lib, B, H, W)
    spec = ws.numel() == n * (ws_bytes + scratch_bytes)  # the forward was speculative: gbuf follows each workspace
    stride = ws_bytes + (scratch_bytes if spec else 0)
    # one allocation for every depth-gradient buffer (each of its map's shape, 64-byte aligned); the library
    # stores into them
    maps = list(tgt_depths) + [d for r in ref_depths for d in r]
    if len(tgt_depths) == 1:
        g_maps = torch.empty((len(maps),) + tuple(maps[0].shape), dtype=tgt_img.dtype, device=tgt_img.device).unbind(0)
    else:
        starts, total = [], 0
        for m in maps:
            starts.append(total)
            total += (m.numel() + 15) // 16 * 16
        g_all = torch.empty(total, dtype=tgt_img.dtype, device=tgt_img.device)
        g_maps = [g_all[o:o + m.numel()].view(m.shape) for o, m in zip(starts, maps)]
    g_td = g_maps[:len(tgt_depths)]
    g_rd = [g_maps[len(tgt_depths) * (1 + i):len(tgt_depths) * (2 + i)] for i in range(len(ref_depths))]
    g_pose_all = torch.empty(n, B, 6, dtype=tgt_img.dtype, device=tgt_img.device)
    # one private scratch region per pair (they run concurrently); the speculative forward already
    # placed it behind each pair's workspace
    scratch = None if spec else torch.empty(n * scratch_bytes, dtype=torch.uint8, device=tgt_img.device)

    def gbuf(key):
        return g_td[key[1]] if key[0] == "t" else g_rd[key[1]][key[2]]

    descs = (PairDesc * n)()
    if hint_dev is not None:
        descs[0].hint = hint_dev.data_ptr()
    wp, gp, psz = ws.data_ptr(), g_pose_all.data_ptr(), g_pose_all.element_size() * B * 6
    for j, (ti, ri, dt, dr, po, kt, kr) in enumerate(pairs):
        d = descs[j]
        d.tgt_img, d.ref_img, d.tgt_depth, d.ref_depth, d.pose = ti.data_ptr(), ri.data_ptr(), dt.data_ptr(), \
            dr.data_ptr(), po.data_ptr()
        d.ws = wp + j * stride
        d.gbuf = wp + j * stride + ws_bytes if spec else None
        d.g_tgt_depth, d.g_ref_depth, d.g_pose = gbuf(kt).data_ptr(), gbuf(kr).data_ptr(), gp + j * psz
        d.depth_shift = _pair_shift(dt, dr, B, H, W)
    bflags = flags | (DEBUG_CHECK_WINDOW if check_window else 0)
    if smooth is not None:
        sws, g_smooth = smooth
        nf = 1 + len(ref_depths)
        sw_bytes = _sizes(lib, B, H, W)[2]
        plane_bytes = ((B * H * W * tgt_img.element_size() + 255) // 256) * 256
        assert sws.numel() == nf * (sw_bytes + plane_bytes), "smooth workspace without edge planes"
        frames = [g_td[0]] + [r[0] for r in g_rd]
        grads = _ptr_array(frames)
        edges = (_ct.c_void_p * nf)(*[sws.data_ptr() + nf * sw_bytes + i * plane_bytes for i in range(nf)])
        stats = (_ct.c_void_p * nf)(*[sws.data_ptr() + i * sw_bytes for i in range(nf)])
        lib.call(f"scsfm_pairs_bwd_smooth_{_suffix(tgt_img)}", n, _ct.addressof(descs), B, H, W, _p(K), bflags, _p(scratch),
                 _p(g_photo), _p(g_geom), nf, grads, edges, stats, _p(g_smooth), _stream(tgt_img))
    else:
        lib.call(f"scsfm_pairs_bwd_{_suffix(tgt_img)}", n, _ct.addressof(descs), B, H, W, _p(K), bflags, _p(scratch),
                 _p(g_photo), _p(g_geom), _stream(tgt_img))
    if check_window:  # (debugging: the speculative tail counted into these words; synchronises)
        wrapped = sum(window_overflows(lib, ws, n, B, H, W, spec))
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # data parallel: every rank raises, or none -- a rank that raised alone would leave the others in the gradient
            # all-reduce that follows this backward
            t = torch.tensor([float(wrapped)], device=tgt_img.device if tgt_img.is_cuda else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wrapped = int(t.item())
        if wrapped:
            raise WindowOverflow(f"scsfm_hip: fixed-point cells of a scatter window wrapped {wrapped} time(s) in this step's "
                                 "forward although their unit is bounded per tile (csrc/scsfm_geom.h: win_units_of): a bug, "
                                 "please report the inputs; depth gradients may be off")
    g_inputs = None
    if need_imgs is not None or need_K:
        # gradients of the data inputs (scsfm_pairs_bwd_inputs): images accumulate, intrinsics are stored
        imgs = [tgt_img] + list(ref_imgs)
        want = list(need_imgs) if need_imgs is not None else [False] * len(imgs)
        g_imgs = [torch.zeros_like(im) if w else None for im, w in zip(imgs, want)]
        g_K = torch.empty_like(K) if need_K else None
        # by POSITION in the argument list (key = ("t", s) -> image 0, ("r", i, s) -> image 1 + i), not by storage: the
        # same tensor passed twice (a reference frame repeated, the target among the references) gets one gradient
        # per argument, as the reference's autograd would return, and autograd adds them up
        pos = lambda key: 0 if key[0] == "t" else 1 + key[1]
        for j, (_ti, _ri, _dt, _dr, _po, kt, kr) in enumerate(pairs):
            descs[j].g_tgt_img = _p(g_imgs[pos(kt)]) or None
            descs[j].g_ref_img = _p(g_imgs[pos(kr)]) or None
        lib.call(f"scsfm_pairs_bwd_inputs_{_suffix(tgt_img)}", n, _ct.addressof(descs), B, H, W, _p(K), flags,
                 _p(g_photo), _p(g_geom), _p(g_K), _stream(tgt_img))
        g_inputs = (g_imgs, g_K)
    n_scales = len(tgt_depths)
    # pair j = 2 * (i * n_scales + s) + direction; a pose feeds every scale of its ref
    per = g_pose_all.view(len(ref_imgs), n_scales, 2, B, 6).sum(dim=1) if n_scales > 1 else \
        g_pose_all.view(len(ref_imgs), 2, B, 6)
    g_poses = [per[i, 0] for i in range(len(ref_imgs))]
    g_poses_inv = [per[i, 1] for i in range(len(ref_imgs))]
    if g_inputs is not None:
        return g_td, g_rd, g_poses, g_poses_inv, g_inputs[0], g_inputs[1]
    return g_td, g_rd, g_poses, g_poses_inv


def _ptr_array(tensors):
    arr = (_ct.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def smooth_multi_fwd(lib, depths, imgs, keep_edges=True, step=None):
    """compute_smooth_loss (loss_functions.py:154-159): sum over frames -> (loss, ws).  With
    ``keep_edges`` the forward also leaves every pixel's summed edge terms behind ``ws`` (one fp plane per
    frame) and the backward becomes a pure stream."""
    _chk(*depths, *imgs)
    B, _, H, W = imgs[0].shape
    n = len(depths)
    for d, im in zip(depths, imgs):
        check_sizes(d, "depth", (B, 1, H, W))
        check_sizes(im, "img", (B, 3, H, W))
    ws_bytes = _sizes(lib, B, H, W)[2]
    plane_bytes = ((B * H * W * imgs[0].element_size() + 255) // 256) * 256 if keep_edges else 0
    ws = torch.empty(n * (ws_bytes + plane_bytes), dtype=torch.uint8, device=imgs[0].device)
    outs = torch.empty(n + 1, dtype=imgs[0].dtype, device=imgs[0].device)  # one loss per frame, then their sum
    edges = (_ct.c_void_p * n)(*[ws.data_ptr() + n * ws_bytes + i * plane_bytes for i in range(n)]) if keep_edges else None
    if step is not None:  # (photo_geom[2], w_photo, w_smooth, w_geom): the step's objective in the same launch
        pg, w1, w2, w3 = step
        step_out = torch.empty(4, dtype=imgs[0].dtype, device=imgs[0].device)
        lib.call(f"scsfm_smooth_multi_fwd_step_{_suffix(imgs[0])}", n, _ptr_array(depths), _ptr_array(imgs), B, H, W, _p(ws),
                 edges, _p(outs), _p(pg), float(w1), float(w2), float(w3), _p(step_out), _stream(imgs[0]))
        return outs[n], ws, step_out
    lib.call(f"scsfm_smooth_multi_fwd_{_suffix(imgs[0])}", n, _ptr_array(depths), _ptr_array(imgs), B, H, W, _p(ws),
             edges, _p(outs), _stream(imgs[0]))
    return outs[n], ws


def smooth_multi_bwd(lib, depths, imgs, ws, g_loss, need=None, into=None):
    """-> list of dL/d depth (None where ``need[i]`` is False).  ``into``: existing gradient buffers (one per
    frame) that the smooth gradient is ADDED to instead of being stored into fresh ones."""
    B, _, H, W = imgs[0].shape
    n = len(depths)
    ws_bytes = _sizes(lib, B, H, W)[2]
    plane_bytes = ((B * H * W * imgs[0].element_size() + 255) // 256) * 256
    has_edges = ws.numel() == n * (ws_bytes + plane_bytes)
    edges = (_ct.c_void_p * n)(*[ws.data_ptr() + n * ws_bytes + i * plane_bytes for i in range(n)]) if has_edges else None
    if into is not None:
        grads = [into[i] if (need is None or need[i]) else None for i in range(n)]
    else:
        # the library stores every pixel of a wanted gradient: no zero-fill
        g_all = torch.empty((n,) + tuple(depths[0].shape), dtype=depths[0].dtype, device=depths[0].device)
        grads = [g_all[i] if (need is None or need[i]) else None for i in range(n)]
    lib.call(f"scsfm_smooth_multi_bwd_{_suffix(imgs[0])}", n, _ptr_array(depths), _ptr_array(imgs), B, H, W, _p(ws),
             edges, _p(g_loss), _ptr_array(grads), 1 if into is not None else 0, _stream(imgs[0]))
    return grads


def smooth_multi_bwd_images(lib, depths, imgs, ws, g_loss, need, into=None):
    """-> list of dL/d img (None where ``need[i]`` is False).  ``into``: existing buffers the gradient is ADDED to."""
    B, _, H, W = imgs[0].shape
    n = len(depths)
    if into is not None:
        grads = [into[i] if need[i] else None for i in range(n)]
    else:
        grads = [torch.empty_like(imgs[i]) if need[i] else None for i in range(n)]
    lib.call(f"scsfm_smooth_multi_bwd_images_{_suffix(imgs[0])}", n, _ptr_array(depths), _ptr_array(imgs), B, H, W,
             _p(ws), _p(g_loss), _ptr_array(grads), 1 if into is not None else 0, _stream(imgs[0]))
    return grads


def step_total(lib, photo_geom, smooth, w_photo, w_smooth, w_geom):
    """-> tensor[4] = {w_photo * photo + w_smooth * smooth + w_geom * geometry, photo, smooth, geometry}."""
    out = torch.empty(4, dtype=photo_geom.dtype, device=photo_geom.device)
    lib.call(f"scsfm_step_total_{_suffix(out)}", _p(photo_geom), _p(smooth), float(w_photo), float(w_smooth),
             float(w_geom), _p(out), _stream(out))
    return out


def step_weights(lib, g_loss, w_photo, w_smooth, w_geom):
    """-> tensor[3] = {w_photo * g, w_geom * g, w_smooth * g}: the upstream gradients of the three terms."""
    out = torch.empty(3, dtype=g_loss.dtype, device=g_loss.device)
    lib.call(f"scsfm_step_weights_{_suffix(out)}", _p(g_loss), float(w_photo), float(w_smooth), float(w_geom), _p(out),
             _stream(out))
    return out
