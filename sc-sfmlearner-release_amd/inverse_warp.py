"""Drop-in for the reference's ``inverse_warp`` module, backed by hand-written HIP kernels.

Same public names, positional signatures and error behaviour as the reference
(/root/reference/inverse_warp.py); put this directory in front of the reference on PYTHONPATH and
``from inverse_warp import *`` (test_vo.py:13) resolves here.  All tensors must be HIP ('cuda')
tensors -- there is no CPU path in this package.
"""
from __future__ import division

import torch
import torch.nn.functional as F  # noqa: F401  (part of the reference module's namespace: `from inverse_warp import *`)

from scsfm_hip import capi as _capi, ops as _ops  # (underscored: a star import must bring the reference's names only)

pixel_coords = None  # kept for API compatibility (inverse_warp.py:5); the kernels need no cached grid


def set_id_grid(depth):
    """inverse_warp.py:8-17: the reference caches a [1,3,H,W] grid (x = column, y = row, 1) in a
    module global.  The HIP kernels derive pixel coordinates from thread indices; this function only
    reproduces the cache for callers that read ``pixel_coords``."""
    global pixel_coords
    b, h, w = depth.size()
    i_range = torch.arange(0, h, device=depth.device).view(1, h, 1).expand(1, h, w).type_as(depth)
    j_range = torch.arange(0, w, device=depth.device).view(1, 1, w).expand(1, h, w).type_as(depth)
    pixel_coords = torch.stack((j_range, i_range, torch.ones_like(i_range)), dim=1)


def check_sizes(input, input_name, expected):
    """inverse_warp.py:20-26 -- `expected` is a string such as 'B3HW'; digits are checked."""
    condition = [input.ndimension() == len(expected)]
    for i, size in enumerate(expected):
        if size.isdigit():
            condition.append(input.size(i) == int(size))
    assert (all(condition)), "wrong size for {}, expected {}, got  {}".format(
        input_name, 'x'.join(expected), list(input.size()))


def pose_vec2mat(vec, rotation_mode='euler'):
    """[B,6] (tx,ty,tz,rx,ry,rz) -> [B,3,4] (inverse_warp.py:139-154).  HIP forward + backward."""
    if rotation_mode not in ('euler', 'quat'):
        # the reference falls through to an UnboundLocalError here (inverse_warp.py:149-153)
        raise UnboundLocalError("rotation_mode must be 'euler' or 'quat', got {!r}".format(rotation_mode))
    check_sizes(vec, 'pose', 'B6')
    return _ops.PoseVec2Mat.apply(vec, rotation_mode)


def euler2mat(angle):
    """[B,3] -> [B,3,3], R = Rx Ry Rz (inverse_warp.py:77-112)."""
    vec = torch.cat([torch.zeros_like(angle), angle], dim=1)
    return _ops.PoseVec2Mat.apply(vec, 'euler')[:, :, :3]


def quat2mat(quat):
    """[B,3] -> [B,3,3] (inverse_warp.py:115-136)."""
    vec = torch.cat([torch.zeros_like(quat), quat], dim=1)
    return _ops.PoseVec2Mat.apply(vec, 'quat')[:, :, :3]


def inverse_warp2(img, depth, ref_depth, pose, intrinsics, padding_mode='zeros'):
    """inverse_warp.py:230-269 -> (projected_img, valid_mask, projected_depth, computed_depth).

    One HIP kernel forward (back-projection, SE(3), projection, both bilinear gathers), one
    backward (dense dL/d depth, atomic scatter into dL/d ref_depth, per-batch reduction to dL/d
    pose).  The training loss does not call this -- it uses the fused pair kernels -- but the maps
    are part of the reference's public API."""
    check_sizes(img, 'img', 'B3HW')
    check_sizes(depth, 'depth', 'B1HW')
    check_sizes(ref_depth, 'ref_depth', 'B1HW')
    check_sizes(pose, 'pose', 'B6')
    check_sizes(intrinsics, 'intrinsics', 'B33')
    flags = _capi.make_flags(padding_mode=padding_mode)
    return _ops.InverseWarp2.apply(flags, img, depth, ref_depth, pose, intrinsics)


def inverse_warp(img, depth, pose, intrinsics, rotation_mode='euler', padding_mode='zeros'):
    """Legacy single-view warp (inverse_warp.py:157-191): depth is [B,H,W]; returns
    (projected_img, valid_points[bool]).  Imported but never called by the reference's loss
    (loss_functions.py:5).  Served by the same HIP kernel as inverse_warp2 (euler or quaternion rotation)."""
    check_sizes(img, 'img', 'B3HW')
    check_sizes(depth, 'depth', 'BHW')
    check_sizes(pose, 'pose', 'B6')
    check_sizes(intrinsics, 'intrinsics', 'B33')
    if rotation_mode not in ('euler', 'quat'):
        raise UnboundLocalError("rotation_mode must be 'euler' or 'quat', got {!r}".format(rotation_mode))
    d = depth.unsqueeze(1)
    # the legacy path has no zeros-mode coordinate overwrite (cam2pixel, inverse_warp.py:47-74)
    flags = _capi.make_flags(padding_mode=padding_mode) | _capi.LEGACY_GRID
    if rotation_mode == 'quat':
        flags |= _capi.ROT_QUAT_FLAG
    projected_img, valid, _, _ = _ops.InverseWarp2.apply(flags, img, d, d.detach(), pose, intrinsics)
    return projected_img, valid.squeeze(1) > 0.5


def pixel2cam(depth, intrinsics_inv):
    """inverse_warp.py:29-44: depth [B,H,W], intrinsics_inv [B,3,3] -> camera coordinates [B,3,H,W] =
    K^-1 (u, v, 1) * depth.  One HIP kernel each way (gradient to the depth map)."""
    check_sizes(depth, 'depth', 'BHW')
    check_sizes(intrinsics_inv, 'intrinsics_inv', 'B33')
    return _ops.Pixel2Cam.apply(depth, intrinsics_inv)


def _check_c2p(cam_coords, rot, tr):
    """The kernels index the rotation as 9 and the translation as 3 contiguous scalars per batch element: any other shape
    (a [B,3,4] matrix, a [B,3] vector, another batch size) would be read with the wrong stride where the reference's
    matmul / broadcast raises.  Same message format as check_sizes (inverse_warp.py:20-26)."""
    B = cam_coords.size(0)
    if rot is not None:
        check_sizes(rot, 'proj_c2p_rot', 'B33')
        assert rot.size(0) == B, "wrong size for proj_c2p_rot, expected {}x3x3, got {}".format(B, list(rot.size()))
    if tr is not None:
        assert list(tr.size()) == [B, 3, 1], "wrong size for proj_c2p_tr, expected {}x3x1, got {}".format(B, list(tr.size()))


def cam2pixel(cam_coords, proj_c2p_rot, proj_c2p_tr, padding_mode):
    """inverse_warp.py:47-74: camera coordinates [B,3,H,W], rotation [B,3,3] | None, translation [B,3,1] | None ->
    normalised sampling grid [B,H,W,2].  (``padding_mode`` is unused by the reference too.)"""
    check_sizes(cam_coords, 'cam_coords', 'B3HW')
    _check_c2p(cam_coords, proj_c2p_rot, proj_c2p_tr)
    return _ops.Cam2Pixel.apply(0, False, cam_coords, proj_c2p_rot, proj_c2p_tr)


def cam2pixel2(cam_coords, proj_c2p_rot, proj_c2p_tr, padding_mode):
    """inverse_warp.py:194-227 -> (grid [B,H,W,2], computed depth [B,1,H,W]); under 'zeros' padding out-of-range
    coordinates are overwritten with 2 (and carry no gradient)."""
    check_sizes(cam_coords, 'cam_coords', 'B3HW')
    _check_c2p(cam_coords, proj_c2p_rot, proj_c2p_tr)
    flags = _capi.C2P_OVERWRITE if padding_mode == 'zeros' else 0
    return _ops.Cam2Pixel.apply(flags, True, cam_coords, proj_c2p_rot, proj_c2p_tr)
