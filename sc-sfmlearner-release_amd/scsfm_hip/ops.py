"""torch.autograd.Function wrappers around the HIP kernels (through the C ABI, ``capi.py``).

These are the product path: CUDA (HIP) tensors only, no eager / CPU fallback -- a missing
libscsfm_hip.so or a CPU tensor raises.  Host-side there is no synchronisation: the reference's
``if mask.sum() > 10000`` (loss_functions.py:125) is evaluated on the device.
"""
from __future__ import annotations

import weakref

import torch

from . import _lib, capi


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("scsfm_hip: tensors must live on a HIP device (torch 'cuda'); "
                               "this package has no CPU fallback")


def _c(t):
    return t.contiguous()


def _scalar(g, like):
    if g is None:
        return torch.zeros(1, dtype=like.dtype, device=like.device)
    return g.reshape(1).contiguous()


# ------------------------------------------------------------------------------------------------
# compute_photo_and_geometry_loss: all refs x scales x both directions behind one autograd node
# ------------------------------------------------------------------------------------------------
class _SmoothStash:
    """What the speculative forward of compute_photo_and_geometry_loss left for the compute_smooth_loss call that
    train.py:262-266 makes next on the SAME frames: the smooth loss as a THIRD OUTPUT of the PhotoGeometryLoss node
    (loss_functions.compute_photo_and_geometry_loss hands the caller the first two, as the reference does, and parks the
    third here).  compute_smooth_loss returns that very tensor: its gradient then arrives in the same backward call as
    the pair losses' and the smooth term's depth gradients are added by the combining pass that stores theirs
    (scsfm_pairs_bwd_smooth) -- no smooth backward launch, no second gradient per depth map for autograd to add.
    Autograd-correct for every use: a caller who differentiates only one of the losses gets None for the others'
    upstream gradients, which count as zero.  Keyed on the identity of the tensor OBJECTS (weak references: a new
    tensor at a recycled address is another object) and their version counters (an in-place write in between invalidates
    the entry); consumed by the first matching call, replaced by the next forward.  A miss costs nothing but the
    stand-alone smooth forward."""
    slot = None

    @staticmethod
    def key(depths, imgs):
        return [(weakref.ref(t), t._version) for t in list(depths) + list(imgs)]

    @classmethod
    def put(cls, depths, imgs, loss):
        cls.slot = (cls.key(depths, imgs), loss, _stream_id(imgs[0]), torch.is_grad_enabled())

    @classmethod
    def take(cls, depths, imgs):
        slot, cls.slot = cls.slot, None
        if slot is None:
            return None
        keys, loss, stream, grad_mode = slot
        ts = list(depths) + list(imgs)
        if len(keys) != len(ts) or stream != _stream_id(imgs[0]) or grad_mode != torch.is_grad_enabled():
            return None
        for (ref, ver), t in zip(keys, ts):
            if ref() is not t or t._version != ver:
                return None
        return loss


def _stream_id(t):
    return capi._stream(t)


class PhotoGeometryLoss(torch.autograd.Function):
    """forward(flags, n_ref, n_scales, tgt_img, K, *ref_imgs, *tgt_depths, *ref_depths, *poses,
    *poses_inv) -> (photo_loss, geometry_loss, smooth_loss | None)

    The third output (round 6): compute_smooth_loss of the frames [tgt, refs...] at scale 0, evaluated by the speculative
    forward on the way (scsfm_pair_desc::smooth_ws) whenever it speculates on full-resolution maps; None otherwise.

    Depth maps are full resolution (scale s > 0 is nearest-upsampled by the caller, under
    autograd).  ref_depths is flattened ref-major: ref_depths[i * n_scales + s].
    Gradients: every depth map and every pose; images and K too when they require grad (train.py never asks).
    """

    @staticmethod
    def _split(rest, n_ref, n_scales):
        o = 0
        ref_imgs = list(rest[o:o + n_ref]); o += n_ref
        tgt_depths = list(rest[o:o + n_scales]); o += n_scales
        ref_depths = [list(rest[o + i * n_scales:o + (i + 1) * n_scales]) for i in range(n_ref)]; o += n_ref * n_scales
        poses = list(rest[o:o + n_ref]); o += n_ref
        poses_inv = list(rest[o:o + n_ref]); o += n_ref
        return ref_imgs, tgt_depths, ref_depths, poses, poses_inv, o

    @staticmethod
    def forward(ctx, flags, n_ref, n_scales, tgt_img, K, *rest):
        from . import config as _config, dist as _dist
        lib = _lib.get()
        rest = [_c(t) for t in rest]
        tgt_img, K = _c(tgt_img), _c(K)
        _need_cuda(tgt_img, K, *rest)
        ref_imgs, tgt_depths, ref_depths, poses, poses_inv, _ = PhotoGeometryLoss._split(rest, n_ref, n_scales)
        # speculate only when a backward can follow (some depth / pose requires grad)
        hint = _config.weight_hint() if any(ctx.needs_input_grad) else None
        # the pair the kernels speculate on lives on the device: every backward leaves the upstream gradients it saw there
        hint_dev = _config.hint_tensor(tgt_img.device) if hint is not None else None
        # the smooth loss of the same frames (the call train.py:262-266 makes next) rides in the speculative tiles
        ride = _config.smooth_rides_along() and capi.smooth_rides_along(flags, tgt_img, tgt_depths, ref_depths, hint)
        res = capi.photo_geometry_fwd(lib, flags, tgt_img, K, ref_imgs, tgt_depths, ref_depths, poses,
                                      poses_inv, group=_dist.exact_group(), hint=hint, hint_dev=hint_dev,
                                      check_window=hint is not None and _config.check_window(), smooth=ride)
        photo, geom, _, ws = res[:4]
        smooth, sws = (res[4], res[5]) if ride else (None, None)
        ctx.flags, ctx.n_ref, ctx.n_scales = flags, n_ref, n_scales
        ctx.hint_dev = hint_dev
        ctx.rides = ride
        ctx.set_materialize_grads(False)  # an output nobody differentiates arrives as None, not as a zero tensor
        ctx.save_for_backward(tgt_img, K, *rest, ws, *([sws] if ride else []))
        return photo, geom, smooth

    @staticmethod
    def backward(ctx, g_photo, g_geom, g_smooth=None):
        from . import config as _config
        lib = _lib.get()
        n_ref, n_scales, flags = ctx.n_ref, ctx.n_scales, ctx.flags
        saved = ctx.saved_tensors
        tgt_img, K = saved[0], saved[1]
        ref_imgs, tgt_depths, ref_depths, poses, poses_inv, n_in = PhotoGeometryLoss._split(saved[2:], n_ref, n_scales)
        ws = saved[2 + n_in]
        # the smooth loss of this node was used (compute_smooth_loss found it in the stash): its depth gradients ride in
        # the combining pass of the pair terms
        smooth = (saved[3 + n_in], _scalar(g_smooth, tgt_img)) if (ctx.rides and g_smooth is not None) else None
        # images and intrinsics are data to train.py; the reference's autograd reaches them all the same, and so does
        # this node when asked (one extra tiled pass for the images, a reduction for K)
        need_imgs = [ctx.needs_input_grad[3]] + list(ctx.needs_input_grad[5:5 + n_ref])
        need_K = ctx.needs_input_grad[4]
        res = capi.photo_geometry_bwd(
            lib, flags, tgt_img, K, ref_imgs, tgt_depths, ref_depths, poses, poses_inv, ws,
            _scalar(g_photo, tgt_img), _scalar(g_geom, tgt_img), hint_dev=ctx.hint_dev,
            need_imgs=need_imgs if any(need_imgs) else None, need_K=need_K, smooth=smooth, check_window=_config.check_window())
        g_td, g_rd, g_poses, g_poses_inv = res[:4]
        g_imgs, g_K = res[4:] if len(res) > 4 else ([None] * (1 + n_ref), None)
        if smooth is not None and any(need_imgs):  # (the data inputs: the smooth term reaches the images through its edge weights)
            frames = [tgt_depths[0]] + [r[0] for r in ref_depths]
            capi.smooth_multi_bwd_images(lib, frames, [tgt_img] + list(ref_imgs), smooth[0], smooth[1], need_imgs, into=g_imgs)
        return (None, None, None, g_imgs[0], g_K, *g_imgs[1:], *g_td, *[g for r in g_rd for g in r], *g_poses,
                *g_poses_inv)


class PairwiseLoss(torch.autograd.Function):
    """compute_pairwise_loss (loss_functions.py:95-119), one pair-direction -> (photo, geom)."""

    @staticmethod
    def forward(ctx, flags, tgt_img, ref_img, tgt_depth, ref_depth, pose, K):
        lib = _lib.get()
        args = [_c(t) for t in (tgt_img, ref_img, tgt_depth, ref_depth, pose, K)]
        _need_cuda(*args)
        out = torch.empty(8, dtype=args[0].dtype, device=args[0].device)
        ws = capi.pair_fwd_into(lib, *args, flags, out)
        ctx.flags = flags
        ctx.save_for_backward(*args, ws)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_photo, g_geom):
        lib = _lib.get()
        tgt_img, ref_img, tgt_depth, ref_depth, pose, K, ws = ctx.saved_tensors
        need = ctx.needs_input_grad
        res = capi.pair_bwd(lib, tgt_img, ref_img, tgt_depth, ref_depth, pose, K, ctx.flags, ws,
                            _scalar(g_photo, tgt_img), _scalar(g_geom, tgt_img), need_tgt_img=need[1], need_ref_img=need[2],
                            need_K=need[6])
        g_td, g_rd, g_pose = res[:3]
        g_ti, g_ri, g_K = res[3:] if len(res) > 3 else (None, None, None)
        return None, g_ti, g_ri, g_td, g_rd, g_pose, g_K


# ------------------------------------------------------------------------------------------------
# compute_smooth_loss: every frame behind one autograd node
# ------------------------------------------------------------------------------------------------
class SmoothLoss(torch.autograd.Function):
    """forward(n, *depths, *imgs) -> sum_i get_smooth_loss(depths[i], imgs[i])
    (loss_functions.py:132-159)."""

    @staticmethod
    def forward(ctx, n, *rest):
        lib = _lib.get()
        rest = [_c(t) for t in rest]
        _need_cuda(*rest)
        depths, imgs = rest[:n], rest[n:]
        loss, ws = capi.smooth_multi_fwd(lib, depths, imgs, keep_edges=any(ctx.needs_input_grad[1:1 + n]))
        ctx.n = n
        ctx.save_for_backward(*rest, ws)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.get()
        n = ctx.n
        saved = ctx.saved_tensors
        depths, imgs, ws = saved[:n], saved[n:2 * n], saved[2 * n]
        gs = _scalar(g, imgs[0])
        grads = capi.smooth_multi_bwd(lib, depths, imgs, ws, gs, ctx.needs_input_grad[1:1 + n]) \
            if any(ctx.needs_input_grad[1:1 + n]) else [None] * n
        need_img = ctx.needs_input_grad[1 + n:]
        g_imgs = capi.smooth_multi_bwd_images(lib, depths, imgs, ws, gs, need_img) if any(need_img) else [None] * n
        return (None, *grads, *g_imgs)


class StepLoss(torch.autograd.Function):
    """forward(flags, n_ref, n_scales, w_photo, w_smooth, w_geom, tgt_img, K, *ref_imgs, *tgt_depths, *ref_depths,
    *poses, *poses_inv) -> (loss, photo, smooth, geometry) with loss = w_photo * photo + w_smooth * smooth +
    w_geom * geometry (train.py:259-268) behind ONE autograd node; photo / smooth / geometry are returned for
    logging only (not differentiable).

    Compared with the reference's two functions plus the weighted sum this saves what autograd does between
    them: five scalar kernels forward, three backward, and one elementwise addition per depth map (the smooth
    gradient is accumulated into the buffers the pair backward just stored).  The speculation hint is the
    weights themselves -- this node knows them, unlike PhotoGeometryLoss, which learns the ratio from the upstream
    gradients it sees and therefore keeps it on the device (config.hint_tensor) -- so the speculation holds whatever
    the upstream gradient of the loss is (the check is on the RATIO of the two terms' gradients)."""

    @staticmethod
    def forward(ctx, flags, n_ref, n_scales, w_photo, w_smooth, w_geom, tgt_img, K, *rest):
        import numpy as np
        from . import dist as _dist
        lib = _lib.get()
        rest = [_c(t) for t in rest]
        tgt_img, K = _c(tgt_img), _c(K)
        _need_cuda(tgt_img, K, *rest)
        ref_imgs, tgt_depths, ref_depths, poses, poses_inv, _ = PhotoGeometryLoss._split(rest, n_ref, n_scales)
        hint = (float(np.float32(w_photo)), float(np.float32(w_geom))) if (w_photo != 0 and any(ctx.needs_input_grad)) else None
        from . import config as _config
        group = _dist.exact_group()
        ride = group is None and _config.smooth_rides_along() and \
            capi.smooth_rides_along(flags, tgt_img, tgt_depths, ref_depths, hint)
        if ride:
            # one launch sequence for the whole forward: the speculative tiles carry the frames' smooth loss and the
            # finalize launch forms the weighted sum (scsfm_pairs_fwd_step)
            photo, geom, _, ws, smooth, sws, out = capi.photo_geometry_fwd(
                lib, flags, tgt_img, K, ref_imgs, tgt_depths, ref_depths, poses, poses_inv, hint=hint,
                check_window=_config.check_window(), smooth=True, keep_edges=True, step=(w_photo, w_smooth, w_geom))
        else:
            photo, geom, _, ws = capi.photo_geometry_fwd(lib, flags, tgt_img, K, ref_imgs, tgt_depths, ref_depths, poses,
                                                         poses_inv, group=group, hint=hint,
                                                         check_window=hint is not None and _config.check_window())
            frames = [tgt_depths[0]] + [r[0] for r in ref_depths]
            imgs = [tgt_img] + list(ref_imgs)
            # (photo, geom) are elements 0 and 1 of one contiguous row -- the library's totals, or the exact mode's sums;
            # the smooth forward's finalize launch also forms the weighted sum (no scsfm_step_total launch)
            assert geom.data_ptr() == photo.data_ptr() + photo.element_size()
            smooth, sws, out = capi.smooth_multi_fwd(lib, frames, imgs, keep_edges=any(ctx.needs_input_grad),
                                                     step=(photo, w_photo, w_smooth, w_geom))
        ctx.cfg = (flags, n_ref, n_scales, w_photo, w_smooth, w_geom)
        ctx.save_for_backward(tgt_img, K, *rest, ws, sws)
        loss, photo_o, smooth_o, geom_o = out[0], out[1], out[2], out[3]
        ctx.mark_non_differentiable(photo_o, smooth_o, geom_o)
        return loss, photo_o, smooth_o, geom_o

    @staticmethod
    def backward(ctx, g_loss, *_unused):
        from . import config as _config
        lib = _lib.get()
        flags, n_ref, n_scales, w_photo, w_smooth, w_geom = ctx.cfg
        saved = ctx.saved_tensors
        tgt_img, K = saved[0], saved[1]
        ref_imgs, tgt_depths, ref_depths, poses, poses_inv, n_in = PhotoGeometryLoss._split(saved[2:], n_ref, n_scales)
        ws, sws = saved[2 + n_in], saved[3 + n_in]
        gw = capi.step_weights(lib, _scalar(g_loss, tgt_img), w_photo, w_smooth, w_geom)
        need_imgs = [ctx.needs_input_grad[6]] + list(ctx.needs_input_grad[8:8 + n_ref])
        need_K = ctx.needs_input_grad[7]
        # the smooth term's depth gradients ride along in the pass that stores the pair terms' (scsfm_pairs_bwd_smooth)
        res = capi.photo_geometry_bwd(lib, flags, tgt_img, K, ref_imgs, tgt_depths, ref_depths, poses, poses_inv, ws,
                                      gw[0:1], gw[1:2], need_imgs=need_imgs if any(need_imgs) else None, need_K=need_K,
                                      smooth=(sws, gw[2:3]), check_window=_config.check_window())
        g_td, g_rd, g_poses, g_poses_inv = res[:4]
        g_imgs, g_K = res[4:] if len(res) > 4 else ([None] * (1 + n_ref), None)
        frames = [tgt_depths[0]] + [r[0] for r in ref_depths]
        imgs = [tgt_img] + list(ref_imgs)
        if any(need_imgs):  # (the data inputs: the smooth term reaches the images through its edge weights)
            capi.smooth_multi_bwd_images(lib, frames, imgs, sws, gw[2:3], need_imgs, into=g_imgs)
        return (None,) * 6 + (g_imgs[0], g_K, *g_imgs[1:], *g_td, *[g for r in g_rd for g in r], *g_poses, *g_poses_inv)


# ------------------------------------------------------------------------------------------------
# inverse_warp2 as maps, pose_vec2mat
# ------------------------------------------------------------------------------------------------
class InverseWarp2(torch.autograd.Function):
    """inverse_warp2 (inverse_warp.py:230-269) -> projected_img, valid_mask, projected_depth,
    computed_depth."""

    @staticmethod
    def forward(ctx, flags, img, depth, ref_depth, pose, K):
        lib = _lib.get()
        args = [_c(t) for t in (img, depth, ref_depth, pose, K)]
        _need_cuda(*args)
        o_img, o_valid, o_pd, o_cd = capi.warp_fwd(lib, *args, flags)
        ctx.flags = flags
        ctx.save_for_backward(*args)
        ctx.mark_non_differentiable(o_valid)
        ctx.set_materialize_grads(False)  # unused maps arrive as None and are skipped by the kernel
        return o_img, o_valid, o_pd, o_cd

    @staticmethod
    def backward(ctx, g_img, g_valid, g_pd, g_cd):
        lib = _lib.get()
        img, depth, ref_depth, pose, K = ctx.saved_tensors
        cc = lambda t: None if t is None else t.contiguous()
        res = capi.warp_bwd(lib, img, depth, ref_depth, pose, K, ctx.flags, cc(g_img), cc(g_pd), cc(g_cd),
                            need_img=ctx.needs_input_grad[1], need_K=ctx.needs_input_grad[5])
        g_depth, g_ref, g_pose = res[:3]
        g_src, g_K = res[3:] if len(res) > 3 else (None, None)
        return None, g_src, g_depth, g_ref, g_pose, g_K


class Pixel2Cam(torch.autograd.Function):
    """pixel2cam (inverse_warp.py:29-44): depth [B,H,W], intrinsics_inv [B,3,3] -> [B,3,H,W]."""

    @staticmethod
    def forward(ctx, depth, Kinv):
        depth, Kinv = _c(depth), _c(Kinv)
        _need_cuda(depth, Kinv)
        ctx.save_for_backward(Kinv, depth)
        return capi.pixel2cam_fwd(_lib.get(), depth, Kinv)

    @staticmethod
    def backward(ctx, g_cam):
        Kinv, depth = ctx.saved_tensors
        g_cam = g_cam.contiguous()
        g_depth = capi.pixel2cam_bwd(_lib.get(), Kinv, g_cam) if ctx.needs_input_grad[0] else None
        g_kinv = capi.pixel2cam_bwd_intrinsics(_lib.get(), depth, g_cam) if ctx.needs_input_grad[1] else None
        return g_depth, g_kinv


class Cam2Pixel(torch.autograd.Function):
    """cam2pixel (inverse_warp.py:47-74) and cam2pixel2 (:194-227; ``overwrite`` = zeros padding, ``want_z``):
    forward(flags, want_z, cam, rot | None, tr | None) -> grid [B,H,W,2] (, z [B,1,H,W])."""

    @staticmethod
    def forward(ctx, flags, want_z, cam, rot, tr):
        cam = _c(cam)
        rot = None if rot is None else _c(rot)
        tr = None if tr is None else _c(tr)
        _need_cuda(cam, rot, tr)
        grid, z = capi.cam2pixel_fwd(_lib.get(), cam, rot, tr, flags, want_z)
        ctx.flags, ctx.want_z = flags, want_z
        ctx.has = (rot is not None, tr is not None)
        ctx.save_for_backward(cam, *(t for t in (rot, tr) if t is not None))
        ctx.set_materialize_grads(False)
        return (grid, z) if want_z else grid

    @staticmethod
    def backward(ctx, g_grid, g_z=None):
        saved = list(ctx.saved_tensors)
        cam = saved.pop(0)
        rot = saved.pop(0) if ctx.has[0] else None
        tr = saved.pop(0) if ctx.has[1] else None
        if g_grid is None:
            g_grid = torch.zeros(cam.shape[0], cam.shape[2], cam.shape[3], 2, dtype=cam.dtype, device=cam.device)
        cc = lambda t: None if t is None else t.contiguous()
        g_cam, g_rot, g_tr = capi.cam2pixel_bwd(_lib.get(), cam, rot, tr, ctx.flags, cc(g_grid), cc(g_z))
        return None, None, g_cam, g_rot if rot is not None else None, g_tr if tr is not None else None


class PoseVec2Mat(torch.autograd.Function):
    """pose_vec2mat (inverse_warp.py:139-154): [B,6] -> [B,3,4]."""

    @staticmethod
    def forward(ctx, vec, mode):
        lib = _lib.get()
        vec = _c(vec)
        _need_cuda(vec)
        ctx.mode = mode
        ctx.save_for_backward(vec)
        return capi.pose_fwd(lib, vec, mode)

    @staticmethod
    def backward(ctx, g_mat):
        (vec,) = ctx.saved_tensors
        return capi.pose_bwd(_lib.get(), vec, ctx.mode, g_mat.contiguous()), None


# ------------------------------------------------------------------------------------------------
# Stand-alone public helpers of the loss module
# ------------------------------------------------------------------------------------------------
class SsimMap(torch.autograd.Function):
    """SSIM.forward (loss_functions.py:28-42) with gradients to both images."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = _c(x), _c(y)
        _need_cuda(x, y)
        ctx.save_for_backward(x, y)
        return capi.ssim_fwd(_lib.get(), x, y)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        nx, ny = ctx.needs_input_grad
        if not (nx or ny):
            return None, None
        return capi.ssim_bwd(_lib.get(), x, y, g.contiguous(), nx, ny)


class MaskedMean(torch.autograd.Function):
    """mean_on_mask (loss_functions.py:123-129), with gradients to diff and -- when it requires grad -- to the mask."""

    @staticmethod
    def forward(ctx, diff, mask):
        diff, mask = _c(diff), _c(mask)
        _need_cuda(diff, mask)
        out, ws = capi.masked_mean_fwd(_lib.get(), diff, mask)
        ctx.shape = diff.shape
        ctx.save_for_backward(mask, ws, *([diff] if ctx.needs_input_grad[1] else []))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        mask, ws = ctx.saved_tensors[:2]
        gs = _scalar(g, mask)
        g_diff = capi.masked_mean_bwd(_lib.get(), ctx.shape, mask, ws, gs) if ctx.needs_input_grad[0] else None
        g_mask = capi.masked_mean_bwd_mask(_lib.get(), ctx.saved_tensors[2], mask.shape, ws, gs) \
            if ctx.needs_input_grad[1] else None
        return g_diff, g_mask
