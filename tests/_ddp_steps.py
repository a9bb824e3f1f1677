"""Shared by tests/test_ddp_train_step.py (CPU: gloo + hostsim kernels) and tests/test_gpu_ddp.py (two ranks on
one MI355X): train.train_step under DistributedDataParallel at world 2 against a single-process emulation of
the same data-parallel step.

The emulation runs each rank's shard through the SAME nets one after the other (BatchNorm in train mode
normalises with per-replica batch statistics, as under DDP and under the reference's nn.DataParallel,
train.py:168-169), then
  * default mode: mean over the shards of  w1*l1_s + w2*l2_s + w3*l3_s  (per-shard masked means), or
  * exact mode:   w1*l1 + w3*l3 evaluated on the CONCATENATED depth maps / poses (global sums, global 10000
                  gate: the reference's whole-batch semantics, loss_functions.py:123-129) + w2 * mean_s l2_s.
With plain SGD the parameter update is -lr * gradient, so comparing parameters after a step compares the
(averaged) gradients themselves; the smooth weight w2 is raised so that a wrongly scaled smooth term shows.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sc-sfmlearner-release_amd")
W1, W2, W3 = 1.0, 1.0, 0.5
LR = 1e-2
WATCH = ("decoder.decoder.13.conv.weight", "decoder.decoder.0.conv.conv.weight", "encoder.encoder.conv1.weight",
         "encoder.encoder.layer3.0.conv1.weight")
WATCH_POSE = ("decoder.net.3.weight", "encoder.encoder.conv1.weight")


def _paths():
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def make_args(exact, world):
    return argparse.Namespace(photo_loss_weight=W1, smooth_loss_weight=W2, geometry_consistency_weight=W3, num_scales=1,
                              with_ssim=1, with_mask=1, with_auto_mask=1, padding_mode="zeros", world=world,
                              exact_mask_normalisation=exact, single_loss_node=1)


def make_data(B, H, W, seed, device):
    """Image-like frames: a smooth random field shifted by a few pixels between the frames (so that the auto mask
    keeps most pixels), KITTI-like intrinsics."""
    import numpy as np
    from scsfm_hip import synth
    g = torch.Generator().manual_seed(seed)
    coarse = torch.rand(B, 3, H // 8 + 2, W // 8 + 2, generator=g)
    base = torch.nn.functional.interpolate(coarse, size=(H + 8, W + 8), mode="bilinear", align_corners=False)
    base = base + 0.05 * torch.rand(B, 3, H + 8, W + 8, generator=g)
    frames = [((base[:, :, 4 + dy:4 + dy + H, 4 + dx:4 + dx + W] - 0.45) / 0.225).contiguous().to(device)
              for dy, dx in ((0, 0), (1, 3), (-1, -2))]
    K = synth.intrinsics(np.random.default_rng(seed), B, H, W, "kitti").to(device)
    return frames[0], frames[1:], K


def make_nets(device):
    import models
    torch.manual_seed(7)
    disp = models.DispResNet(18, False).to(device).train()
    pose = models.PoseResNet(18, False).to(device).train()
    return disp, pose


def make_opt(disp, pose):
    return torch.optim.SGD([{"params": [p for p in disp.parameters() if p.requires_grad]},
                            {"params": [p for p in pose.parameters() if p.requires_grad]}], lr=LR)


def snapshot(disp, pose):
    d, p = dict(disp.named_parameters()), dict(pose.named_parameters())
    # numpy: pickled by value through the result queue (torch tensors would travel as shared-memory handles that
    # die with the worker)
    return {**{"disp." + k: d[k].detach().float().cpu().numpy().copy() for k in WATCH},
            **{"pose." + k: p[k].detach().float().cpu().numpy().copy() for k in WATCH_POSE}}


def emulate(exact, world, steps, B, H, W, device):
    """Single process: the data-parallel step of `world` ranks, each with its own B-sample shard."""
    import loss_functions as LF
    import train as T
    disp, pose = make_nets(device)
    opt = make_opt(disp, pose)
    shards = [make_data(B, H, W, 100 + r, device) for r in range(world)]
    losses, snaps = [], [snapshot(disp, pose)]
    for _ in range(steps):
        outs = []
        for tgt, refs, K in shards:
            td, rd = T.compute_depth(disp, tgt, refs)
            ps, pi = T.compute_pose_with_inv(pose, tgt, refs)
            outs.append((tgt, refs, K, td, rd, ps, pi))
        flags = (1, 1, 1, 1, "zeros")
        l2s = [LF.compute_smooth_loss(td, tgt, rd, refs) for tgt, refs, K, td, rd, ps, pi in outs]
        if exact:
            cat = lambda xs: torch.cat(xs, 0)
            tgt = cat([o[0] for o in outs]); K = cat([o[2] for o in outs])
            refs = [cat([o[1][i] for o in outs]) for i in range(2)]
            td = [cat([o[3][0] for o in outs])]
            rd = [[cat([o[4][i][0] for o in outs])] for i in range(2)]
            ps = [cat([o[5][i] for o in outs]) for i in range(2)]
            pi = [cat([o[6][i] for o in outs]) for i in range(2)]
            l1, l3 = LF.compute_photo_and_geometry_loss(tgt, refs, K, td, rd, ps, pi, *flags)
            loss = W1 * l1 + W2 * sum(l2s) / world + W3 * l3
            rank0 = loss  # every rank reports the global loss (its own smooth term differs: compare the mean below)
        else:
            per = []
            for (tgt, refs, K, td, rd, ps, pi), l2 in zip(outs, l2s):
                l1, l3 = LF.compute_photo_and_geometry_loss(tgt, refs, K, td, rd, ps, pi, *flags)
                per.append(W1 * l1 + W2 * l2 + W3 * l3)
            loss = sum(per) / world
            rank0 = per[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append({"mean": float(loss.detach()), "rank0": float(rank0.detach()),
                       "photo_geom0": None})
        snaps.append(snapshot(disp, pose))
    return losses, snaps


def ddp_worker(rank, world, port, exact, steps, B, H, W, device_kind, hostsim, q):
    """One rank of the real thing: train.wrap_ddp + train.train_step."""
    _paths()
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    torch.set_num_threads(2)
    if hostsim:
        from hostsim import harness
        from scsfm_hip import _lib, ops
        lib = harness.lib()
        _lib.get = lambda: lib
        ops._need_cuda = lambda *a: None
    from scsfm_hip import config as hip_config, dist as sdist
    import train as T
    device = torch.device("cuda", 0) if device_kind == "cuda" else torch.device("cpu")
    if device_kind == "cuda":
        torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        disp, pose = make_nets(device)
        T.freeze_unused_scale_heads(disp, 1)
        disp, pose = T.wrap_ddp(disp, 0 if device_kind == "cuda" else None), T.wrap_ddp(pose, 0 if device_kind == "cuda" else None)
        opt = make_opt(disp, pose)
        args = make_args(exact, world)
        if exact:
            sdist.enable_exact_normalisation()
        hip_config.set_weight_hint(W1 * (world if exact else 1), W3 * (world if exact else 1))
        tgt, refs, K = make_data(B, H, W, 100 + rank, device)
        losses, snaps = [], [snapshot(disp.module, pose.module)]
        for _ in range(steps):
            loss, l1, l2, l3 = T.train_step(args, disp, pose, opt, tgt, refs, K)
            t = torch.stack([loss.detach().float().cpu(), l1.detach().float().cpu(), l2.detach().float().cpu(),
                             l3.detach().float().cpu()])
            mean = t.clone()
            dist.all_reduce(mean)
            losses.append({"rank": [float(v) for v in t], "mean": [float(v) / world for v in mean]})
            snaps.append(snapshot(disp.module, pose.module))
        q.put((rank, {"losses": losses, "snaps": snaps}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def compare(res_ddp, ref_losses, ref_snaps, exact, loss_tol, grad_tol, later_loss_tol=None):
    """res_ddp: {rank: {...}} from ddp_worker; ref_*: from emulate().  Snapshot 0 is the initial state; the
    update of a step is -LR * (averaged) gradient, so the parameters after the FIRST step (same parameters on both
    sides going in) are compared relative to the size of that step: that is the gradient check.  Later steps start
    from parameters that already differ in the last bits and the loss is full of discontinuous gates,
    so there only the losses are compared (`later_loss_tol`, default = loss_tol) and the ranks must stay
    bit-identical to each other."""
    later_loss_tol = loss_tol if later_loss_tol is None else later_loss_tol
    r0, r1 = res_ddp[0], res_ddp[1]
    for k, v in ref_snaps[0].items():
        assert np.array_equal(r0["snaps"][0][k], v) and np.array_equal(r1["snaps"][0][k], v), f"initial {k} differs"
    worst = 0.0
    for step, rl in enumerate(ref_losses):
        got = r0["losses"][step]
        tol = loss_tol if step == 0 else later_loss_tol
        assert got["mean"][0] == got["mean"][0]
        # total loss averaged over the ranks == the emulation's objective
        assert abs(got["mean"][0] - rl["mean"]) <= tol * max(1.0, abs(rl["mean"])), (step, got, rl)
        if exact:  # every rank holds the same global photo / geometry losses
            assert abs(r0["losses"][step]["rank"][1] - r1["losses"][step]["rank"][1]) <= 1e-6
            assert abs(r0["losses"][step]["rank"][3] - r1["losses"][step]["rank"][3]) <= 1e-6
        else:
            assert abs(got["rank"][0] - rl["rank0"]) <= tol * max(1.0, abs(rl["rank0"])), (step, got, rl)
        for k, v in ref_snaps[step + 1].items():
            a, b = r0["snaps"][step + 1][k], r1["snaps"][step + 1][k]
            assert np.array_equal(a, b), f"ranks diverged on {k} at step {step}"
            if step == 0:
                upd = float(np.abs(v - ref_snaps[0][k]).max())  # = LR * max |gradient|
                err = float(np.abs(a - v).max())
                worst = max(worst, err / max(upd, 1e-12))
                assert err <= grad_tol * upd + 1e-7, (k, err, upd)
    return worst
