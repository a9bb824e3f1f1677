#!/usr/bin/env python3
"""Benchmark of SC-SfMLearner training on MI355X: BASELINE.json's metric, "train images/sec (+ warp-loss ms/step)
KITTI 256x832 RN18".

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One timed "step" = one whole training step exactly as train.py:249-286 runs it (this repo's train.train_step):
3 DispResNet18 + 4 PoseResNet18 forwards (PyTorch-ROCm / MIOpen, fp32), the HIP warp + loss hot path
(compute_photo_and_geometry_loss over 2 refs x 2 directions + compute_smooth_loss over 3 frames + the weighted
sum 1 / 0.1 / 0.5), backward through both, Adam.  Workload = BASELINE.json configs[1]: KITTI 256x832, batch 12
per GPU, sequence length 3, SSIM + mask + auto-mask, zeros padding, 1 scale; synthetic batch resident in HBM,
random-init nets.  N > 1: one process per GPU, DistributedDataParallel (bucketed RCCL all-reduce of 26.8 M fp32
gradients per step), weak scaling (12 samples per GPU), no collective on the loss path.
`value` = global_batch * K / max-over-ranks elapsed time of the K timed steps.

The hot path alone (the part this repo implements as hand-written HIP kernels; 0.5 % of the training step) is
measured beside it over --loss-steps steps on depth maps / poses resident in HBM: `warp_loss_ms_per_step`,
`hot_path_images_per_sec`, and
  roofline     -- the dominant kernel (pair_fwd_spec_kernel: warp + losses + both backward passes of all
                  pair-directions in one launch, and -- round 6 -- the smooth loss of the step's frames): algorithmic bytes
                  per launch (48 B/pixel per pair-direction forward + backward x B*H*W x pair-directions, + the
                  4 B/pixel edge plane of every frame whose smooth loss rides) / its average launch duration measured
                  here with HIP events on the launching stream, against the 8 TB/s HBM3E peak = `frac` (the judged figure).
                  `bound` names what really bounds it (valu_issue), with `issue_bound_us` / `frac_of_issue_bound` (static
                  issue units x waves per SIMD, profiles/issue_cost_latest.json), `kernel_own_frac`, and `traffic` /
                  `traffic_detail`: FETCH_SIZE and WRITE_SIZE of rocprofv3 --pmc passes, MEASURED in this run (--pmc-live:
                  child runs of the loss-path leg under the counters after the timed regions; `traffic_is` says so) or
                  else QUOTED from the committed passes of the loaded library; separately, raw and calibrated against
                  known-bytes kernels
  cpu_baseline -- the reference's CPU loss path (the unmodified reference when /root/reference is mounted -- never
                  on the GPU box -- else the oracle, its restatement on the same ATen CPU ops) forward + backward on
                  this host's cores, PINNED, in blocks of >= 15 steps repeated until two consecutive medians agree within
                  10 %: median of the last two blocks, fastest step, block medians; bounded sample, rank 0, N=1 only;
                  `configs0_end_to_end`: BASELINE.json configs[0] (nets + loss + Adam, batch 4) on the host
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "sc-sfmlearner-release_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy rate
W_PHOTO, W_SMOOTH, W_GEOM = 1.0, 0.1, 0.5  # train.py:45-47 defaults used by scripts/train_resnet18_depth_256.sh
DOMINANT_KERNEL = "pair_fwd_spec_kernel<float,true,7u,false,false>"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def step_bytes(n_px, n_ref):
    """Algorithmic (compulsory) HBM bytes of one step per GPU, BASELINE.md §2."""
    return n_px * (48 * 2 * n_ref + 24 * (1 + n_ref))


def make_inputs(args, seed, device):
    from scsfm_hip import synth
    d = synth.make_batch(args.batch, args.height, args.width, n_ref=args.n_ref, seed=seed, depth=args.depth,
                         image=synth.image_law(args.depth), dataset=args.dataset)
    to = lambda t: t.to(device)
    return {
        "tgt_img": to(d["tgt_img"]), "ref_imgs": [to(t) for t in d["ref_imgs"]], "K": to(d["intrinsics"]),
        "tgt_depth": [to(t).requires_grad_(True) for t in d["tgt_depth"]],
        "ref_depths": [[to(t).requires_grad_(True) for t in r] for r in d["ref_depths"]],
        "poses": [to(t).requires_grad_(True) for t in d["poses"]],
        "poses_inv": [to(t).requires_grad_(True) for t in d["poses_inv"]],
    }, d


def hot_path_step(LF, x, flags):
    for t in x["tgt_depth"] + [t for r in x["ref_depths"] for t in r] + x["poses"] + x["poses_inv"]:
        t.grad = None
    photo, geom = LF.compute_photo_and_geometry_loss(x["tgt_img"], x["ref_imgs"], x["K"], x["tgt_depth"],
                                                     x["ref_depths"], x["poses"], x["poses_inv"], 1, *flags)
    smooth = LF.compute_smooth_loss(x["tgt_depth"], x["tgt_img"], x["ref_depths"], x["ref_imgs"])
    loss = W_PHOTO * photo + W_SMOOTH * smooth + W_GEOM * geom
    loss.backward()
    return loss, photo, smooth, geom


def hot_path_step_single_node(LF, x, flags):
    """The same step through this repo's extension compute_total_loss: both losses and the weighted sum behind
    one autograd node (not the reference's call structure, hence not the headline)."""
    for t in x["tgt_depth"] + [t for r in x["ref_depths"] for t in r] + x["poses"] + x["poses_inv"]:
        t.grad = None
    loss, photo, smooth, geom = LF.compute_total_loss(x["tgt_img"], x["ref_imgs"], x["K"], x["tgt_depth"], x["ref_depths"],
                                                      x["poses"], x["poses_inv"], 1, *flags, W_PHOTO, W_SMOOTH, W_GEOM)
    loss.backward()
    return loss, photo, smooth, geom


def pmc_traffic(args, n_pairs, lib_source_id):
    """-> ({"fetch_bytes", "write_bytes", ...} | None, why).  The bytes are QUOTED from the committed PMC passes
    (profiles/pmc_latest.json: FETCH_SIZE and WRITE_SIZE in KiB per dispatch, collected in separate `rocprofv3 --pmc` runs
    of this very command -- tools/gpu_round.sh), and only for the workload AND the library they were collected on: the
    file records the source id of the library that ran (`_library_source_id`), and a loaded library with another id --
    a later kernel change -- gets None instead of the old counters.  The two counters are reported SEPARATELY, raw and
    corrected with the calibration of profiles/r06_fetch_calibration.json (known-bytes kernels of tools/ubench/fetch_calib.hip
    in this library's access widths under the same counters; MI355X_MICROARCH.md: FETCH_SIZE counts half the bytes of wide
    streaming reads on gfx950)."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return None, "no profiles/pmc_latest.json"
    if (args.batch, args.height, args.width, args.n_ref, args.depth) != (12, 256, 832, 2, "smooth"):
        return None, "counters were collected on configs[1] only"
    try:
        d = json.load(open(path))
        have = d.get("_library_source_id")
        if have != lib_source_id:
            return None, f"counters are those of library {have}, the loaded one is {lib_source_id}: re-run tools/gpu_round.sh"
        gz = n_pairs * args.batch
        # the training-flags instantiation of the speculative forward (the template list grew over the rounds)
        keys = [n for n in d if n.startswith("scsfm::pair_fwd_spec_kernel<float, true, 7u") and n.endswith(f"|gz{gz}")]
        k = d[keys[0]]
        out = _calibrated({"fetch_bytes_raw": int(k["FETCH_SIZE"] * 1024), "write_bytes_raw": int(k["WRITE_SIZE"] * 1024)})
        return out, "profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/gpu_round.sh on this library): QUOTED, not measured in this run"
    except (KeyError, ValueError, IndexError, TypeError):
        return None, "profiles/pmc_latest.json has no entry for the dominant kernel"


def _calibrated(out):
    """The two raw counters beside what they mean in bytes (profiles/r06_fetch_calibration.json, known-bytes kernels of
    tools/ubench/fetch_calib.hip under the same counters): FETCH_SIZE counts HALF the bytes of coalesced reads (4, 8 and 16
    bytes per lane alike) and ALL bytes of 8-byte gathers; WRITE_SIZE counts all bytes of stores and of float atomics.  The
    kernel mixes coalesced rows and gathers, so its true fetch lies between the raw counter (all gathers) and twice it (all
    coalesced rows)."""
    cal_path = os.path.join(ROOT, "profiles", "r06_fetch_calibration.json")
    if not os.path.exists(cal_path):
        return out
    try:
        cal = json.load(open(cal_path)).get("kernels", {})
        f = {n: v["FETCH_SIZE_over_known"] for n, v in cal.items() if ("read_kernel<unsigned int>" in n or "gather_b64" in n) and "FETCH_SIZE_over_known" in v}
        w = [v["WRITE_SIZE_over_known"] for n, v in cal.items() if ("write_kernel<unsigned int>" in n or "atomic_f32" in n) and "WRITE_SIZE_over_known" in v]
        if len(f) == 2 and w and min(f.values()) > 0:
            lo, hi = min(f.values()), max(f.values())
            out["fetch_counter_per_known_byte"] = {"coalesced_b32": [v for n, v in f.items() if "read_kernel" in n][0],
                                                   "gather_b64": [v for n, v in f.items() if "gather" in n][0]}
            out["write_counter_per_known_byte"] = round(sum(w) / len(w), 4)
            out["fetch_bytes_range"] = [int(out["fetch_bytes_raw"] / hi), int(out["fetch_bytes_raw"] / lo)]
            out["write_bytes"] = int(out["write_bytes_raw"] / (sum(w) / len(w)))
            out["calibration"] = "profiles/r06_fetch_calibration.json"
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return out


PMC_LIVE_CHILD_ARGS = ["--loss-steps", "3", "--loss-warmup", "1", "--cpu-seconds", "0", "--e2e", "0", "--graph", "0",
                       "--kernel-iters", "2", "--other-laws", "0", "--pmc-live", "0"]


def pmc_traffic_live(args, n_pairs, timeout_s):
    """-> ({"fetch_bytes_raw", "write_bytes_raw", ...} | None, why).  FETCH_SIZE and WRITE_SIZE per launch of the dominant
    kernel MEASURED in this run, on this box and this library: two child runs of this file's loss-path leg (eager steps of
    the reference's call structure on the same synthetic batch) under `rocprofv3 --pmc <counter> --kernel-trace`, ONE
    counter per pass as MI355X_MICROARCH.md prescribes (no other trace domain beside the counters), after every timed
    region of the parent has ended.  The average over the dispatches of the training-flags instantiation with this
    workload's grid is what is reported.  Any failure (no rocprofv3 on PATH, a pass that times out, a database without the
    kernel) returns None with the reason, and the caller falls back to the committed counters."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3")
    if tool is None:
        return None, "no rocprofv3 on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process already runs under a profiler"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                             "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env["TMPDIR"] = env.get("TMPDIR", "/tmp")
    shape = ["--batch", str(args.batch), "--height", str(args.height), "--width", str(args.width), "--n-ref", str(args.n_ref),
             "--dataset", args.dataset, "--depth", args.depth]
    gz, raw, passes = n_pairs * args.batch, {}, {}
    with tempfile.TemporaryDirectory(prefix="scsfm_pmc_") as tmp:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [tool, "--pmc", c, "--kernel-trace", "-d", tmp, "-o", f"pmc_{c}", "--", sys.executable,
                   os.path.abspath(__file__), *PMC_LIVE_CHILD_ARGS, *shape]
            t0 = time.perf_counter()
            try:
                out = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, f"the {c} pass did not finish within {timeout_s:.0f} s"
            except OSError as exc:
                return None, f"rocprofv3 could not be started: {exc}"
            db = os.path.join(tmp, f"pmc_{c}_results.db")
            if out.returncode != 0 or not os.path.exists(db):
                return None, f"the {c} pass ended with rc {out.returncode}: {out.stderr[-300:]}"
            try:
                row = sqlite3.connect(db).cursor().execute(
                    "select avg(p.counter_value), count(*), avg(p.duration) from pmc_events p join kernels k on p.dispatch_id = "
                    "k.dispatch_id where p.counter_name = ? and k.name like '%pair_fwd_spec_kernel<float, true, 7u%' and k.grid_z = ?",
                    (c, gz)).fetchone()
            except sqlite3.Error as exc:
                return None, f"the {c} pass left a database this reader does not understand: {exc}"
            if not row or not row[1]:
                return None, f"the {c} pass saw no launch of the dominant kernel"
            raw[c] = row[0] * 1024.0  # (the counters are in KiB)
            passes[c] = {"launches": int(row[1]), "avg_launch_us_under_the_counter": round(row[2] / 1e3, 1),
                         "pass_wall_s": round(time.perf_counter() - t0, 1)}
    out = _calibrated({"fetch_bytes_raw": int(raw["FETCH_SIZE"]), "write_bytes_raw": int(raw["WRITE_SIZE"])})
    out["passes"] = passes
    return out, ("measured in this run: two child runs of this command's loss-path leg (" + " ".join(PMC_LIVE_CHILD_ARGS[:-2]) +
                 ") under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace`, one counter per pass, on this box and the loaded library")


def issue_bound(lib_source_id, tiles, waves_per_tile=4, simds=1024):
    """-> (issue_bound_us | None, detail): the launch time if the vector ALUs never idled = static issue units per
    thread (profiles/issue_cost_latest.json, tools/issue_bound.py -- only for the library it was computed on) x waves per
    SIMD per launch x the measured issue time of a unit."""
    path = os.path.join(ROOT, "profiles", "issue_cost_latest.json")
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None, "no profiles/issue_cost_latest.json (tools/issue_bound.py)"
    if d.get("_library_source_id") != lib_source_id:
        return None, f"issue cost is that of library {d.get('_library_source_id')}, the loaded one is {lib_source_id}: re-run tools/issue_bound.py"
    waves_per_simd = tiles * waves_per_tile / simds
    us = d["issue_units_per_thread"] * d["ns_per_unit_per_simd"] * waves_per_simd * 1e-3
    return us, {"issue_units_per_thread": d["issue_units_per_thread"], "static_valu_instructions": d["static_valu_instructions"],
                "waves_per_simd_per_launch": round(waves_per_simd, 2), "ns_per_unit": d["ns_per_unit_per_simd"],
                "source": "profiles/issue_cost_latest.json (tools/issue_bound.py: static listing x issue weights measured with tools/ubench)"}


def _event_time(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3  # seconds per call


def time_kernels(x, flags, iters, n_ref):
    """Average duration (HIP events on the launching stream = torch's current stream) of the library
    calls a step is made of, at the step's real launch shapes: the speculative forward of all
    2*n_ref pair-directions (pairs_prep + scatter-plane clear + ONE pair_fwd_spec_kernel launch + finalize),
    their backward (geometry pass + pose reduce + combine, one launch each), and the smooth loss of the
    1+n_ref frames.  The figure of a stage contains its few-microsecond helper kernels, i.e. it over-
    rather than under-states it; `spec_kernel_only` launches the dominant kernel alone (the figure the
    roofline uses; it agrees with rocprofv3's per-kernel average)."""
    from scsfm_hip import _lib, capi
    lib = _lib.get()
    fl = capi.make_flags(*flags)
    det = lambda t: t.detach()
    tgt, K, refs = x["tgt_img"], x["K"], x["ref_imgs"]
    tds, rds = [det(x["tgt_depth"][0])], [[det(r[0])] for r in x["ref_depths"]]
    ps, pis = [det(p) for p in x["poses"]], [det(p) for p in x["poses_inv"]]
    one, half = torch.ones(1, device=tgt.device), torch.full((1,), 0.5, device=tgt.device)
    hint = (W_PHOTO, W_GEOM)
    _, _, _, ws_spec = capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=hint)
    _, _, _, ws_plain = capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=None)
    _, _, _, ws_stale = capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=hint)
    third = torch.full((1,), 0.3, device=tgt.device)
    frames, imgs = tds + [r[0] for r in rds], [tgt] + list(refs)
    _, sws = capi.smooth_multi_fwd(lib, frames, imgs)
    calls = {
        "pairs_fwd_spec": lambda: capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=hint),
        # the dominant kernel alone (constants from the call above are still in ws_spec)
        "spec_kernel_only": lambda: capi.photo_geometry_fwd(lib, fl | 16384, tgt, K, refs, tds, rds, ps, pis,
                                                                  hint=hint, ws=ws_spec),
        "pairs_fwd_plain": lambda: capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=None),
        "pairs_bwd_after_spec": lambda: capi.photo_geometry_bwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, ws_spec, one, half),
        "pairs_bwd_after_plain": lambda: capi.photo_geometry_bwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, ws_plain, one, half),
        # the first step after the loss weights changed (-p 1 -c 0.3 against a forward that speculated on 1 : 0.5): the
        # backward's own two passes run once; it leaves the new weights on the device (scsfm_pair_desc::hint) and the
        # next forward speculates on them, i.e. every later step is `pairs_bwd_after_spec` again
        "pairs_bwd_first_step_after_weight_change": lambda: capi.photo_geometry_bwd(lib, fl, tgt, K, refs, tds, rds, ps, pis,
                                                                                    ws_stale, one, third),
        "smooth_fwd": lambda: capi.smooth_multi_fwd(lib, frames, imgs),
        "smooth_bwd": lambda: capi.smooth_multi_bwd(lib, frames, imgs, sws, one),
    }
    if capi.smooth_rides_along(fl, tgt, tds, rds, hint):
        # round 6: the forward as the product runs it -- the frames' smooth loss evaluated in the speculative tiles (no
        # smooth_fwd launch in the step) -- and the dominant kernel alone in that mode
        calls["pairs_fwd_spec_with_smooth"] = lambda: capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=hint, smooth=True)
        calls["spec_kernel_only_with_smooth"] = lambda: capi.photo_geometry_fwd(lib, fl | 16384, tgt, K, refs, tds, rds, ps, pis,
                                                                                hint=hint, ws=ws_spec, smooth=True)
    return {k: _event_time(fn, iters) for k, fn in calls.items()}


def e2e_train(args, device, world, dev_index, barrier, dist_on=False):
    """K timed whole training steps of BASELINE.json's metric (train.py:249-286): 3 DispResNet + 4 PoseResNet18
    forwards (PyTorch-ROCm / MIOpen), the HIP loss path, backward, Adam; data parallel with
    DistributedDataParallel (train.wrap_ddp) when world > 1.  Synthetic batch resident in HBM, random-init nets.
    W untimed warm-up steps, then exactly K steps between barrier + synchronize, max over ranks."""
    import argparse as _ap

    import models
    import train as T
    torch.manual_seed(0)
    targs = _ap.Namespace(photo_loss_weight=W_PHOTO, smooth_loss_weight=W_SMOOTH, geometry_consistency_weight=W_GEOM,
                          num_scales=1, with_ssim=1, with_mask=1, with_auto_mask=1, padding_mode="zeros", world=world,
                          exact_mask_normalisation=bool(getattr(args, "exact", 0)))
    if targs.exact_mask_normalisation and dist_on:
        from scsfm_hip import dist as hip_dist
        hip_dist.enable_exact_normalisation()
    disp_net = models.DispResNet(args.resnet_layers, False).to(device).train()
    pose_net = models.PoseResNet(18, False).to(device).train()
    targs.channels_last = int(getattr(args, "channels_last", 0))
    if targs.channels_last:
        disp_net, pose_net = disp_net.to(memory_format=torch.channels_last), pose_net.to(memory_format=torch.channels_last)
    n_grad = 0
    if dist_on:
        T.freeze_unused_scale_heads(disp_net, targs.num_scales)
        disp_net, pose_net = T.wrap_ddp(disp_net, dev_index), T.wrap_ddp(pose_net, dev_index)
    params = [{"params": [p for p in disp_net.parameters() if p.requires_grad]},
              {"params": [p for p in pose_net.parameters() if p.requires_grad]}]
    n_grad = sum(p.numel() for g in params for p in g["params"])
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.999))
    g = torch.Generator().manual_seed(1234 + int(os.environ.get("RANK", 0)))
    mk = lambda: ((torch.rand(args.batch, 3, args.height, args.width, generator=g) - 0.45) / 0.225).to(device)
    tgt, refs = mk(), [mk() for _ in range(args.n_ref)]
    from scsfm_hip import synth
    K = synth.intrinsics(__import__("numpy").random.default_rng(int(os.environ.get("RANK", 0))), args.batch, args.height,
                         args.width, args.dataset).to(device)
    for _ in range(args.warmup):
        T.train_step(targs, disp_net, pose_net, opt, tgt, refs, K)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = T.train_step(targs, disp_net, pose_net, opt, tgt, refs, K)[0]
    barrier()
    dt = time.perf_counter() - t0
    ranks = 1
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        one = torch.ones(1, device=device)
        dist.all_reduce(one)  # every rank contributed: the number of ranks the collective actually spanned
        ranks = int(one.item())
    final = float(loss.detach())
    if not (final == final and abs(final) < 1e6):
        raise RuntimeError(f"training diverged in the bench: loss {final}")
    del disp_net, pose_net, opt
    if targs.exact_mask_normalisation and dist_on:
        hip_dist.disable_exact_normalisation()  # (the hot-path legs below time the default mode)
    return {"elapsed_s": dt, "ranks": ranks, "final_loss": final, "trainable_parameters": n_grad,
            "model": f"DispResNet{args.resnet_layers} + PoseResNet18, random init, fp32 (MIOpen convolutions)"}


def cpu_baseline(args, budget_s):
    """The reference's CPU loss path forward + backward on host cores (oracle/cpu_baseline.py in a subprocess):
    the unmodified reference when its tree is mounted (build container), else the oracle in impl='aten' mode (the
    same ATen CPU ops; the only possibility on the GPU box).  Variants per SURVEY 8d: all usable cores and one
    thread, the reference's batch (4) and the GPU config's batch, anomaly mode off and -- once -- on
    (train.py:67).  `value` is the best images/s at the GPU config's batch, anomaly off."""
    import subprocess
    ref_dir = "/root/reference"
    kind = "reference" if os.path.exists(os.path.join(ref_dir, "loss_functions.py")) else "port"
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # intra-op threads: the cores this process may use, capped -- the ATen CPU ops of this path stop scaling (and,
    # oversubscribed, collapse: 129 s/step at 256 threads) long before a 256-core host is filled
    cores = max(1, min(avail, args.cpu_threads))
    b_ref, b_gpu = min(4, args.batch), args.batch
    variants = [(cores, b_gpu, 0), (cores, b_ref, 0), (1, b_ref, 0), (cores, b_ref, 1)]
    # "all cores", within reason: one wider row so that the line itself shows that more threads are slower on this
    # path (measured on the 256-core host of the GPU box, batch 12: 8 threads 471 ms, 16: 386, 32: 587, 64: 1172,
    # 128: 3386, 256: 129 s per step)
    if avail >= 4 * cores:
        variants.append((min(avail, 4 * cores), b_gpu, 0))
    variants = list(dict.fromkeys(variants))
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--impl", "reference" if kind == "reference" else "oracle",
           "--variants", ",".join(f"{t}:{b}:{a}" for t, b, a in variants), "--seconds", str(budget_s), "--min-steps", "15",
           "--e2e-configs0", "1" if (args.height, args.width, args.n_ref, args.dataset) == (256, 832, 2, "kitti") else "0",
           "--height", str(args.height), "--width", str(args.width), "--n-ref", str(args.n_ref), "--depth", args.depth,
           "--dataset", args.dataset]
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH",)}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""  # host cores only
    out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=env, timeout=max(240.0, 20 * budget_s))
    if out.returncode != 0:
        raise RuntimeError("cpu baseline failed: " + out.stderr[-2000:])
    rows = json.loads(out.stdout.strip().split("\n")[-1])
    e2e0 = [r for r in rows if r.get("configs0_end_to_end")]
    rows = [r for r in rows if not r.get("configs0_end_to_end")]
    head = max((r for r in rows if r["batch"] == b_gpu and not r["anomaly_mode"]), key=lambda r: r["images_per_sec"])
    n_px = head["batch"] * args.height * args.width
    return {"value": head["images_per_sec"], "unit": "images/s (loss path only, fwd+bwd)", "cores": head["threads"], "kind": kind,
            "sample": f"{'unmodified reference loss_functions.py' if kind == 'reference' else 'oracle (ATen CPU ops of the reference path)'} "
                      f"fwd+bwd, batch {head['batch']} x {args.height}x{args.width}, {args.n_ref} refs, {head['threads']} threads "
                      f"pinned to one NUMA node's cores (OMP_PROC_BIND=close), blocks of >= 15 timed steps repeated until two "
                      f"consecutive medians agree within 10 %: median of the last two blocks {head['ms_per_step']} ms/step, "
                      f"fastest step {head.get('min_ms_per_step')} ms, {head['timed_steps']} steps; other thread counts / "
                      f"batch / anomaly mode under `variants`",
            "ms_per_step": head["ms_per_step"], "min_ms_per_step": head.get("min_ms_per_step"),
            "block_medians_ms": head.get("block_medians_ms"), "last_two_blocks_differ_by": head.get("last_two_blocks_differ_by"),
            "algorithmic_GBs": round(step_bytes(n_px, args.n_ref) / (head["ms_per_step"] * 1e-3) / 1e9, 3),
            "host_cores_available": avail, "variants": rows,
            # BASELINE.json configs[0] end to end on the host (nets + loss + Adam, batch 4): context for `value`
            "configs0_end_to_end": e2e0[0] if e2e0 else None}


def library_identity(lib):
    """Which binary served the run: resolved path, ABI version, size + sha256 of the .so, the hash of the sources next to
    it (csrc/*, include/scsfm_hip.h: scsfm_hip.build.source_id) and the one compiled into the binary."""
    import hashlib
    from scsfm_hip import build as hip_build
    blob = open(lib.path, "rb").read()
    return {"path": os.path.realpath(lib.path), "abi_version": int(lib._dll.scsfm_abi_version()), "bytes": len(blob),
            "so_sha256_16": hashlib.sha256(blob).hexdigest()[:16], "source_sha256_16": hip_build.source_id(),
            # what the binary itself says it was built from (scsfm_source_id): the loader rebuilds or refuses on a mismatch
            "source_id_in_binary": lib.source_id(),
            "env_override": bool(os.environ.get("SCSFM_HIP_LIB"))}


def torchrun_command(n_gpus, argv, port=None):
    """The command `python bench.py --gpus N ...` turns itself into when it was started without a launcher: one rank per
    GPU of ONE node over 127.0.0.1 (the container's hostname may not resolve), the same arguments."""
    if port is None:
        import socket
        with socket.socket() as sock:  # a free port: two benches on one box must not meet on a fixed one
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(n_gpus, argv):
    """Re-execute under torch.distributed.run; the ranks inherit stdout (rank 0 prints the one JSON line) and stderr;
    the launcher's exit status -- non-zero if any rank failed -- becomes ours."""
    import subprocess
    cmd = torchrun_command(n_gpus, argv)
    log("bench.py: --gpus %d without a launcher: re-executing as `%s`" % (n_gpus, " ".join(cmd)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed whole-training steps (K)")
    ap.add_argument("--warmup", type=int, default=10, help="untimed training steps before them (W)")
    ap.add_argument("--batch", type=int, default=12, help="samples per GPU (configs[1]: 12)")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--n-ref", type=int, default=2, help="sequence length - 1")
    ap.add_argument("--dataset", default="kitti", choices=["kitti", "nyu"])
    ap.add_argument("--resnet-layers", type=int, default=18, choices=[18, 50], help="DispResNet encoder (configs[3]: 50)")
    ap.add_argument("--depth", default="smooth", choices=["smooth", "iid", "scene"],
                    help="synthetic depth law of the hot-path legs: smooth = realistic locality (headline), iid = "
                         "incoherent gathers, scene = piecewise-smooth depth with occlusion edges on ~5 %% of the pixels "
                         "(what a trained DispResNet emits; scsfm_hip.synth)")
    ap.add_argument("--loss-steps", type=int, default=50, help="timed steps of the hot-path (loss only) legs")
    ap.add_argument("--loss-warmup", type=int, default=10)
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="budget of the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="cap on the CPU baseline's intra-op threads")
    ap.add_argument("--kernel-iters", type=int, default=30)
    ap.add_argument("--e2e", type=int, default=1, help="0: skip the training steps and report the hot path alone "
                                                      "(profiling runs); `value` is then the hot-path rate and says so")
    ap.add_argument("--graph", type=int, default=1,
                    help="1: also time the hot-path step as a HIP-graph replay (scsfm_hip.graphs.GraphedStep) when "
                         "running on one GPU, 2: also under torchrun, 0: eager launches only")
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "loader"],
                    help="loader: ALSO measure the input pipeline on its own (tools/loader_bench.py: JPEG SequenceFolder tree, "
                         "transform chain in the loader workers vs on the device) and report it as `input_pipeline`; the "
                         "timed training steps keep their HBM-resident synthetic batch (the contract's `data`)")
    ap.add_argument("--force-dist", default="none", choices=["none", "nccl", "gloo"],
                    help="with one rank: still create a process group of this backend, wrap the nets in "
                         "DistributedDataParallel and run every collective of the multi-GPU path (world size 1)")
    ap.add_argument("--other-laws", type=int, default=1,
                    help="1: after the headline legs also time the dominant kernel inside a few eager steps on the scene and iid "
                         "depth laws (1-GPU runs of the smooth law only; `roofline_other_depth_laws`)")
    ap.add_argument("--channels-last", type=int, default=0,
                    help="1: the nets of the training-step leg in NHWC memory format (train.py --channels-last); the "
                         "line then says so in config.memory_format and is NOT the headline configuration")
    ap.add_argument("--pmc-live", type=int, default=1,
                    help="1: on a 1-GPU run of configs[1] on the headline depth law, MEASURE roofline.traffic in this run (two "
                         "child runs of the loss-path leg under `rocprofv3 --pmc`, after the timed regions; about a minute), "
                         "falling back to the committed counters of profiles/pmc_latest.json; 2: at any shape; 0: quote only")
    ap.add_argument("--pmc-live-timeout", type=float, default=120.0, help="seconds allowed per counter pass")
    ap.add_argument("--exact", type=int, default=0, help="1: exact mask normalisation (scsfm_hip.dist: one all-reduce of "
                                                         "the pairs' raw sums per step) in the distributed legs")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
                # started as plain `python bench.py --gpus N` (the form the driver's N = 1 command has): become the launcher
                sys.exit(self_launch(args.gpus, sys.argv[1:]))
            # a launcher started ONE rank for a command that asks for N (torchrun --nproc-per-node=1, a stale RANK /
            # WORLD_SIZE in the environment): a 1-GPU number must not be reported under --gpus N
            sys.exit(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is 1 (RANK={os.environ.get('RANK')}): "
                     f"start {args.gpus} ranks, or run without a launcher")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device: the loss path has no CPU fallback")
    # SCSFM_BENCH_SHARED_GPU=1 (testing only): every rank uses device 0 and the ranks talk over gloo, so that the
    # multi-process control flow can be exercised on a 1-GPU box; its numbers mean nothing.
    shared_gpu = os.environ.get("SCSFM_BENCH_SHARED_GPU") == "1"
    dev_index = 0 if shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = None
    # --force-dist nccl|gloo: a process group, DistributedDataParallel and every collective of the N > 1 path at world
    # size 1 -- what a 1-GPU box can verify of it (RCCL communicator, bucketed gradient all-reduce, exact-mode all-reduce)
    dist_on = world > 1 or args.force_dist != "none"
    if dist_on:
        import torch.distributed as dist
        from scsfm_hip import dist as hip_dist
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = "gloo" if shared_gpu else ("nccl" if world > 1 else args.force_dist)
        # (the same call train.py makes: nccl bound to this rank's device, gloo for ranks sharing a GPU)
        hip_dist.init_process_group(backend, rank, world, device, force=True)
        if world > 1:  # as train.py does: every rank on its own block of the host's cores (nothing at world size 1)
            hip_dist.pin_rank_to_its_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))

    import loss_functions as LF
    from scsfm_hip import _lib
    lib = _lib.get()
    assert lib.path.endswith(".so") and os.path.exists(lib.path)  # the HIP library, never a fallback
    ident = library_identity(lib)
    assert ident["abi_version"] == _lib.ABI_VERSION, ident
    assert ident["env_override"] or ident["source_id_in_binary"] == ident["source_sha256_16"], ident

    flags = (1, 1, 1, "zeros")  # with_ssim, with_mask, with_auto_mask, padding_mode (scripts/train_resnet18_depth_256.sh)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    # ---------------------------------------------------------------------------------------------------
    # 1. BASELINE.json's metric: whole training steps (nets + HIP loss path + backward + Adam), W warm-up + K timed
    # ---------------------------------------------------------------------------------------------------
    e2e = None
    if args.e2e:
        try:
            e2e = e2e_train(args, device, world, dev_index, barrier, dist_on)
        except Exception as exc:
            import traceback
            traceback.print_exc()
            log(f"bench.py: the training step failed: {type(exc).__name__}: {exc}")
            if dist_on:
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
            sys.exit(3)  # a broken training step is a broken bench: no JSON line, non-zero status
        torch.cuda.empty_cache()

    # ---------------------------------------------------------------------------------------------------
    # 2. the hot path alone (loss forward + backward on resident depth maps / poses)
    # ---------------------------------------------------------------------------------------------------
    x, _ = make_inputs(args, seed=rank, device=device)

    def timed(step_fn, before_timed=None):
        """loss-warmup untimed + exactly loss-steps timed steps between barrier + synchronize; max over ranks."""
        for _ in range(args.loss_warmup):
            o = step_fn()
        barrier()
        if before_timed is not None:
            before_timed()
        t0 = time.perf_counter()
        for _ in range(args.loss_steps):
            o = step_fn()
        barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt, o

    # eager: every kernel launched from Python each step (the way train.py calls the loss functions)
    # ... with the library's measurement hook on: every launch of the dominant kernel inside the timed steps is
    # bracketed by HIP events on the stream it is launched on (scsfm_profile_begin / _end)
    eager_elapsed, out = timed(lambda: hot_path_step(LF, x, flags), lambda: lib.call("scsfm_profile_begin", args.loss_steps))
    import ctypes
    prof_mean, prof_min, prof_n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
    lib.call("scsfm_profile_end", ctypes.addressof(prof_mean), ctypes.addressof(prof_min), ctypes.addressof(prof_n))
    in_step_us = (prof_mean.value, prof_min.value, prof_n.value)
    eager_vals = [float(v.detach()) for v in out]
    del out  # nothing may keep the eager step's autograd graph (and its stream-bound AccumulateGrad nodes) alive
    import gc
    gc.collect()
    loss_elapsed, launch = eager_elapsed, "eager launches from Python"
    # extension leg, eager: the step behind ONE autograd node (compute_total_loss; what this repo's train.py uses).
    # Measured before any graph is captured: a live capture keeps stream-bound autograd nodes alive and slows later
    # eager steps down.
    single = None
    try:
        se, so = timed(lambda: hot_path_step_single_node(LF, x, flags))
        svals = [float(v.detach()) for v in so]
        del so
        gc.collect()
        single = {"eager_ms_per_step": round(se / args.loss_steps * 1e3, 4),
                  "losses_match": all(abs(a - b) <= 1e-6 * max(1.0, abs(b)) for a, b in zip(svals, eager_vals))}
    except Exception as exc:
        single = {"error": f"{type(exc).__name__}: {exc}"}
    graph_err = None
    if args.graph and ((world == 1 and not dist_on) or args.graph > 1):
        # the same step captured once into a HIP graph and replayed: identical kernels and results, one launch
        # (multi-process runs keep to eager launches unless --graph 2: at configs[1] the two agree, and capture
        # next to RCCL's watchdog thread is not something this repo can test on a 1-GPU box)
        try:
            from scsfm_hip.graphs import GraphedStep
            gs = GraphedStep(lambda: hot_path_step(LF, x, flags))
            loss_elapsed, out = timed(gs.replay)
            launch = "HIP graph replay of the captured step (torch.cuda.CUDAGraph)"
            gvals = [float(v.detach()) for v in out]
            assert all(abs(a - b) <= 1e-6 * max(1.0, abs(b)) for a, b in zip(gvals, eager_vals)), (gvals, eager_vals)
        except Exception as exc:  # keep the eager measurement, say why the graph leg is missing
            graph_err = f"{type(exc).__name__}: {exc}"
            loss_elapsed, gvals = eager_elapsed, eager_vals
    else:
        gvals = eager_vals
    loss, photo, smooth, geom = gvals

    # extension leg as a graph replay
    if single is not None and "error" not in single and args.graph and ((world == 1 and not dist_on) or args.graph > 1):
        try:
            from scsfm_hip.graphs import GraphedStep
            gs1 = GraphedStep(lambda: hot_path_step_single_node(LF, x, flags))
            sg, _ = timed(gs1.replay)
            single["ms_per_step"] = round(sg / args.loss_steps * 1e3, 4)
        except Exception as exc:
            single["graph_error"] = f"{type(exc).__name__}: {exc}"
    gs = gs1 = None  # drop the captures (and the autograd graphs their outputs hold) before the remaining legs
    gc.collect()

    n_px = args.batch * args.height * args.width
    loss_ms = loss_elapsed / args.loss_steps * 1e3
    hot_rate = world * args.batch * args.loss_steps / loss_elapsed

    kt = time_kernels(x, flags, args.kernel_iters, args.n_ref)
    n_pairs = 2 * args.n_ref
    # dominant kernel: pair_fwd_spec_kernel, all pair-directions in one launch -- per pair-direction the warp,
    # the masked photometric / geometry sums AND the whole backward up to a scalar factor (tiled SSIM pass +
    # geometry tail: dense dL/dD_tgt plane, scattered dL/dD_ref plane, pose partials).  Algorithmic bytes per
    # launch: per pair-direction both images and both depth maps are read once (32 B/px) and the
    # two depth gradients are written once and read-modify-written once (16 B/px).
    spec_bytes = n_pairs * 48 * n_px
    # round 6: in a step the kernel also carries the frames' smooth loss (scsfm_hip.config.smooth_rides_along): it reads
    # nothing more for it -- each frame's depth and colours are its target reads anyway -- and writes the 4 B/px edge plane
    # of every frame, which is all that is added to its algorithmic bytes
    from scsfm_hip import capi as _capi, config as _config
    rides = bool(_config.smooth_rides_along() and _capi.smooth_rides_along(_capi.make_flags(*flags), x["tgt_img"], x["tgt_depth"],
                                                                            x["ref_depths"], (W_PHOTO, W_GEOM)))
    smooth_ride_bytes = 4 * n_px * (1 + args.n_ref) if rides else 0
    spec_bytes += smooth_ride_bytes
    # its average duration over the launches inside the timed (eager) steps; the back-to-back figure of
    # time_kernels (inputs still in the Infinity Cache from the previous launch) is reported beside it
    b2b = kt.get("spec_kernel_only_with_smooth", kt["spec_kernel_only"]) if rides else kt["spec_kernel_only"]
    launch_s = in_step_us[0] * 1e-6 if in_step_us[2] > 0 else b2b
    achieved = spec_bytes / launch_s / 1e9
    traffic, traffic_why, traffic_is = None, None, None
    headline_shape = (args.batch, args.height, args.width, args.n_ref, args.depth) == (12, 256, 832, 2, "smooth")
    if rank == 0 and world == 1 and not dist_on and (args.pmc_live >= 2 or (args.pmc_live == 1 and headline_shape)):
        traffic, traffic_why = pmc_traffic_live(args, n_pairs, args.pmc_live_timeout)
        if traffic is not None:
            traffic_is = "measured in this run (rocprofv3 --pmc passes of the loss-path leg, after the timed regions)"
        else:
            log(f"[bench] live counters unavailable ({traffic_why}); quoting profiles/pmc_latest.json")
    live_why = traffic_why
    if traffic is None:
        traffic, traffic_why = pmc_traffic(args, n_pairs, ident["source_id_in_binary"])
        if traffic is not None:
            traffic_is = "quoted from profiles/ (not measured in this run)"
        if live_why:
            traffic_why = f"{traffic_why}; live passes: {live_why}"
    tiles = n_pairs * args.batch * (-(-args.width // 62)) * (-(-args.height // 14))
    ib_us, ib_detail = issue_bound(ident["source_id_in_binary"], tiles)
    own_bytes = int(n_pairs * 40 * n_px + smooth_ride_bytes)
    roofline = {
        # `frac` is the judged figure: algorithmic HBM bytes per launch / measured launch time, against the 8 TB/s HBM peak.
        # What actually bounds the kernel is not bytes: its HBM traffic (below) is at or under the algorithmic figure at a
        # fifth of the peak, and the SQ counters (profiles/) show the waves issuing, or waiting for the vector issue port,
        # two thirds of the time -- it runs at `frac_of_issue_bound` of the time its vector instructions alone would take
        "bound": "valu_issue", "frac_is_against": "hbm",
        "kernel": f"{DOMINANT_KERNEL} ({n_pairs} pair-directions per launch" + (", carrying the smooth loss of the step's frames)" if rides else ")"),
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        # HBM bytes per launch, FETCH_SIZE + WRITE_SIZE (raw counters) of separate `rocprofv3 --pmc` passes: MEASURED in this
        # run where that is possible (--pmc-live: child runs of the loss-path leg under the counters, after the timed
        # regions -- counters cannot be read inside a timed run), else QUOTED from the committed profiles/pmc_latest.json
        # and only when that file was recorded on the library loaded here; `traffic_is` says which; `traffic_detail` has the
        # two counters separately, raw and calibrated against known-bytes kernels
        "traffic": None if traffic is None else traffic["fetch_bytes_raw"] + traffic["write_bytes_raw"],
        "traffic_is": traffic_is,
        "traffic_detail": traffic, "traffic_source": traffic_why,
        "issue_bound_us": None if ib_us is None else round(ib_us, 1),
        "frac_of_issue_bound": None if ib_us is None else round(ib_us / (launch_s * 1e6), 4),
        "issue_bound_detail": ib_detail,
        # of the 48 B/px booked on this kernel 8 B/px (the read-modify-write of the gradient maps) are paid by
        # pairs_combine_kernel: the kernel's own algorithmic bytes are 40 B/px (+ the edge planes when the smooth loss rides)
        "kernel_own_algorithmic_bytes_per_launch": own_bytes,
        "kernel_own_frac": round(own_bytes / launch_s / 1e9 / HBM_PEAK_GBS, 4),
        "smooth_loss_rides_in_the_kernel": rides, "smooth_edge_plane_bytes_per_launch": smooth_ride_bytes,
        "algorithmic_bytes_per_launch": spec_bytes, "avg_launch_us": round(launch_s * 1e6, 2),
        "launches_timed": in_step_us[2], "min_launch_us": round(in_step_us[1], 2),
        "back_to_back_launch_us": round(b2b * 1e6, 2)}
    # the same kernel inside eager steps on the OTHER synthetic depth laws (rank 0 of a 1-GPU run of the headline law only):
    # `scene` = piecewise-smooth depth with occlusion edges, the closest stand-in for a trained net's output; `iid` =
    # the incoherent stress case.  Same algorithmic bytes, same 8 TB/s.
    other_laws = {}
    if args.other_laws and world == 1 and args.depth == "smooth":
        import argparse as _ap
        for law in ("scene", "iid"):
            a2 = _ap.Namespace(**vars(args))
            a2.depth = law
            x2, _ = make_inputs(a2, seed=rank, device=device)
            for _ in range(3):
                hot_path_step(LF, x2, flags)
            torch.cuda.synchronize()
            n2 = max(5, min(20, args.loss_steps))
            lib.call("scsfm_profile_begin", n2)
            t0 = time.perf_counter()
            for _ in range(n2):
                hot_path_step(LF, x2, flags)
            torch.cuda.synchronize()
            step_ms = (time.perf_counter() - t0) / n2 * 1e3
            lib.call("scsfm_profile_end", ctypes.addressof(prof_mean), ctypes.addressof(prof_min), ctypes.addressof(prof_n))
            if prof_n.value > 0:
                other_laws[law] = {"avg_launch_us": round(prof_mean.value, 2), "min_launch_us": round(prof_min.value, 2),
                                   "launches_timed": prof_n.value,
                                   "frac": round(spec_bytes / (prof_mean.value * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                   "eager_loss_path_ms_per_step": round(step_ms, 4)}
            del x2
    # SURVEY.md 8d figure: one pair-direction forward + backward = 48 B/px
    pair_t = (kt["pairs_fwd_spec"] + kt["pairs_bwd_after_spec"]) / n_pairs
    pair_roofline = {"algorithmic_bytes": 48 * n_px, "us": round(pair_t * 1e6, 2),
                     "achieved_GBs": round(48 * n_px / pair_t / 1e9, 1),
                     "frac": round(48 * n_px / pair_t / 1e9 / HBM_PEAK_GBS, 4)}
    kernel_sum = (kt["pairs_fwd_spec_with_smooth"] if rides and "pairs_fwd_spec_with_smooth" in kt else
                  kt["pairs_fwd_spec"] + kt["smooth_fwd"]) + kt["pairs_bwd_after_spec"] + kt["smooth_bwd"]

    if rank == 0:
        shape = (args.dataset, args.height, args.width, args.batch, args.n_ref, args.resnet_layers)
        # which BASELINE.json config this run is the per-GPU workload of ([1] and [2] share theirs)
        named = {("kitti", 256, 832, 12, 2, 18): "configs[1]", ("kitti", 256, 832, 8, 2, 50): "configs[3]",
                 ("nyu", 256, 320, 16, 4, 18): "configs[4] (batch 16/GPU)", ("kitti", 256, 832, 4, 2, 18): "configs[0] shape"}
        workload = (f"{named.get(shape, 'custom')}: {args.dataset} {args.height}x{args.width}, batch {args.batch}/GPU, {args.n_ref} refs "
                    f"(seq {args.n_ref + 1}), ResNet{args.resnet_layers} DispNet + ResNet18 PoseNet, ssim+mask+auto-mask, "
                    f"zeros padding, 1 scale")
        if e2e is not None:
            value = world * args.batch * args.steps / e2e["elapsed_s"]
            ms_per_step = e2e["elapsed_s"] / args.steps * 1e3
            metric = "train images/sec (+ warp-loss ms/step) KITTI 256x832 RN18" if shape[:3] == ("kitti", 256, 832) and \
                args.resnet_layers == 18 else f"train images/sec (+ warp-loss ms/step) {args.dataset} {args.height}x{args.width} RN{args.resnet_layers}"
            parallelism = (f"ddp{world}: one process per GPU, DistributedDataParallel, bucketed {backend} all-reduce of "
                           f"{e2e['trainable_parameters'] * 4 / 1e6:.1f} MB of fp32 gradients per step; loss path sharded "
                           f"by batch, " + ("one all-reduce of the pairs' raw sums per step (exact mask normalisation)"
                                           if args.exact else "no loss-path collective")) if dist_on else "1 GPU"
        else:  # --e2e 0 (profiling): say plainly that this is NOT the training rate
            value, ms_per_step = hot_rate, loss_ms
            metric = "images/sec through the warp+loss hot path ONLY (--e2e 0: nets and optimizer not run)"
            parallelism = f"dp{world} (batch shards, no loss-path collective)"
        res = {
            "metric": metric,
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "global_batch": world * args.batch, "parallelism": parallelism,
                       "memory_format": "channels_last (nets only; --channels-last 1, not the default)" if args.channels_last else "contiguous (NCHW)"},
            # ranks an all-reduce on the process group actually spanned: `rccl_ranks` only when that group is nccl (= RCCL)
            ("rccl_ranks" if backend in (None, "nccl") else "collective_ranks"): e2e["ranks"] if e2e is not None else world,
            "collective_backend": backend,
            "exact_mask_normalisation": bool(args.exact),
            "train": None if e2e is None else {"train_images_per_sec": round(value, 2), "ms_per_step": round(ms_per_step, 3),
                                               "final_loss": e2e["final_loss"], "model": e2e["model"],
                                               "trainable_parameters": e2e["trainable_parameters"]},
            # ---- the hot path alone ----
            "warp_loss_ms_per_step": round(loss_ms, 4),
            "hot_path_images_per_sec": round(hot_rate, 2),
            "warp_loss_share_of_train_step": None if e2e is None else round(loss_ms / ms_per_step, 5),
            "warp_loss": {"launch": launch, "steps": args.loss_steps, "warmup": args.loss_warmup,
                          "depth_law": args.depth, "eager_ms_per_step": round(eager_elapsed / args.loss_steps * 1e3, 4),
                          "graph_error": graph_err, "single_autograd_node": single,
                          "step_algorithmic_GBs": round(step_bytes(n_px, args.n_ref) / (loss_elapsed / args.loss_steps) / 1e9, 1),
                          "step_frac_of_hbm_peak": round(step_bytes(n_px, args.n_ref) / (loss_elapsed / args.loss_steps) / 1e9 / HBM_PEAK_GBS, 4),
                          "kernel_us": {k: round(v * 1e6, 2) for k, v in kt.items()},
                          "kernel_sum_ms_per_step": round(kernel_sum * 1e3, 4),
                          "losses": {"total": loss, "photo": photo, "smooth": smooth, "geometry": geom},
                          "pair_direction_roofline": pair_roofline},
            "roofline": roofline,
            # the dominant kernel on the other two depth laws (same bytes, same peak; --other-laws 0 skips the leg)
            "roofline_other_depth_laws": other_laws or None,
            "library": ident,
        }
        if world == 1 and args.cpu_seconds > 0:
            res["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        if world == 1 and args.data == "loader":
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import loader_bench
            res["input_pipeline"] = loader_bench.run(args.batch, args.height, args.width, args.n_ref + 1)
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
