#!/usr/bin/env python3
"""CPU baseline of the loss path, timed on host cores.  TEST / BENCH INFRASTRUCTURE ONLY (run by bench.py's
``cpu_baseline`` leg as a subprocess, never by the product).

    python oracle/cpu_baseline.py --impl reference|oracle --variants T:B:A[,T:B:A...] --seconds S [--ref-dir DIR]

One step = what train.py:262-268,280 does with the loss: compute_photo_and_geometry_loss (2 refs x 2 directions)
+ compute_smooth_loss (3 frames) + the weighted sum 1 / 0.1 / 0.5 + backward down to the depth maps and poses, on
the same seeded synthetic batch bench.py draws (scsfm_hip/synth.py, loaded by file path so that nothing of the
product package is imported here).

  --impl reference : the UNMODIFIED reference (``loss_functions.py`` / ``inverse_warp.py`` imported from --ref-dir,
                     default /root/reference; only possible where that tree is mounted -- never on the GPU box)
  --impl oracle    : oracle/scsfm_oracle.py in impl='aten' mode, the restatement that calls the very same ATen CPU
                     entry points (bit-identical losses, tests/test_oracle_vs_reference.py)
  variant T:B:A    : T intra-op threads, batch B, A = 1 runs under torch.autograd.set_detect_anomaly(True) as the
                     reference's train.py:67 does globally
Prints one JSON list (one object per variant) on stdout.

Round 6: every variant runs in a process of its own, pinned (OMP_PROC_BIND=close, OMP_PLACES=cores, the process confined
to as many physical cores of one NUMA node as it has threads), in blocks of >= 15 timed steps that are repeated until two
consecutive block medians agree within 10 %; `ms_per_step` is the median of the last two blocks, `min_ms_per_step` the
fastest step seen.  (The unpinned median of 7 steps moved 2.7x between the driver's runs of rounds 1-5.)
"""
import argparse
import importlib.util
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
W_PHOTO, W_SMOOTH, W_GEOM = 1.0, 0.1, 0.5


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pick_cores(n):
    """The first n physical cores of ONE NUMA node (one hardware thread per core), among those this process may use;
    fewer than n on that node -> the next nodes' cores follow.  [] where the topology cannot be read."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return []

    def cpulist(path):
        out = []
        try:
            for part in open(path).read().strip().split(","):
                if "-" in part:
                    a, b = part.split("-")
                    out.extend(range(int(a), int(b) + 1))
                elif part:
                    out.append(int(part))
        except OSError:
            pass
        return out

    nodes = []
    base = "/sys/devices/system/node"
    try:
        for name in sorted((n_ for n_ in os.listdir(base) if n_.startswith("node") and n_[4:].isdigit()), key=lambda x: int(x[4:])):
            nodes.append([c for c in cpulist(os.path.join(base, name, "cpulist")) if c in allowed])
    except OSError:
        pass
    if not nodes:
        nodes = [allowed]
    picked, seen_cores = [], set()
    for cpus in nodes:
        for c in cpus:
            sib = tuple(cpulist(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list")) or (c,)
            if sib in seen_cores:
                continue  # a second hardware thread of a core already taken
            seen_cores.add(sib)
            picked.append(c)
            if len(picked) == n:
                return picked
    return picked


def run_one(args, threads, batch, anomaly, budget):
    """One variant in THIS process (a fresh one per variant: thread pools and affinity masks are per process).  The
    threads are pinned -- OMP_PROC_BIND / OMP_PLACES were set by the parent before torch was imported here, and the
    process is confined to `threads` physical cores of one NUMA node -- because an unpinned run of this path moved
    2.7x between the driver's runs of rounds 1-5 (175 ... 468 ms per step for unchanged code)."""
    cores = pick_cores(threads)
    if cores:
        try:
            os.sched_setaffinity(0, set(cores))
        except OSError:
            cores = []
    import torch
    torch.set_num_threads(threads)
    synth = load_by_path("_scsfm_synth", os.path.join(ROOT, "sc-sfmlearner-release_amd", "scsfm_hip", "synth.py"))
    if args.impl == "reference":
        sys.path.insert(0, args.ref_dir)
        import loss_functions as RF  # the reference's own module (it imports its own inverse_warp)
        assert os.path.realpath(RF.__file__).startswith(os.path.realpath(args.ref_dir)), RF.__file__

        def step(d, td, rd, ps, pi):
            photo, geom = RF.compute_photo_and_geometry_loss(d["tgt_img"], d["ref_imgs"], d["intrinsics"], td, rd, ps, pi,
                                                             1, 1, 1, 1, "zeros")
            smooth = RF.compute_smooth_loss(td, d["tgt_img"], rd, d["ref_imgs"])
            return photo, smooth, geom
    else:
        sys.path.insert(0, ROOT)
        from oracle import scsfm_oracle as O

        def step(d, td, rd, ps, pi):
            photo, geom = O.photo_and_geometry_loss(d["tgt_img"], d["ref_imgs"], d["intrinsics"], td, rd, ps, pi, 1, 1, 1, 1,
                                                    "zeros", impl="aten")
            smooth = O.smooth_loss(td, d["tgt_img"], rd, d["ref_imgs"])
            return photo, smooth, geom

    torch.autograd.set_detect_anomaly(bool(anomaly))
    d = synth.make_batch(batch, args.height, args.width, n_ref=args.n_ref, seed=0, depth=args.depth,
                         image="smooth" if args.depth == "smooth" else "iid", dataset=args.dataset)
    if args.e2e:
        return run_e2e(args, torch, d, step, threads, batch, budget, cores)

    def one():
        lf = lambda t: t.clone().requires_grad_(True)
        td = [lf(t) for t in d["tgt_depth"]]
        rd = [[lf(t) for t in r] for r in d["ref_depths"]]
        ps, pi = [lf(p) for p in d["poses"]], [lf(p) for p in d["poses_inv"]]
        photo, smooth, geom = step(d, td, rd, ps, pi)
        loss = W_PHOTO * photo + W_SMOOTH * smooth + W_GEOM * geom
        loss.backward()
        return float(loss)

    loss = one()  # warm-up (allocator, thread pool)
    one()
    med = lambda xs: sorted(xs)[len(xs) // 2]
    # blocks of `args.min_steps` timed steps (fewer for variants that take seconds per step: the time budget bounds
    # them), repeated -- at most four blocks -- until the medians of two consecutive blocks agree within 10 %
    blocks, t_start = [], time.perf_counter()
    while len(blocks) < 4:
        times = []
        while len(times) < args.min_steps:
            t0 = time.perf_counter()
            one()
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget and len(times) >= 3:
                break
        blocks.append(times)
        if len(blocks) >= 2 and abs(med(blocks[-1]) / med(blocks[-2]) - 1.0) <= 0.10:
            break
        if time.perf_counter() - t_start > budget:
            break
    last = blocks[-1] + (blocks[-2] if len(blocks) >= 2 else [])
    every = [t for b in blocks for t in b]
    m = med(last)
    agree = abs(med(blocks[-1]) / med(blocks[-2]) - 1.0) if len(blocks) >= 2 else None
    return {"threads": threads, "batch": batch, "anomaly_mode": bool(anomaly), "ms_per_step": round(m * 1e3, 2),
            "min_ms_per_step": round(min(every) * 1e3, 2), "images_per_sec": round(batch / m, 3), "timed_steps": len(every),
            "block_medians_ms": [round(med(b) * 1e3, 2) for b in blocks],
            "last_two_blocks_differ_by": None if agree is None else round(agree, 4),
            "pinned_to_cpus": cores, "omp_proc_bind": os.environ.get("OMP_PROC_BIND"), "loss": loss}


def run_e2e(args, torch, d, loss_step, threads, batch, budget, cores):
    """BASELINE.json configs[0] END TO END on host cores (last bullet): what train.py:249-286 does per
    iteration -- DispResNet18 on the three frames, PoseResNet18 on the four ordered pairs, the loss (the unmodified reference
    or its ATen-mode restatement, as above), backward, Adam -- random-init nets (--with-pretrain 0), batch 4.  The nets are
    this repo's plain-torch models (the reference's need torchvision, which is not installed; same layers and state-dict
    keys: tests/test_models_vs_reference.py).  A few steps only: one takes seconds."""
    sys.path.insert(0, os.path.join(ROOT, "sc-sfmlearner-release_amd"))
    import models
    torch.manual_seed(0)
    disp_net, pose_net = models.DispResNet(18, False).train(), models.PoseResNet(18, False).train()
    opt = torch.optim.Adam([{"params": disp_net.parameters()}, {"params": pose_net.parameters()}], lr=1e-4, betas=(0.9, 0.999))
    tgt, refs, K = d["tgt_img"], d["ref_imgs"], d["intrinsics"]

    def one():
        tgt_depth = [1 / disp for disp in disp_net(tgt)]
        ref_depths = [[1 / disp for disp in disp_net(r)] for r in refs]
        poses = [pose_net(tgt, r) for r in refs]
        poses_inv = [pose_net(r, tgt) for r in refs]
        photo, smooth, geom = loss_step(d, tgt_depth[:1], [r[:1] for r in ref_depths], poses, poses_inv)
        loss = W_PHOTO * photo + W_SMOOTH * smooth + W_GEOM * geom
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return float(loss)

    one()  # warm-up
    times, t_start = [], time.perf_counter()
    while len(times) < 2 or (time.perf_counter() - t_start < budget and len(times) < 8):
        t0 = time.perf_counter()
        loss = one()
        times.append(time.perf_counter() - t0)
    m = sorted(times)[len(times) // 2]
    return {"configs0_end_to_end": True, "threads": threads, "batch": batch, "anomaly_mode": False,
            "ms_per_step": round(m * 1e3, 1), "min_ms_per_step": round(min(times) * 1e3, 1), "images_per_sec": round(batch / m, 3),
            "timed_steps": len(times), "pinned_to_cpus": cores, "omp_proc_bind": os.environ.get("OMP_PROC_BIND"), "loss": loss,
            "what": "DispResNet18 + PoseResNet18 forward/backward (plain torch), the loss path, Adam: one training iteration of "
                    "train.py:249-286 on host cores, random-init nets, synthetic batch"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", choices=["reference", "oracle"], required=True)
    ap.add_argument("--ref-dir", default="/root/reference")
    ap.add_argument("--variants", required=True)
    ap.add_argument("--seconds", type=float, default=15.0)
    ap.add_argument("--min-steps", type=int, default=15, help="timed steps per block of a variant")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--n-ref", type=int, default=2)
    ap.add_argument("--depth", default="smooth")
    ap.add_argument("--dataset", default="kitti")
    ap.add_argument("--e2e-configs0", type=int, default=0, help="1: also one variant that runs configs[0] END TO END (nets + "
                                                                "loss + Adam, batch 4) on the first variant's thread count")
    ap.add_argument("--e2e", type=int, default=0, help="(internal) --one runs the end-to-end step")
    ap.add_argument("--one", default=None, help="(internal) run this single variant in this process")
    ap.add_argument("--budget", type=float, default=0.0, help="(internal) seconds for --one")
    args = ap.parse_args()

    if args.one is not None:
        t, b, a = (int(x) for x in args.one.split(":"))
        print(json.dumps(run_one(args, t, b, a, args.budget)))
        return
    import subprocess
    variants = [tuple(int(x) for x in v.split(":")) for v in args.variants.split(",")]
    # the first variant is the headline (two blocks of >= 15 steps): it gets half the budget, the rest share the other half
    out = []
    for i, (threads, batch, anomaly) in enumerate(variants):
        budget = args.seconds * (0.5 if i == 0 else 0.5 / max(1, len(variants) - 1)) if len(variants) > 1 else args.seconds
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores")
        cmd = [sys.executable, os.path.abspath(__file__), "--impl", args.impl, "--ref-dir", args.ref_dir, "--variants", "-",
               "--one", f"{threads}:{batch}:{anomaly}", "--budget", str(budget), "--min-steps", str(args.min_steps),
               "--height", str(args.height), "--width", str(args.width), "--n-ref", str(args.n_ref), "--depth", args.depth,
               "--dataset", args.dataset]
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            sys.exit(r.returncode)
        out.append(json.loads(r.stdout.strip().split("\n")[-1]))
    if args.e2e_configs0:
        threads = variants[0][0]
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores")
        cmd = [sys.executable, os.path.abspath(__file__), "--impl", args.impl, "--ref-dir", args.ref_dir, "--variants", "-",
               "--one", f"{threads}:4:0", "--e2e", "1", "--budget", str(max(10.0, 0.5 * args.seconds)),
               "--height", str(args.height), "--width", str(args.width), "--n-ref", str(args.n_ref), "--depth", args.depth,
               "--dataset", args.dataset]
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        if r.returncode == 0:
            out.append(json.loads(r.stdout.strip().split("\n")[-1]))
        else:  # (reported, not fatal: the headline rows above stand)
            out.append({"configs0_end_to_end": True, "error": r.stderr[-600:]})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
