#!/usr/bin/env python3
"""Host side of a hot-path step: wall time to ENQUEUE steps (no synchronisation inside the loop) against the
device time of the same steps, and a cProfile of the enqueue loop.  If enqueue >= device the step is host-bound.

    python tools/host_overhead.py [--steps 200] [--profile]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sc-sfmlearner-release_amd"))

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--n-ref", type=int, default=2)
    ap.add_argument("--depth", default="smooth")
    ap.add_argument("--dataset", default="kitti")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--single", action="store_true", help="the single-autograd-node step (compute_total_loss)")
    a = ap.parse_args()
    import loss_functions as LF
    dev = torch.device("cuda:0")
    x, _ = bench.make_inputs(a, 0, dev)
    flags = (1, 1, 1, "zeros")
    if a.single:
        bench.hot_path_step = bench.hot_path_step_single_node
    for _ in range(20):
        bench.hot_path_step(LF, x, flags)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        bench.hot_path_step(LF, x, flags)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0) / a.steps:.4f} ms/step, until device idle {1e3 * (t2 - t0) / a.steps:.4f} ms/step")
    if a.profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(a.steps):
            bench.hot_path_step(LF, x, flags)
        pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr)
        st.sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
