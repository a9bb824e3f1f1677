#!/usr/bin/env python3
"""N eager single-node steps (loss_functions.compute_total_loss + backward) and nothing else -- run under
`rocprofv3 --kernel-trace --stats` to count the library's launches per step (tools/gpu_round.sh).
    python tools/step_launches.py [--steps 20]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sc-sfmlearner-release_amd"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    for k, v in (("batch", 12), ("height", 256), ("width", 832), ("n_ref", 2), ("steps", 20)):
        ap.add_argument("--" + k.replace("_", "-"), type=int, default=v)
    ap.add_argument("--dataset", default="kitti")
    ap.add_argument("--depth", default="smooth")
    a = ap.parse_args()
    import loss_functions as LF
    x, _ = bench.make_inputs(a, 0, torch.device("cuda:0"))
    for _ in range(a.steps):
        bench.hot_path_step_single_node(LF, x, (1, 1, 1, "zeros"))
    torch.cuda.synchronize()
    print("steps", a.steps)


if __name__ == "__main__":
    main()
