#!/usr/bin/env python3
"""Stage timeline of the speculative forward's column march from a -DPROBE_TIMING build (wave 0's clock at the stage
boundaries of every chunk, tools/build_variants.sh).   SCSFM_HIP_LIB=variants/b3time.so python tools/march_timing.py"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sc-sfmlearner-release_amd"))
import bench  # noqa: E402

# stamps in the order they are taken, and what the interval that ENDS at each one is
ORDER = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23]
NAMES = ["W", "box", "bar1", "S0", "bar2.0", "O0", "bar3.0", "S1", "bar2.1", "O1", "bar3.1", "S2", "bar2.2", "O2", "(to A)",
         "barA", "zero+box", "barB", "tail", "barC", "flush"]


def main():
    ap = argparse.ArgumentParser()
    for k, v in (("batch", 12), ("height", 256), ("width", 832), ("n_ref", 2)):
        ap.add_argument("--" + k.replace("_", "-"), type=int, default=v)
    ap.add_argument("--dataset", default="kitti")
    ap.add_argument("--depth", default="smooth")
    a = ap.parse_args()
    from scsfm_hip import _lib, capi
    lib = _lib.get()
    dev = torch.device("cuda:0")
    x, _ = bench.make_inputs(a, 0, dev)
    det = lambda t: t.detach()
    tgt, K, refs = x["tgt_img"], x["K"], x["ref_imgs"]
    tds, rds = [det(x["tgt_depth"][0])], [[det(r[0])] for r in x["ref_depths"]]
    ps, pis = [det(p) for p in x["poses"]], [det(p) for p in x["poses_inv"]]
    fl = capi.make_flags(1, 1, 1, "zeros")
    for _ in range(3):
        capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=(1.0, 0.5))
    torch.cuda.synchronize()
    NS, NC, NWG = 24, 6, 4096
    buf = np.zeros(NWG * NC * NS, dtype=np.uint64)
    fn = lib._dll.scsfm_probe_read
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    rc = fn(buf.ctypes.data, buf.size)
    assert rc == 0, rc
    t = buf.reshape(NWG, NC, NS).astype(np.int64)
    if os.environ.get("SCSFM_SPEC_KERNEL", "")[:1] == "m":
        sys.exit("the column march's stamps were taken out with its per-colour rewrite; the timelines in "
                 "profiles/r03_march_experiments.json were recorded at commits 1caf5b2 .. fd08494")
    if os.environ.get("SCSFM_SPEC_KERNEL", "")[:1] != "m":  # the tile kernel: every third tile stamps 9 times
        names = ["loads+warp", "ring warp", "barrier", "3 x (S, O)", "block sum", "stage taps", "tail", "block sum 12", "flush"]
        m = t[:, 0, 8] > 0
        d = np.diff(t[m, 0, :9], axis=1)
        print(json.dumps({"tiles": int(m.sum()), "total": float((t[m, 0, 8] - t[m, 0, 0]).mean()),
                          "stages": {names[i]: round(float(d[:, i].mean())) for i in range(8)}}))
        return
    used = t[:, :, 0] > 0
    out = {}
    for chunk in range(NC):
        m = used[:, chunk] & (t[:, chunk, 23] > 0)
        if m.sum() == 0:
            continue
        d = np.diff(t[m, chunk, :][:, ORDER], axis=1)
        out[f"chunk{chunk}"] = {"n": int(m.sum()), "total": float((t[m, chunk, 23] - t[m, chunk, 0]).mean()),
                                "stages": {NAMES[i]: round(float(d[:, i].mean())) for i in range(len(ORDER) - 1)}}
    # wall time of a workgroup and the spread of start times
    m = used[:, 0]
    life = t[m][:, :, 23].max(axis=1) - t[m, 0, 0]
    out["wg_life_mean_cycles"] = float(life.mean())
    out["start_spread_cycles"] = float(t[m, 0, 0].max() - t[m, 0, 0].min())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
