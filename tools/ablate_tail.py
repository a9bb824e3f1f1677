#!/usr/bin/env python3
"""Where does the speculative forward spend its time?  Times the fused kernel alone (SCSFM_DEBUG_KERNEL_ONLY) with the
profiling switches of include/scsfm_hip.h, on BOTH instantiations: "ct" = the product (training flags as template
arguments), "rt" = the runtime-flag instantiation (debug bit 2048).  Timings only -- with a switch set the results are
wrong by construction.  Schema of profiles/r0N_ablation.json: {"<depth>/<base>/<case>": us}.

    python tools/ablate_tail.py [--depths smooth,iid,scene] [--rounds 3]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sc-sfmlearner-release_amd"))

import bench  # noqa: E402

CASES = {"full": 0, "no_lds_scatter(X1)": 1024, "no_flush(X5)": 32768, "no_tail_pixels(X4)": 8192}
BASES = {"ct": 0, "rt": 2048}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--n-ref", type=int, default=2)
    ap.add_argument("--depths", default="smooth,iid")
    ap.add_argument("--dataset", default="kitti")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3, help="alternations; the minimum over rounds is reported next to all")
    ap.add_argument("--ssim", type=int, default=1)
    a = ap.parse_args()
    from scsfm_hip import _lib, capi
    lib = _lib.get()
    dev = torch.device("cuda:0")
    out, every = {}, {}
    for depth in a.depths.split(","):
        a.depth = depth
        x, _ = bench.make_inputs(a, 0, dev)
        fl = capi.make_flags(a.ssim, 1, 1, "zeros")
        det = lambda t: t.detach()
        tgt, K, refs = x["tgt_img"], x["K"], x["ref_imgs"]
        tds, rds = [det(x["tgt_depth"][0])], [[det(r[0])] for r in x["ref_depths"]]
        ps, pis = [det(p) for p in x["poses"]], [det(p) for p in x["poses_inv"]]
        _, _, _, ws = capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=(1.0, 0.5))
        for r in range(a.rounds):
            for base, bbit in BASES.items():
                for name, extra in CASES.items():
                    fn = lambda: capi.photo_geometry_fwd(lib, fl | extra | bbit | 16384, tgt, K, refs, tds, rds, ps, pis,
                                                         hint=(1.0, 0.5), ws=ws)
                    every.setdefault(f"{depth}/{base}/{name}", []).append(round(bench._event_time(fn, a.iters) * 1e6, 1))
    out = {k: min(v) for k, v in every.items()}
    print(json.dumps({"lib": os.path.basename(lib.path), "source_id": lib.source_id(), "us_min": out, "us_all": every}))


if __name__ == "__main__":
    main()
