#!/usr/bin/env python3
"""The speculative forward alone: the compile-time-flag instantiation (training flags) against the runtime-flag one
(selected by an otherwise unused debug bit), alternating in one process."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sc-sfmlearner-release_amd"))
import argparse, bench
a = argparse.Namespace(batch=12, height=256, width=832, n_ref=2, dataset="kitti", depth=os.environ.get("DEPTH", "smooth"))
from scsfm_hip import _lib, capi
lib = _lib.get()
x, _ = bench.make_inputs(a, 0, torch.device("cuda:0"))
det = lambda t: t.detach()
tgt, K, refs = x["tgt_img"], x["K"], x["ref_imgs"]
tds, rds = [det(x["tgt_depth"][0])], [[det(r[0])] for r in x["ref_depths"]]
ps, pis = [det(p) for p in x["poses"]], [det(p) for p in x["poses_inv"]]
fl = capi.make_flags(1, 1, 1, "zeros")
_, _, _, ws = capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=(1.0, 0.5))
out = {"ct": [], "rt": []}
for r in range(6):
    for name, extra in (("ct", 0), ("rt", 2048)):
        fn = lambda: capi.photo_geometry_fwd(lib, fl | extra | 16384, tgt, K, refs, tds, rds, ps, pis, hint=(1.0, 0.5), ws=ws)
        out[name].append(round(bench._event_time(fn, 40) * 1e6, 1))
print(json.dumps(out))
