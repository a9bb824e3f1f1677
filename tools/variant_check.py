#!/usr/bin/env python3
"""One tuning variant (SCSFM_HIP_LIB=variants/X.so) on the bench workload: kernel-only time of the speculative forward
(smooth and iid depth) and, for comparing variants with each other, the step's losses and gradient checksums.
    SCSFM_HIP_LIB=variants/f1.so python tools/variant_check.py [--iters 40] [--rounds 2]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sc-sfmlearner-release_amd"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    for k, v in (("batch", 12), ("height", 256), ("width", 832), ("n_ref", 2), ("iters", 40), ("rounds", 2)):
        ap.add_argument("--" + k.replace("_", "-"), type=int, default=v)
    ap.add_argument("--dataset", default="kitti")
    ap.add_argument("--depths", default="smooth,iid")
    ap.add_argument("--extra", type=int, default=int(os.environ.get("VARIANT_EXTRA", "0")),
                    help="1: also time the plain forward and the fallback backward")
    a = ap.parse_args()
    from scsfm_hip import _lib, capi
    import loss_functions as LF
    lib = _lib.get()
    dev = torch.device("cuda:0")
    out = {"lib": os.path.basename(lib.path), "us": {}, "check": {}}
    # -DSCSFM_RGBD builds (round 6) read the reference frames through [B, H, W, 4] texel planes registered here
    register = getattr(lib._dll, "scsfm_debug_register_texels", None) if hasattr(lib._dll, "scsfm_debug_register_texels") else None
    keep = []
    for depth in a.depths.split(","):
        a.depth = depth
        x, _ = bench.make_inputs(a, 0, dev)
        if register is not None:
            import ctypes
            register.argtypes = [ctypes.c_void_p] * 3
            register(None, None, None)
            frames = [(x["tgt_img"], x["tgt_depth"][0])] + [(i, r[0]) for i, r in zip(x["ref_imgs"], x["ref_depths"])]
            pack = lambda: [torch.cat([i, d.detach()], 1).permute(0, 2, 3, 1).contiguous() for i, d in frames]
            tex = pack()
            keep.append(tex)
            for (i, d), t in zip(frames, tex):
                assert register(i.data_ptr(), d.data_ptr(), t.data_ptr()) == 0
            out.setdefault("us_pack_torch", {})[depth] = round(bench._event_time(pack, 10) * 1e6, 1)
        loss, photo, smooth, geom = bench.hot_path_step(LF, x, (1, 1, 1, "zeros"))
        gs = [x["tgt_depth"][0].grad] + [r[0].grad for r in x["ref_depths"]]
        ps = x["poses"] + x["poses_inv"]
        out["check"][depth] = {"photo": float(photo), "geom": float(geom),
                               "gd_sum": [float(g.double().sum()) for g in gs], "gd_abs": [float(g.double().abs().sum()) for g in gs],
                               "gpose": [float(p.grad.double().abs().sum()) for p in ps]}
        det = lambda t: t.detach()
        tgt, K, refs = x["tgt_img"], x["K"], x["ref_imgs"]
        tds, rds = [det(x["tgt_depth"][0])], [[det(r[0])] for r in x["ref_depths"]]
        pp, pis = [det(p) for p in x["poses"]], [det(p) for p in x["poses_inv"]]
        fl = capi.make_flags(1, 1, 1, "zeros")
        _, _, _, ws = capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, pp, pis, hint=(1.0, 0.5))
        fn = lambda: capi.photo_geometry_fwd(lib, fl | 16384, tgt, K, refs, tds, rds, pp, pis, hint=(1.0, 0.5), ws=ws)
        out["us"][depth] = [round(bench._event_time(fn, a.iters) * 1e6, 1) for _ in range(a.rounds)]
        if hasattr(capi, "smooth_rides_along"):  # (round 6) ... with the frames' smooth loss riding in the tiles
            fr = lambda: capi.photo_geometry_fwd(lib, fl | 16384, tgt, K, refs, tds, rds, pp, pis, hint=(1.0, 0.5), ws=ws, smooth=True)
            out.setdefault("us_ride", {})[depth] = [round(bench._event_time(fr, a.iters) * 1e6, 1) for _ in range(a.rounds)]
        if a.extra or register is not None:
            # the smooth loss's forward over the step's frames; with texel planes registered it packs them on the way
            sm_frames, sm_imgs = tds + [r[0] for r in rds], [tgt] + list(refs)
            sm = lambda: capi.smooth_multi_fwd(lib, sm_frames, sm_imgs)
            out.setdefault("us_smooth_fwd", {})[depth] = [round(bench._event_time(sm, a.iters) * 1e6, 1) for _ in range(a.rounds)]
            if register is not None:
                for t_ in keep[-1]:
                    t_.fill_(float("nan"))
                sm()
                ref_tex = pack()
                out.setdefault("pack_matches", {})[depth] = bool(all(torch.equal(p_, q_) for p_, q_ in zip(keep[-1], ref_tex)))
                register(None, None, None)
                out.setdefault("us_smooth_fwd_no_pack", {})[depth] = [round(bench._event_time(sm, a.iters) * 1e6, 1) for _ in range(a.rounds)]
        if a.extra:
            # the plain forward (validation path: prep + pair_fwd_kernel + finalize) and the backward of the first step
            # after a weight change (guards fail: pair_bwd_photo + pair_bwd_geom + combine)
            one, third = torch.tensor([1.0], device=dev), torch.tensor([0.3], device=dev)
            plain = lambda: capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, pp, pis, hint=None)
            stale = lambda: capi.photo_geometry_bwd(lib, fl, tgt, K, refs, tds, rds, pp, pis, ws, one, third)
            out.setdefault("us_plain_fwd", {})[depth] = [round(bench._event_time(plain, a.iters) * 1e6, 1) for _ in range(a.rounds)]
            out.setdefault("us_fallback_bwd", {})[depth] = [round(bench._event_time(stale, a.iters) * 1e6, 1) for _ in range(a.rounds)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
