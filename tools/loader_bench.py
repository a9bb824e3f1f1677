#!/usr/bin/env python3
"""Input pipeline of the training loop measured on its own: samples / s through a SequenceFolder
tree of JPEG frames on local disk -- decode + the reference's transform chain (RandomHorizontalFlip, RandomScaleCrop with
Pillow's bicubic resize, ArrayToTensor, Normalize; train.py:95-100, custom_transforms.py:33-84) in the data-loader
workers, then the host-to-device copy -- against `--gpu-augment`: the workers only decode, the uint8 frames are copied to
the device and the transform chain runs there in one kernel (csrc/scsfm_augment.hip, byte-exact).  Also the device
transform alone (HIP events) as GB/s against its 15 B per output pixel (3 read, 12 written).

    python tools/loader_bench.py [--workers 4,16] [--seconds 8] [--batch 12] [--height 256 --width 832]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sc-sfmlearner-release_amd"))


def run(batch=12, height=256, width=832, seq=3, workers=(4, 16), seconds=8.0, device="cuda:0"):
    import custom_transforms as CT
    from datasets.sequence_folders import SequenceFolder
    from datasets.synthetic import write_sequence_tree
    from scsfm_hip import augment as A
    root = tempfile.mkdtemp(prefix="scsfm_seq_")
    write_sequence_tree(root, n_scenes=6, frames_per_scene=32, height=height, width=width, seed=0, with_depth=False)
    normalize = CT.Normalize(mean=[0.45, 0.45, 0.45], std=[0.225, 0.225, 0.225])
    chains = {"cpu_augment": CT.Compose([CT.RandomHorizontalFlip(), CT.RandomScaleCrop(), CT.ArrayToTensor(), normalize]),
              "gpu_augment": CT.Compose([CT.ArrayToUint8()])}
    dev = torch.device(device)
    out = {"batch": batch, "shape": [height, width], "sequence_length": seq, "host_cores": os.cpu_count(), "legs": []}
    for nw in workers:
        for name, tf in chains.items():
            ds = SequenceFolder(root, transform=tf, seed=0, train=True, sequence_length=seq)
            dl = torch.utils.data.DataLoader(ds, batch_size=batch, shuffle=True, num_workers=nw, pin_memory=True, drop_last=True,
                                             persistent_workers=nw > 0)
            n, t0, warm = 0, None, 2
            while t0 is None or time.perf_counter() - t0 < seconds:
                for tgt, refs, K, _ in dl:
                    tgt = tgt.to(dev, non_blocking=True)
                    refs = [r.to(dev, non_blocking=True) for r in refs]
                    if name == "gpu_augment":
                        frames = torch.stack([tgt] + refs, dim=1).contiguous()
                        recs = A.draw_params(frames.shape[0], height, width)
                        imgs = A.augment(frames, recs)
                        K = torch.from_numpy(A.update_intrinsics(K.numpy(), recs, width))
                    K = K.to(dev, non_blocking=True)
                    torch.cuda.synchronize()
                    if warm > 0:
                        warm -= 1
                        if warm == 0:
                            t0, n = time.perf_counter(), 0
                        continue
                    n += batch
                    if time.perf_counter() - t0 >= seconds:
                        break
            dt = time.perf_counter() - t0
            out["legs"].append({"pipeline": name, "workers": nw, "samples_per_s": round(n / dt, 1), "seconds": round(dt, 2)})
            del dl
    # the device transform alone
    frames = torch.randint(0, 256, (batch, seq, height, width, 3), dtype=torch.uint8, device=dev)
    recs = A.draw_params(batch, height, width)
    for _ in range(3):
        A.augment(frames, recs)
    params, htab, vtab = A.tables(recs, height, width)
    to = lambda a: torch.from_numpy(a).to(dev)
    p, h, v, lut = to(params), to(htab), to(vtab), to(A.byte_lut())
    dst = torch.empty(seq, batch, 3, height, width, dtype=torch.float32, device=dev)
    from scsfm_hip import _lib, capi
    lib = _lib.get()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    iters = 50
    e0.record()
    for _ in range(iters):
        lib.call("scsfm_augment_u8_f32", batch * seq, seq, height, width, frames.data_ptr(), p.data_ptr(), h.data_ptr(), v.data_ptr(),
                 lut.data_ptr(), dst.data_ptr(), capi._stream(frames))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    npx = batch * seq * height * width
    out["augment_kernel"] = {"us_per_batch": round(us, 2), "algorithmic_bytes": npx * 15, "GBs": round(npx * 15 / us / 1e3, 1),
                             "frac_of_hbm_peak": round(npx * 15 / us / 1e3 / 8000.0, 4), "samples_per_s_kernel_only": round(batch / us * 1e6)}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--workers", default="4,16")
    ap.add_argument("--seconds", type=float, default=8.0)
    a = ap.parse_args()
    print(json.dumps(run(a.batch, a.height, a.width, 3, tuple(int(w) for w in a.workers.split(",")), a.seconds)))
