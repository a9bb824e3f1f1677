"""Helpers shared by the test modules: golden loading and tolerant comparisons."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def load_inputs(name):
    """inputs_<name>.npz -> the argument structure of the reference's loss calls (torch, CPU)."""
    z = load_npz(f"inputs_{name}.npz")
    n_ref = sum(1 for k in z if k.startswith("ref_img"))
    n_scales = sum(1 for k in z if k.startswith("tgt_depth_s"))
    t = lambda k: torch.from_numpy(z[k])
    return {
        "tgt_img": t("tgt_img"),
        "ref_imgs": [t(f"ref_img{i}") for i in range(n_ref)],
        "intrinsics": t("K"),
        "tgt_depth": [t(f"tgt_depth_s{s}") for s in range(n_scales)],
        "ref_depths": [[t(f"ref{i}_depth_s{s}") for s in range(n_scales)] for i in range(n_ref)],
        "poses": [t(f"pose{i}") for i in range(n_ref)],
        "poses_inv": [t(f"pose_inv{i}") for i in range(n_ref)],
    }


def probe(n):
    return torch.cos(0.37 * torch.arange(n, dtype=torch.float64))


def grad_stats(g):
    f = g.detach().double().reshape(-1).cpu()
    return np.array([f.sum().item(), f.abs().sum().item(), (f * probe(f.numel())).sum().item()])


def leaf(t):
    return t.clone().requires_grad_(True)


def assert_close_frac(a, b, atol, rtol=0.0, max_bad_frac=0.0, what=""):
    """|a-b| <= atol + rtol*|b| for all but a fraction ``max_bad_frac`` of the entries.  The
    fraction allows for the discontinuous gates of the path (valid / auto mask, clamps): a 1-ulp
    difference in a coordinate can flip a pixel."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = np.abs(a - b) > (atol + rtol * np.abs(b))
    frac = bad.mean() if bad.size else 0.0
    assert frac <= max_bad_frac, f"{what}: {bad.sum()} of {bad.size} entries differ (max |d|={np.abs(a-b).max():.3e})"


REPORT_LINES = []


def report(line):
    """A line for the session's terminal summary (tests/conftest.py: pytest_terminal_summary)."""
    REPORT_LINES.append(line)
    print(line)


def nonuniform_case(B=2, seed=71):
    """Round-5 review: a NON-uniform compression inside a large footprint, which the bounding-box heuristic of round 5
    could not see.  Forward motion tz; the left half of every row is very near (depth << tz: those pixels collapse onto
    a few texels around the principal point, their unscaled scatter terms 2 Z / (Z + D_p)^2 ~ 25 each), the right half is
    far (depth >> tz: they stay where they are and spread over hundreds of cells).  Photo-only upstream gradient: all
    terms of one sign."""
    import torch
    from scsfm_hip import synth
    H, W = 72, 100
    d = synth.make_batch(B, H, W, n_ref=1, seed=seed, depth="smooth")
    g = torch.Generator().manual_seed(9)
    tz = 0.06
    near = 0.002 + 0.002 * torch.rand(B, 1, H, W, generator=g)
    far = 5.0 + 2.0 * torch.rand(B, 1, H, W, generator=g)
    left = (torch.arange(W) % 2 == 0).view(1, 1, 1, W)  # alternate columns: every tile holds both kinds
    tds = [torch.where(left, near, far).contiguous()]
    rds = [[(0.004 + 0.004 * torch.rand(B, 1, H, W, generator=g)).contiguous()]]
    p = torch.zeros(B, 6)
    p[:, 2] = tz
    return d["tgt_img"], d["intrinsics"], d["ref_imgs"], tds, rds, [p], [-p.clone()]
