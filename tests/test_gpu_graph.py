"""The whole loss-path step (forward + backward) captured into a HIP graph (scsfm_hip.graphs.GraphedStep)
gives the eager results, and a replay follows inputs that were updated in place.  Runs in a child process:
a failure inside graph capture can take the interpreter down with it."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sc-sfmlearner-release_amd")

CHILD = textwrap.dedent("""
    import gc, sys, torch
    sys.path.insert(0, %r)
    import loss_functions as LF
    from scsfm_hip import synth
    from scsfm_hip.graphs import GraphedStep
    dev = torch.device("cuda:0")
    d = synth.make_batch(3, 96, 160, n_ref=2, seed=5, depth="smooth", image="smooth", dataset="kitti")
    to = lambda t: t.to(dev)
    tgt, refs, K = to(d["tgt_img"]), [to(t) for t in d["ref_imgs"]], to(d["intrinsics"])
    td = [to(t).requires_grad_(True) for t in d["tgt_depth"]]
    rd = [[to(t).requires_grad_(True) for t in r] for r in d["ref_depths"]]
    ps = [to(t).requires_grad_(True) for t in d["poses"]]
    pi = [to(t).requires_grad_(True) for t in d["poses_inv"]]
    leaves = td + [t for r in rd for t in r] + ps + pi

    def step():
        for t in leaves:
            t.grad = None
        photo, geom = LF.compute_photo_and_geometry_loss(tgt, refs, K, td, rd, ps, pi, 1, 1, 1, 1, "zeros")
        smooth = LF.compute_smooth_loss(td, tgt, rd, refs)
        loss = 1.0 * photo + 0.1 * smooth + 0.5 * geom
        loss.backward()
        return loss

    def snapshot(loss):
        return [float(loss.detach())] + [t.grad.detach().clone() for t in leaves]

    def check(a, b, what):
        assert abs(a[0] - b[0]) <= 1e-6 * max(1.0, abs(b[0])), (what, a[0], b[0])
        for x, y in zip(a[1:], b[1:]):
            # the scatter's atomics make the last bits order-dependent
            assert float((x - y).abs().max()) <= 1e-5 * max(1e-12, float(y.abs().max())), what

    ref0 = snapshot(step())
    gc.collect()
    g = GraphedStep(step)
    check(snapshot(g.replay()), ref0, "replay vs eager")
    with torch.no_grad():                      # new data in the same tensors
        td[0].mul_(1.07)
        ps[0].add_(0.01)
    got = snapshot(g.replay())
    for t in leaves:
        t.grad = None
    want = snapshot(step())
    check(got, want, "replay after in-place update vs eager")
    assert abs(got[0] - ref0[0]) > 1e-7      # and the update did change the result
    print("GRAPH-OK")
""") % PKG


def test_graphed_step_matches_eager():
    out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, PYTHONPATH=PKG))
    assert out.returncode == 0 and "GRAPH-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
