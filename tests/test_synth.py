"""The synthetic input laws (scsfm_hip.synth): seeded, reproducible, and -- for the `scene` law of round 5 -- with the
statistics it promises: piecewise-smooth depth, occlusion edges on about 5 % of the pixels, disparity jumps of roughly
5 .. 50 px across them for the bench's motion, image edges where the depth's are."""
import numpy as np
import pytest
import torch

from scsfm_hip import synth


@pytest.mark.parametrize("depth", ["smooth", "iid", "scene"])
def test_batches_are_reproducible_and_in_range(depth):
    a = synth.make_batch(2, 64, 96, n_ref=2, seed=5, depth=depth, image=synth.image_law(depth), num_scales=2)
    b = synth.make_batch(2, 64, 96, n_ref=2, seed=5, depth=depth, image=synth.image_law(depth), num_scales=2)
    flat = lambda d: [d["tgt_img"], *d["ref_imgs"], d["intrinsics"], *d["tgt_depth"], *[t for r in d["ref_depths"] for t in r],
                      *d["poses"], *d["poses_inv"]]
    assert all(torch.equal(x, y) for x, y in zip(flat(a), flat(b)))
    assert [tuple(t.shape) for t in a["tgt_depth"]] == [(2, 1, 64, 96), (2, 1, 32, 48)]
    for t in a["tgt_depth"] + [t for r in a["ref_depths"] for t in r]:
        assert float(t.min()) >= 0.0999 and float(t.max()) <= 100.0 + 1e-3  # DispResNet's range
    for t in [a["tgt_img"]] + a["ref_imgs"]:
        assert -2.001 <= float(t.min()) and float(t.max()) <= 2.45


def test_scene_law_statistics():
    d = synth.make_batch(4, 256, 832, n_ref=2, seed=0, depth="scene", image="scene")
    K = d["intrinsics"]
    for dm in [d["tgt_depth"][0]] + [r[0] for r in d["ref_depths"]]:
        frac = synth.scene_edge_fraction(dm)
        assert 0.03 <= frac <= 0.08, frac
    # disparity jumps at the edges for the bench's motion: f |t| px per unit of inverse depth
    inv = 1.0 / d["tgt_depth"][0]
    dx = (inv[..., :, 1:] - inv[..., :, :-1]).abs()
    jumps = dx[dx > 0.3].numpy()
    ft = float(d["poses"][0][:, :3].norm(dim=1).mean() * K[0, 0, 0])
    lo, med, hi = np.quantile(jumps, [0.05, 0.5, 0.95]) * ft
    assert 2.0 <= lo and 8.0 <= med <= 30.0 and hi <= 70.0, (lo, med, hi)
    # away from the edges the map is as smooth as the `smooth` law's background (whose 1/Z changes by < 0.31 per pixel)
    assert float(dx[dx <= 0.3].median()) < 0.1 and float((dx <= 0.3).double().mean()) >= 0.95
    # the image has an edge where the depth has one (mean colour step across depth edges >> elsewhere)
    img = d["tgt_img"]
    step = (img[..., :, 1:] - img[..., :, :-1]).abs().mean(dim=1, keepdim=True)
    at_edges, elsewhere = float(step[dx > 0.3].mean()), float(step[dx <= 0.3].mean())
    assert at_edges > 1.5 * elsewhere, (at_edges, elsewhere)


def test_scene_image_needs_scene_depth():
    with pytest.raises(ValueError):
        synth.make_batch(1, 32, 48, depth="smooth", image="scene")
