"""The fp32 speculative forward -- the kernel that ships: 40 KB of aliased LDS (kLean), the tail's LDS-staged taps
(kStage), the fixed-point scatter window -- in the CPU simulation, against the fp64 oracle; and its variant with the
FORWARD warp's taps staged in LDS as well (variants/src/scsfm_spec_stagefwd.inc, kStageFwd; SCSFM_SPEC_KERNEL=stagefwd in builds
with -DSCSFM_WITH_MARCH, which the simulation's build defines; profiles/HISTORY.md 3a: measured 2.4 % slower, not the
product) against both the oracle and the product kernel: the two evaluate the same arithmetic on the same texels, so
the forward sums and the pose gradients must agree bit for bit; only the placement of the scatter window differs
(fixed-point cells vs fp32 atomics for some taps), which shows in the last bits of the depth gradients."""
import pytest
import torch

from _util import assert_close_frac, leaf
from hostsim import harness
from oracle import scsfm_oracle as O
from scsfm_hip import capi, synth

CASES = [  # B, H, W, depth, padding (the 10000-pixel gates are open in every case)
    (4, 72, 100, "smooth", "zeros"),
    (4, 72, 100, "iid", "zeros"),      # landing positions spread over more than a window: nothing is staged
    (4, 72, 100, "smooth", "border"),  # (runtime-flag instantiation)
    (24, 15, 63, "smooth", "zeros"),   # lower than the staged window and narrower: clamped rows / columns
    (5, 41, 150, "smooth", "zeros"),   # partial tiles in both directions
]


@pytest.fixture(scope="module")
def lib():
    return harness.lib()


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))


def _run(lib, d, pad, pose_gain=1.0):
    ti, K, ris = d["tgt_img"], d["intrinsics"], d["ref_imgs"]
    tds, rds = [d["tgt_depth"][0]], [[r[0]] for r in d["ref_depths"]]
    ps, pis = [pose_gain * p for p in d["poses"]], [pose_gain * p for p in d["poses_inv"]]
    fl = capi.make_flags(1, 1, 1, pad)
    photo, geom, _, ws = capi.photo_geometry_fwd(lib, fl, ti, K, ris, tds, rds, ps, pis, hint=(1.0, 0.5))
    g = capi.photo_geometry_bwd(lib, fl, ti, K, ris, tds, rds, ps, pis, ws, torch.tensor([1.0]), torch.tensor([0.5]))
    return photo.clone(), geom.clone(), g


def _oracle(d, pad, pose_gain=1.0):
    c = lambda x: x.double().contiguous()
    td = [leaf(c(d["tgt_depth"][0]))]
    rd = [[leaf(c(r[0]))] for r in d["ref_depths"]]
    pp, pi = [leaf(c(pose_gain * p)) for p in d["poses"]], [leaf(c(pose_gain * p)) for p in d["poses_inv"]]
    po, go = O.photo_and_geometry_loss(c(d["tgt_img"]), [c(r) for r in d["ref_imgs"]], c(d["intrinsics"]), td, rd, pp, pi,
                                       1, 1, 1, 1, pad)
    (1.0 * po + 0.5 * go).backward()
    return po, go, td, rd, pp, pi


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(map(str, c)))
def test_fp32_speculative_forward_and_its_staged_forward_variant(lib, monkeypatch, case):
    B, H, W, depth, pad = case
    d = synth.make_batch(B, H, W, n_ref=2, seed=7 + H, depth=depth)
    po, go, td, rd, pp, pi = _oracle(d, pad)
    assert float(po) > 0 and float(go) > 0
    monkeypatch.delenv("SCSFM_SPEC_KERNEL", raising=False)
    tile = _run(lib, d, pad)
    monkeypatch.setenv("SCSFM_SPEC_KERNEL", "stagefwd")
    staged = _run(lib, d, pad)
    # entry-wise, but for the few pixels whose valid / auto-mask / clamp decision rounds the other way in fp32
    # (tests/test_gpu_parity.py looks at those through the oracle's gate margins)
    bad = 2e-3 if depth == "smooth" else 3e-2

    def close(a, ref, what):
        ref = ref.numpy()
        assert_close_frac(a.numpy(), ref, atol=2e-3 * abs(ref).max(), rtol=1e-3, max_bad_frac=bad, what=what)

    for name, (photo, geom, (g_td, g_rd, g_p, g_pi)) in (("tile", tile), ("stagefwd", staged)):
        assert abs(float(photo) - float(po)) <= 1e-5 and abs(float(geom) - float(go)) <= 1e-5
        close(g_td[0], td[0].grad, name + " tgt depth")
        for i in range(2):
            close(g_rd[i][0], rd[i][0].grad, f"{name} ref depth {i}")
            assert _rel(g_p[i], pp[i].grad) < 5e-2 and _rel(g_pi[i], pi[i].grad) < 5e-2  # (sums dominated by a few near pixels)
    # the variant against the product kernel: same arithmetic on the same texels (every depth map's gradient holds
    # scattered contributions of the pairs that sampled it: the window's placement shows in the last bits)
    assert torch.equal(tile[0], staged[0]) and torch.equal(tile[1], staged[1])
    assert _rel(staged[2][0][0], tile[2][0][0]) < 1e-5
    for i in range(2):
        assert torch.equal(tile[2][2][i], staged[2][2][i]) and torch.equal(tile[2][3][i], staged[2][3][i])
        assert _rel(staged[2][1][i][0], tile[2][1][i][0]) < 1e-5


def test_staged_window_misses_fall_back_to_gathers(lib, monkeypatch):
    """Ten times the usual camera motion: the landing positions of a tile stretch beyond the 72 x 20 window (part of
    the pixels read LDS, the others gather) or leave the image."""
    d = synth.make_batch(4, 72, 100, n_ref=2, seed=19, depth="smooth")
    monkeypatch.delenv("SCSFM_SPEC_KERNEL", raising=False)
    tile = _run(lib, d, "zeros", pose_gain=10.0)
    monkeypatch.setenv("SCSFM_SPEC_KERNEL", "stagefwd")
    staged = _run(lib, d, "zeros", pose_gain=10.0)
    assert torch.equal(tile[0], staged[0]) and torch.equal(tile[1], staged[1])
    assert _rel(staged[2][0][0], tile[2][0][0]) < 1e-5
    for i in range(2):
        assert torch.equal(tile[2][2][i], staged[2][2][i]) and torch.equal(tile[2][3][i], staged[2][3][i])
        assert _rel(staged[2][1][i][0], tile[2][1][i][0]) < 1e-5


def test_wide_scatter_window_on_incoherent_depth(lib, monkeypatch):
    """Footprints more than twice as tall as the scatter window (iid depth on an image tall enough for taps +-48 rows
    away): the tile kernel widens the window with its three staging regions (csrc/scsfm_geom.h: WideWin) and does
    without staged taps.  Against the fp64 oracle with the allowance of the iid cases, and against the backward's own
    two passes (plain forward, floating-point window of the normal size), which share no scatter code with it."""
    monkeypatch.delenv("SCSFM_SPEC_KERNEL", raising=False)
    d = synth.make_batch(3, 160, 100, n_ref=2, seed=23, depth="iid", image="iid")
    po, go, td, rd, pp, pi = _oracle(d, "zeros")
    assert float(po) > 0 and float(go) > 0
    photo, geom, g = _run(lib, d, "zeros")
    assert abs(float(photo) - float(po)) <= 1e-5 and abs(float(geom) - float(go)) <= 1e-5
    ti, K, ris = d["tgt_img"], d["intrinsics"], d["ref_imgs"]
    tds, rds = [d["tgt_depth"][0]], [[r[0]] for r in d["ref_depths"]]
    fl = capi.make_flags(1, 1, 1, "zeros")
    _, _, _, ws = capi.photo_geometry_fwd(lib, fl, ti, K, ris, tds, rds, d["poses"], d["poses_inv"])  # no speculation
    g2 = capi.photo_geometry_bwd(lib, fl, ti, K, ris, tds, rds, d["poses"], d["poses_inv"], ws, torch.tensor([1.0]),
                                 torch.tensor([0.5]))
    assert _rel(g[0][0], g2[0][0]) < 2e-4
    for i in range(2):
        assert _rel(g[1][i][0], g2[1][i][0]) < 2e-4
        assert _rel(g[2][i], g2[2][i]) < 1e-3
        ref = rd[i][0].grad.numpy()
        assert_close_frac(g[1][i][0].numpy(), ref, atol=2e-3 * abs(ref).max(), rtol=1e-3, max_bad_frac=3e-2, what=f"ref depth {i}")
