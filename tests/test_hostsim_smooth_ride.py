"""The smooth loss riding in the speculative forward (round 6; scsfm_pair_desc::smooth_ws, loss_functions.py:132-159 next to
:50-92 as train.py:262-268 calls them) in the CPU simulation of the product sources: against the stand-alone smooth forward
(the edge planes and the per-image records it leaves must be THE SAME -- the backward entry points are shared), against the
fp64 oracle, and with the pair losses of the same call untouched."""
import pytest
import torch

from hostsim import harness
from oracle import scsfm_oracle as O
from scsfm_hip import capi, synth

CASES = [  # B, H, W, n_ref, dtype, depth law
    (4, 72, 100, 2, torch.float32, "smooth"),
    (3, 41, 150, 2, torch.float32, "scene"),   # partial tiles in both directions
    (2, 30, 63, 1, torch.float32, "iid"),      # one reference: two pair-directions, two frames
    (2, 33, 70, 3, torch.float64, "smooth"),   # fp64 instantiation (8-row tiles), three references
]


@pytest.fixture(scope="module")
def lib():
    return harness.lib()


def _inputs(B, H, W, n_ref, dtype, depth):
    d = synth.make_batch(B, H, W, n_ref=n_ref, seed=11 + H, depth=depth)
    c = lambda t: t.to(dtype).contiguous()
    return (c(d["tgt_img"]), c(d["intrinsics"]), [c(r) for r in d["ref_imgs"]], [c(d["tgt_depth"][0])],
            [[c(r[0])] for r in d["ref_depths"]], [c(p) for p in d["poses"]], [c(p) for p in d["poses_inv"]])


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(map(str, c)))
def test_smooth_loss_riding_in_the_speculative_forward(lib, case):
    B, H, W, n_ref, dtype, depth = case
    ti, K, ris, tds, rds, ps, pis = _inputs(B, H, W, n_ref, dtype, depth)
    fl = capi.make_flags(1, 1, 1, "zeros")
    hint = (1.0, 0.5)
    assert capi.smooth_rides_along(fl, ti, tds, rds, hint)
    photo0, geom0, outs0, _ = capi.photo_geometry_fwd(lib, fl, ti, K, ris, tds, rds, ps, pis, hint=hint)
    photo, geom, outs, ws, smooth, sws = capi.photo_geometry_fwd(lib, fl, ti, K, ris, tds, rds, ps, pis, hint=hint, smooth=True)
    # the pair losses do not notice the passenger
    assert torch.equal(outs0, outs) and float(photo0) == float(photo) and float(geom0) == float(geom)
    frames, imgs = tds + [r[0] for r in rds], [ti] + ris
    ref_loss, ref_ws = capi.smooth_multi_fwd(lib, frames, imgs, keep_edges=True)
    nf = len(frames)
    sw_bytes = capi._sizes(lib, B, H, W)[2]
    plane_bytes = ((B * H * W * ti.element_size() + 255) // 256) * 256
    assert sws.numel() == ref_ws.numel() == nf * (sw_bytes + plane_bytes)
    for f in range(nf):
        e = lambda w: w[nf * sw_bytes + f * plane_bytes:nf * sw_bytes + f * plane_bytes + B * H * W * ti.element_size()].view(dtype)
        assert torch.equal(e(sws), e(ref_ws)), f"edge plane of frame {f}"
        s = lambda w: w[f * sw_bytes:f * sw_bytes + 16 * B].view(torch.float64)
        # {mean + 1e-7, L} per image: the same sums in another grouping
        assert torch.allclose(s(sws), s(ref_ws), rtol=1e-6 if dtype == torch.float32 else 1e-13, atol=0)
    tol = 1e-6 if dtype == torch.float32 else 1e-12
    assert abs(float(smooth) - float(ref_loss)) <= tol * max(1.0, abs(float(ref_loss)))
    c = lambda x: x.double()
    want = O.smooth_loss([c(t) for t in tds], c(ti), [[c(r[0])] for r in rds], [c(r) for r in ris])
    assert abs(float(smooth) - float(want)) <= (1e-5 if dtype == torch.float32 else 1e-10)
    # the shared backward takes the workspace from there
    g = torch.ones(1, dtype=dtype)
    ga = capi.smooth_multi_bwd(lib, frames, imgs, sws, g)
    gb = capi.smooth_multi_bwd(lib, frames, imgs, ref_ws, g)
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= (1e-6 if dtype == torch.float32 else 1e-13) * float(b.abs().max())  # (den in another grouping)


def test_step_total_formed_by_the_finalize_launch(lib):
    B, H, W = 4, 72, 100
    ti, K, ris, tds, rds, ps, pis = _inputs(B, H, W, 2, torch.float32, "smooth")
    fl = capi.make_flags(1, 1, 1, "zeros")
    w = (1.0, 0.1, 0.5)
    photo, geom, _, _, smooth, _, out = capi.photo_geometry_fwd(lib, fl, ti, K, ris, tds, rds, ps, pis, hint=(w[0], w[2]),
                                                                smooth=True, step=w)
    want = torch.tensor([w[0] * float(photo) + w[1] * float(smooth) + w[2] * float(geom), float(photo), float(smooth), float(geom)])
    assert torch.allclose(out, want, rtol=1e-6, atol=0)


def test_descriptors_the_fold_cannot_serve_are_rejected(lib):
    """smooth_ws on a plain (non-speculative) forward or on a coarser scale is an argument error, not a silent skip."""
    import ctypes as ct
    B, H, W = 2, 32, 64
    ti, K, ris, tds, rds, ps, pis = _inputs(B, H, W, 1, torch.float32, "smooth")
    ws_bytes, scratch_bytes, sw_bytes = capi._sizes(lib, B, H, W)
    ws = torch.empty(ws_bytes + scratch_bytes, dtype=torch.uint8)
    sws = torch.empty(sw_bytes, dtype=torch.uint8)
    out = torch.empty(8)
    d = (capi.PairDesc * 1)()
    d[0].tgt_img, d[0].ref_img, d[0].tgt_depth, d[0].ref_depth, d[0].pose = ti.data_ptr(), ris[0].data_ptr(), \
        tds[0].data_ptr(), rds[0][0].data_ptr(), ps[0].data_ptr()
    d[0].ws, d[0].out, d[0].smooth_ws = ws.data_ptr(), out.data_ptr(), sws.data_ptr()
    call = lambda wp: lib._fn["scsfm_pairs_fwd_f32"](1, ct.addressof(d), B, H, W, K.data_ptr(), 7, wp, 0.5, 0)
    assert call(1.0) == -1            # no gbuf: the plain forward
    d[0].gbuf = ws.data_ptr() + ws_bytes
    assert call(0.0) == -1            # a gbuf, but nothing to speculate on
    assert call(1.0) == 0             # the speculative forward carries it


# ---- the host mirror: the reference's two calls (train.py:262-266) with the smooth loss found waiting -----------------
@pytest.fixture()
def on_sim(lib, monkeypatch):
    from scsfm_hip import _lib, config, ops
    monkeypatch.setattr(_lib, "get", lambda: lib)
    monkeypatch.setattr(ops, "_need_cuda", lambda *a: None)
    calls = {"smooth_fwd": 0}
    real = capi.smooth_multi_fwd

    def counted(*a, **k):
        calls["smooth_fwd"] += 1
        return real(*a, **k)

    monkeypatch.setattr(capi, "smooth_multi_fwd", counted)
    prev = config.smooth_rides_along()
    yield calls
    config.set_smooth_rides_along(prev)
    ops._SmoothStash.slot = None


def _leaves(B=3, H=50, W=90, n_ref=2, seed=5):
    d = synth.make_batch(B, H, W, n_ref=n_ref, seed=seed, depth="smooth")
    lf = lambda t: t.clone().requires_grad_(True)
    return (d["tgt_img"], d["ref_imgs"], d["intrinsics"], [lf(d["tgt_depth"][0])], [[lf(r[0])] for r in d["ref_depths"]],
            [lf(p) for p in d["poses"]], [lf(p) for p in d["poses_inv"]])


def _reference_calls(x, between=None):
    import loss_functions as LF
    ti, ris, K, td, rd, ps, pis = x
    photo, geom = LF.compute_photo_and_geometry_loss(ti, ris, K, td, rd, ps, pis, 1, 1, 1, 1, "zeros")
    if between is not None:
        between(x)
    smooth = LF.compute_smooth_loss(td, ti, rd, ris)
    (1.0 * photo + 0.1 * smooth + 0.5 * geom).backward()
    grads = [td[0].grad] + [r[0].grad for r in rd] + [p.grad for p in ps + pis]
    return float(photo.detach()), float(smooth.detach()), float(geom.detach()), [g.clone() for g in grads]


def test_compute_smooth_loss_finds_its_result_waiting(on_sim):
    from scsfm_hip import config
    config.set_smooth_rides_along(False)
    want = _reference_calls(_leaves())
    assert on_sim["smooth_fwd"] == 1
    config.set_smooth_rides_along(True)
    got = _reference_calls(_leaves())
    assert on_sim["smooth_fwd"] == 1, "the stand-alone smooth forward ran although the speculative forward carried it"
    assert got[0] == want[0] and got[2] == want[2] and abs(got[1] - want[1]) <= 1e-6 * abs(want[1])
    for a, b in zip(got[3], want[3]):
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())


def test_a_stale_or_foreign_stash_is_not_used(on_sim):
    """An in-place write into a depth map between the two calls, or other tensor objects (even with equal values), miss."""
    from scsfm_hip import config
    config.set_smooth_rides_along(True)

    import loss_functions as LF
    ti, ris, K, td, rd, ps, pis = _leaves()
    LF.compute_photo_and_geometry_loss(ti, ris, K, td, rd, ps, pis, 1, 1, 1, 1, "zeros")
    with torch.no_grad():
        td[0].mul_(1.5)  # the target depth map changes: what waits in the stash is the loss of the OLD map
    got = float(LF.compute_smooth_loss(td, ti, rd, ris))  # (forward only: autograd itself refuses a backward through the old map)
    assert on_sim["smooth_fwd"] == 1
    config.set_smooth_rides_along(False)
    want = float(LF.compute_smooth_loss(td, ti, rd, ris))
    assert abs(got - want) <= 1e-6 * abs(want)  # the smooth loss of the map as compute_smooth_loss saw it
    # other objects: the second call gets clones
    config.set_smooth_rides_along(True)
    ti, ris, K, td, rd, ps, pis = _leaves()
    n0 = on_sim["smooth_fwd"]
    LF.compute_photo_and_geometry_loss(ti, ris, K, td, rd, ps, pis, 1, 1, 1, 1, "zeros")
    LF.compute_smooth_loss([td[0].clone()], ti, [[r[0].clone()] for r in rd], ris)
    assert on_sim["smooth_fwd"] == n0 + 1
    # ... and a hit is consumed: a second compute_smooth_loss on the same frames computes
    LF.compute_photo_and_geometry_loss(ti, ris, K, td, rd, ps, pis, 1, 1, 1, 1, "zeros")
    a = LF.compute_smooth_loss(td, ti, rd, ris)
    b = LF.compute_smooth_loss(td, ti, rd, ris)
    assert on_sim["smooth_fwd"] == n0 + 2 and abs(float(a) - float(b)) <= 1e-6 * abs(float(b))


def test_single_node_step_with_the_smooth_loss_riding(on_sim):
    from scsfm_hip import config
    import loss_functions as LF

    def step(x):
        ti, ris, K, td, rd, ps, pis = x
        loss, photo, smooth, geom = LF.compute_total_loss(ti, ris, K, td, rd, ps, pis, 1, 1, 1, 1, "zeros", 1.0, 0.1, 0.5)
        loss.backward()
        return [float(v) for v in (loss, photo, smooth, geom)], [td[0].grad] + [r[0].grad for r in rd] + [p.grad for p in ps + pis]

    config.set_smooth_rides_along(False)
    v0, g0 = step(_leaves())
    n0 = on_sim["smooth_fwd"]
    config.set_smooth_rides_along(True)
    v1, g1 = step(_leaves())
    assert on_sim["smooth_fwd"] == n0  # no stand-alone smooth forward
    for a, b in zip(v1, v0):
        assert abs(a - b) <= 1e-6 * abs(b)
    for a, b in zip(g1, g0):
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())


def test_each_loss_can_be_differentiated_on_its_own(on_sim):
    """The smooth loss found waiting is the third output of the pair losses' autograd node: its gradient normally arrives
    with theirs (one backward call, the smooth term added by the combining pass).  Differentiating only ONE of the three
    must still give that loss's gradients: the others' upstream gradients arrive as None and count as zero."""
    from scsfm_hip import config
    import loss_functions as LF

    def three(x, which):
        ti, ris, K, td, rd, ps, pis = x
        photo, geom = LF.compute_photo_and_geometry_loss(ti, ris, K, td, rd, ps, pis, 1, 1, 1, 1, "zeros")
        smooth = LF.compute_smooth_loss(td, ti, rd, ris)
        {"photo": photo, "geom": geom, "smooth": smooth, "all": photo + 0.1 * smooth + 0.5 * geom}[which].backward()
        z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad.clone()
        return [z(td[0])] + [z(r[0]) for r in rd] + [z(p) for p in ps + pis]

    for which in ("smooth", "photo", "geom", "all"):
        config.set_smooth_rides_along(False)
        want = three(_leaves(), which)
        config.set_smooth_rides_along(True)
        n0 = on_sim["smooth_fwd"]
        got = three(_leaves(), which)
        assert on_sim["smooth_fwd"] == n0, which
        for a, b in zip(got, want):
            scale = float(b.abs().max())
            # (the two runs may take different routes to the same gradient: every backward leaves the upstream weights it saw
            # in the device-side hint, so the second run's forward speculates on them and its backward is the scaled add
            # of fixed-point planes where the first ran the fallback passes: 1e-6-sized differences)
            assert float((a - b).abs().max()) <= 2e-5 * scale + (0 if scale else 1e-30), which
