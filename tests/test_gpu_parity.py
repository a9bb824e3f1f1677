"""GPU parity tests proper (run with `pytest -m gpu` on an MI355X): the product path -- the drop-in
``loss_functions`` / ``inverse_warp`` modules -> autograd nodes -> ctypes -> libscsfm_hip.so -- against
(1) the golden fixtures recorded from the unmodified reference, (2) the CPU oracle on seeded inputs,
including BASELINE.json's full size, and (3) size-independent properties at full size.

Tolerances (fp32): losses 1e-5 absolute (north_star); gradients compared entry-wise at 0.2 % of the
tensor's scale with a small share of outliers allowed, because the path's gates (valid / auto mask,
clamps) are discontinuous and a 1-ulp coordinate difference flips isolated pixels.
"""
import itertools

import numpy as np
import pytest
import torch

from _util import assert_close_frac, load_inputs, load_npz, report

pytestmark = pytest.mark.gpu

# see test_hot_path_against_oracle: the reference's own fp32 pose gradients sit within ~1 % of fp64
POSE_RTOL = 1.5e-2
# ... on image-like inputs.  On iid inputs (independent depths in 0.1 .. 100 per pixel) a pose gradient row is carried by
# a handful of near pixels and ONE gate decision that rounds the other way moves it by tens of per cent: measured on
# 4 x 256 x 832 over four seeds (tools/diag_iid.py), the reference arithmetic in fp32 is off its own fp64 result by
# 1 % in the median row, by more than 5 % in 1 .. 4 of 16 rows and by up to 31 %; the HIP kernels: 1 .. 2 % median, 1 .. 3
# rows of 16 beyond 5 %, up to 72 % -- on DIFFERENT rows.  Which rows depends on the last bit of every coordinate, so
# iid cases are judged by row statistics, not by every entry.
POSE_RTOL_IID = 3e-2
# test_depth_gradients_entrywise_away_from_the_gates: HIP's WORST judged entry against the worst entry of the fp32 reference
# arithmetic in the same run.  The quantiles up to 99.99 % are held to 2x; the single worst of ~2.5 M entries is a noisier
# statistic (which near-gate entry the margins just fail to set aside) -- measured in round 4 (gpurun_out/pytest_gpu_r04d.log
# and the session after it): 0.8 .. 1.5x on image-like depth, 2.9x with border padding, 3.6 .. 6.6x on iid depth (5.8e-2
# against 8.8e-3 of the scale).  Round 3 held these to the constants 3e-3 / 1e-1; the bounds are now relative to the run.
# test_hot_path_against_oracle: share of a depth gradient's entries where HIP fp32 and the ORACLE's fp32 may differ by more than
# 0.5 % of the map's scale (pixels whose gate the two fp32 evaluations decided differently); x6 for a coarser scale's map (one
# of its entries pools up to 64 pixels' flips).  Rounds 1-5 allowed 2e-3 (x4: 8e-3) without ever recording what is measured;
# end of round 6 the test reports it (suite summary; profiles/r06_pytest_gpu.txt): 2.8e-5 .. 5.3e-5 on image-like inputs,
# 8.0e-5 on iid, coarser scales 1.2e-4 .. 7.5e-4 -- and the allowance was cut to 3e-4 (1.8e-3): ~4x / 2.4x the largest measured
HOT_PATH_BAD_SHARE = 3e-4
HOT_PATH_BAD_SHARE_COARSE_FACTOR = 6
ENTRYWISE_MAX_FACTOR = {"smooth": 4.0, "iid": 4.0, "scene": 4.0}
# Round 5: the iid bound was 10 (measured 3.6 .. 6.6).  tools/diag_gates.py traced the 20 worst judged entries of every map
# (profiles/r05_iid_worst_entries.json): all of them pixels whose VALUE gates -- sign / clamp of the depth inconsistency,
# the auto-mask comparison -- were decided by 3e-4 .. 2e-3 while the sampled depth / colour changes by tens / by 2 .. 4 per
# pixel of sampling position there, i.e. gates the margins' constant eps_val = 2e-4 did not set aside although a few ulp
# of the coordinate flip them.  The margins now grow with the local slope of what is sampled
# (oracle.pairwise_gate_margins: eps_slope_px); with them the same HIP gradients measure 1.2 / 2.0 / 1.2 x the reference
# arithmetic's own worst entry on the three maps, and 81 .. 90 % of the iid entries are judged (91 .. 96 % elsewhere).
# Round 6: FROZEN (tests/test_oracle_golden.py::test_the_gate_margins_of_the_gradient_judgement_are_frozen holds the margins
# and these constants); the iid share raised from 0.78 to what round 5 measured (0.81 .. 0.90) minus two points.
# End of round 6: the slope-aware margin NARROWED from the constant 5e-4 px to two ulp of the largest coordinate (1.2e-4 px at
# W = 832; oracle.slope_margin_px) -- tools/diag_margins.py on the hardware: every worst-entry ratio below is the same at
# 5e-4, 2.5e-4, 1.2e-4 and 6e-5 px, only a zero margin lets the flipped gates in (profiles/r06_margin_sensitivity.json) --
# and the constant value margin eps_val halved to 1e-4 on the same evidence (identical ratios down to 5e-5), so more entries are
# judged under the same bounds: measured shares 0.953 .. 0.980 (iid 0.931 / 0.964 / 0.964), minus two points
ENTRYWISE_MIN_SHARE = {"smooth": 0.93, "iid": 0.91, "scene": 0.93}
# error quantiles (median, 99 %, 99.9 %, 99.99 %) of the judged entries: HIP against the reference's fp32 arithmetic
ENTRYWISE_QUANTILE_FACTORS = (2.0, 2.0, 2.0, 2.5)
# test_iid_pose_gradients_as_row_statistics_over_seeds: HIP's row errors against the fp32 reference arithmetic's
IID_ROW_FACTOR = 2.0
FLAGS = [(1, 1, 1), (1, 1, 0), (1, 0, 1), (1, 0, 0), (0, 1, 1), (0, 1, 0), (0, 0, 1), (0, 0, 0)]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from scsfm_hip import _lib
    lib = _lib.get()
    assert lib.path.endswith("libscsfm_hip.so")  # the hipcc build, not the simulator
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def LF(dev):
    import loss_functions
    return loss_functions


@pytest.fixture(scope="module")
def IW(dev):
    import inverse_warp
    return inverse_warp


def _leaf(t, dev):
    return t.to(dev).clone().requires_grad_(True)


def _scale_close(a, b, rel=2e-3, bad=2e-3, what=""):
    b = np.asarray(b)
    assert_close_frac(a.detach().cpu().numpy(), b, atol=rel * float(np.abs(b).max()) + 1e-12, rtol=1e-3,
                      max_bad_frac=bad, what=what)


# ------------------------------------------------------------------------------------------------
# 1. golden fixtures of the reference
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["smooth", "iid", "tiny"])
def test_pairwise_loss_goldens(LF, dev, name):
    d = load_inputs(name)
    gold = load_npz(f"pair_{name}.npz")
    tgt, ref, K = d["tgt_img"].to(dev), d["ref_imgs"][0].to(dev), d["intrinsics"].to(dev)
    for (ssim, mask, auto), pad in itertools.product(FLAGS, ("zeros", "border")):
        key = f"{ssim}{mask}{auto}_{pad}"
        td, rd, po = _leaf(d["tgt_depth"][0], dev), _leaf(d["ref_depths"][0][0], dev), _leaf(d["poses"][0], dev)
        photo, geom = LF.compute_pairwise_loss(tgt, ref, td, rd, po, K, ssim, mask, auto, pad)
        assert abs(float(photo) - float(gold[f"{key}/photo"])) <= 1e-5, key
        assert abs(float(geom) - float(gold[f"{key}/geom"])) <= 1e-5, key
        (1.0 * photo + 0.5 * geom).backward()
        _scale_close(po.grad, gold[f"{key}/g_pose"], rel=POSE_RTOL, what=key + " g_pose", bad=0.0) if name != "tiny" else None
        for nm, t in (("g_tgt_depth", td), ("g_ref_depth", rd)):
            st = gold[f"{key}/{nm}_stats"]
            assert abs(float(t.grad.double().abs().sum()) - st[1]) <= 2e-3 * st[1] + 1e-9, (key, nm)
            if f"{key}/{nm}" in gold:
                _scale_close(t.grad, gold[f"{key}/{nm}"], what=f"{key} {nm}")


@pytest.mark.parametrize("name", ["smooth", "iid"])
def test_total_loss_goldens(LF, dev, name):
    d = load_inputs(name)
    gold = load_npz(f"total_{name}.npz")
    tgt, refs, K = d["tgt_img"].to(dev), [r.to(dev) for r in d["ref_imgs"]], d["intrinsics"].to(dev)
    for n_scales in (1, 2):
        for ssim, mask, auto, pad in ((1, 1, 1, "zeros"), (1, 1, 0, "border")):
            key = f"s{n_scales}_{ssim}{mask}{auto}_{pad}"
            td = [_leaf(x, dev) for x in d["tgt_depth"]]
            rd = [[_leaf(x, dev) for x in r] for r in d["ref_depths"]]
            ps, pi = [_leaf(p, dev) for p in d["poses"]], [_leaf(p, dev) for p in d["poses_inv"]]
            photo, geom = LF.compute_photo_and_geometry_loss(tgt, refs, K, td, rd, ps, pi, n_scales, ssim, mask, auto, pad)
            smooth = LF.compute_smooth_loss(td, tgt, rd, refs)
            for nm, v in (("photo", photo), ("geom", geom), ("smooth", smooth)):
                assert abs(float(v) - float(gold[f"{key}/{nm}"])) <= 1e-5, (key, nm)
            (1.0 * photo + 0.1 * smooth + 0.5 * geom).backward()
            for s in range(n_scales):
                _scale_close(td[s].grad, gold[f"{key}/g_tgt_depth_s{s}"], what=f"{key} tgt s{s}")
                for i in range(2):
                    _scale_close(rd[i][s].grad, gold[f"{key}/g_ref{i}_depth_s{s}"], what=f"{key} ref{i} s{s}")
            prt = POSE_RTOL if name == "smooth" else POSE_RTOL_IID
            for i in range(2):
                _scale_close(ps[i].grad, gold[f"{key}/g_pose{i}"], rel=prt, bad=0.0)
                _scale_close(pi[i].grad, gold[f"{key}/g_pose_inv{i}"], rel=prt, bad=0.0)


@pytest.mark.parametrize("name", ["smooth", "iid"])
def test_inverse_warp2_goldens(IW, dev, name):
    d = load_inputs(name)
    gold = load_npz(f"maps_{name}.npz")
    for pad in ("zeros", "border"):
        w, v, pd, cd = IW.inverse_warp2(d["ref_imgs"][0].to(dev), d["tgt_depth"][0].to(dev),
                                        d["ref_depths"][0][0].to(dev), d["poses"][0].to(dev),
                                        d["intrinsics"].to(dev), pad)
        assert (v.cpu().numpy().astype(np.uint8) != gold[f"{pad}/valid_mask"]).mean() <= 1e-3
        assert_close_frac(w.cpu().numpy(), gold[f"{pad}/projected_img"], atol=3e-4, max_bad_frac=1e-3)
        pd_atol = 1e-5 if name == "smooth" else 1e-4 * float(np.abs(gold[f"{pad}/projected_depth"]).max())
        assert_close_frac(pd.cpu().numpy(), gold[f"{pad}/projected_depth"], atol=pd_atol, rtol=2e-5, max_bad_frac=1e-3)
        assert_close_frac(cd.cpu().numpy(), gold[f"{pad}/computed_depth"], atol=1e-5, rtol=1e-5)


def test_pose_and_errors_goldens(LF, IW, dev):
    gold = load_npz("misc.npz")
    r = torch.from_numpy(gold["pose/probe"]).to(dev)
    for mode in ("euler", "quat"):
        v = _leaf(torch.from_numpy(gold["pose/vec"]), dev)
        M = IW.pose_vec2mat(v, mode)
        (M * r).sum().backward()
        np.testing.assert_allclose(M.detach().cpu().numpy(), gold[f"pose/{mode}/mat"], atol=1e-6)
        np.testing.assert_allclose(v.grad.cpu().numpy(), gold[f"pose/{mode}/g_vec"], atol=3e-6)
    for ds in ("kitti", "nyu"):
        out = LF.compute_errors(torch.from_numpy(gold[f"errors/{ds}/gt"]).to(dev),
                                torch.from_numpy(gold[f"errors/{ds}/pred"]).to(dev), ds)
        np.testing.assert_allclose(out, gold[f"errors/{ds}/out"], rtol=1e-5, atol=1e-6)  # AbsRel is out[1]


# ------------------------------------------------------------------------------------------------
# 2. the CPU oracle on seeded inputs, up to BASELINE.json's full size
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,n_ref,dataset,depth,auto,pad,scales", [
    (2, 128, 416, 2, "kitti", "smooth", 0, "zeros", 1),
    (2, 128, 416, 2, "kitti", "smooth", 1, "zeros", 1),
    (3, 100, 210, 1, "kitti", "iid", 1, "zeros", 1),        # ragged: partial tiles
    (12, 256, 832, 2, "kitti", "smooth", 1, "zeros", 1),    # configs[1] / [2] per-GPU workload
    (4, 256, 320, 4, "nyu", "smooth", 1, "zeros", 1),       # configs[4]: NYU intrinsics, 4 refs
    (16, 256, 320, 4, "nyu", "smooth", 1, "zeros", 1),      # ... at the batch SURVEY 8 fixes for it (scripts/train_nyu.sh:7)
    (8, 256, 832, 2, "kitti", "smooth", 1, "zeros", 1),     # configs[3]: batch 8 per GPU
    (4, 256, 832, 2, "kitti", "iid", 1, "zeros", 1),        # full size, incoherent gathers / scatter
    (4, 256, 832, 2, "kitti", "scene", 1, "zeros", 1),      # full size, piecewise-smooth depth with occlusion edges (round 5)
    (4, 256, 832, 2, "kitti", "smooth", 1, "border", 1),    # full size, border padding
    (4, 256, 832, 1, "kitti", "smooth", 1, "zeros", 2),     # full size, two scales (--num-scales 2)
    (2, 128, 416, 2, "kitti", "smooth", 1, "zeros", 4),     # four scales read in place (depth_shift 0..3)
    (2, 256, 832, 1, "kitti", "smooth", 1, "border", 4),    # full size, four scales, border padding
])
def test_hot_path_against_oracle(LF, dev, B, H, W, n_ref, dataset, depth, auto, pad, scales):
    """Losses to 1e-5 per pair term.  Gradients: the HIP fp32 result must be as close to the fp64
    oracle as the fp32 oracle (= the reference's own arithmetic) is, up to a factor, plus 0.5 % of the
    tensor's scale.  The factor matters for the pose gradients: with depths from 0.1 to 100 a handful
    of near pixels carry most of d/d translation, so a single auto-mask decision that rounds the
    other way (1-ulp difference) moves them by several per cent in ANY fp32
    implementation, the reference included."""
    from oracle import scsfm_oracle as O
    from scsfm_hip import synth
    d = synth.make_batch(B, H, W, n_ref=n_ref, seed=17, depth=depth, image=synth.image_law(depth),
                         dataset=dataset, num_scales=scales)
    flags = (1, 1, auto, pad)

    def run(device, fn_pg, fn_s, dtype=torch.float32):
        mv = lambda t: t.to(device=device, dtype=dtype).clone().requires_grad_(True)
        cv = lambda t: t.to(device=device, dtype=dtype)
        td = [mv(t) for t in d["tgt_depth"]]
        rd = [[mv(t) for t in r] for r in d["ref_depths"]]
        ps, pi = [mv(p) for p in d["poses"]], [mv(p) for p in d["poses_inv"]]
        tgt, refs, K = cv(d["tgt_img"]), [cv(r) for r in d["ref_imgs"]], cv(d["intrinsics"])
        photo, geom = fn_pg(tgt, refs, K, td, rd, ps, pi, scales, *flags)
        smooth = fn_s(td, tgt, rd, refs)
        (photo + 0.1 * smooth + 0.5 * geom).backward()
        grads = [td[0].grad] + [r[0].grad for r in rd] + [p.grad for p in ps + pi]
        grads += [t.grad for t in td[1:]] + [t.grad for r in rd for t in r[1:]]  # the coarser scales' maps
        return [float(photo.detach()), float(geom.detach()), float(smooth.detach())], [g.detach().cpu().double() for g in grads]

    vh, gh = run(dev, LF.compute_photo_and_geometry_loss, LF.compute_smooth_loss)
    vo, go = run("cpu", O.photo_and_geometry_loss, O.smooth_loss)
    v64, g64 = run("cpu", O.photo_and_geometry_loss, O.smooth_loss, torch.float64)
    for a, b, nm in zip(vh, vo, ("photo", "geom", "smooth")):
        assert abs(a - b) <= 1e-5, (nm, a, b)  # north_star's bar, flat (photo / geom are sums over 2 * n_ref * scales terms)
    if depth == "iid":  # pose gradients of iid inputs: row statistics (see POSE_RTOL_IID above)
        rel = lambda x, c: (x - c).abs().max(dim=1).values / c.abs().max(dim=1).values
        rows_h = torch.cat([rel(gh[i], g64[i]) for i in range(n_ref + 1, 3 * n_ref + 1)])
        rows_o = torch.cat([rel(go[i], g64[i]) for i in range(n_ref + 1, 3 * n_ref + 1)])
        # (one seed, a handful of rows: the median only; the distribution over four seeds -- median, 90 % quantile and
        # maximum against the reference's own fp32 arithmetic -- is test_iid_pose_gradients_as_row_statistics_over_seeds)
        assert float(rows_h.median()) <= IID_ROW_FACTOR * float(rows_o.median()) + POSE_RTOL, (rows_h, rows_o)
    worst_bad = [0.0, 0.0]  # (full-resolution maps, coarser scales' maps)
    pose_need, pose_hip, pose_ref = -1.0, 0.0, 0.0
    for i, (a, b, c) in enumerate(zip(gh, go, g64)):
        scale = float(c.abs().max())
        if i <= n_ref or i > 3 * n_ref:   # depth maps: entry-wise with a small share of outliers (flipped pixels)
            assert a.shape == b.shape
            bad = ((a - b).abs() > 5e-3 * scale).double().mean().item()
            worst_bad[0 if i <= n_ref else 1] = max(worst_bad[0 if i <= n_ref else 1], bad)
            assert bad <= HOT_PATH_BAD_SHARE * (1 if i <= n_ref else HOT_PATH_BAD_SHARE_COARSE_FACTOR), (i, bad)
        elif depth == "iid":
            continue
        else:            # poses: every entry, noise-aware
            ref_noise = (b - c).abs()
            entrywise = bool(((a - c).abs() <= POSE_RTOL * scale + 4 * ref_noise).all())
            # On incoherent (iid) inputs at full size the fp32 reference arithmetic itself is off by up to a quarter
            # of the scale on single entries (measured: 0.093 of 0.389, tools/diag_iid.py) -- which entries depends
            # on which near pixels' gates round the other way, so two fp32 implementations are off on DIFFERENT
            # entries: also accept being no further from fp64 than 1.5 x the reference's own worst entry
            tensorwise = float((a - c).abs().max()) <= POSE_RTOL * scale + 1.5 * float(ref_noise.max())
            assert entrywise or tensorwise, (i, float((a - c).abs().max()), float(ref_noise.max()), scale)
            # what the entry-wise bound needs of its constant term: max over the entries of (|hip - fp64| - 4 |fp32 ref - fp64|) / scale
            pose_need = max(pose_need, float((((a - c).abs() - 4 * ref_noise) / scale).max()))
            pose_hip = max(pose_hip, float((a - c).abs().max()) / scale)
            pose_ref = max(pose_ref, float(ref_noise.max()) / scale)
    if depth != "iid":
        report(f"hot path vs oracle [{dataset} {B}x{H}x{W} refs {n_ref} {depth} auto {auto} {pad} scales {scales}]: pose gradients, worst entry / tensor scale: "
               f"hip {pose_hip:.2e}, reference fp32 arithmetic {pose_ref:.2e}; constant the noise-aware bound needs: {max(pose_need, 0.0):.2e} (allowed {POSE_RTOL:.1e})")
    report(f"hot path vs oracle [{dataset} {B}x{H}x{W} refs {n_ref} {depth} auto {auto} {pad} scales {scales}]: share of depth-gradient entries "
           f"where HIP fp32 and the oracle's fp32 differ by > 0.5 % of scale: {worst_bad[0]:.2e}" + (f" (coarser scales: {worst_bad[1]:.2e})" if scales > 1 else ""))


def test_forward_only_validation_path_against_oracle(LF, dev):
    """validate_without_gt (train.py:302-362): torch.no_grad, auto-mask off -- the plain forward kernel (32 B/px),
    no speculative backward -- at full size against the oracle."""
    from oracle import scsfm_oracle as O
    from scsfm_hip import synth
    d = synth.make_batch(4, 256, 832, n_ref=2, seed=23, depth="smooth", image="smooth", dataset="kitti")
    to = lambda t: t.to(dev)
    with torch.no_grad():
        photo, geom = LF.compute_photo_and_geometry_loss(to(d["tgt_img"]), [to(r) for r in d["ref_imgs"]], to(d["intrinsics"]),
                                                         [to(t) for t in d["tgt_depth"]],
                                                         [[to(t) for t in r] for r in d["ref_depths"]],
                                                         [to(p) for p in d["poses"]], [to(p) for p in d["poses_inv"]],
                                                         1, 1, 1, 0, "zeros")
        smooth = LF.compute_smooth_loss([to(t) for t in d["tgt_depth"]], to(d["tgt_img"]),
                                        [[to(t) for t in r] for r in d["ref_depths"]], [to(r) for r in d["ref_imgs"]])
        po, go = O.photo_and_geometry_loss(d["tgt_img"], d["ref_imgs"], d["intrinsics"], d["tgt_depth"], d["ref_depths"],
                                           d["poses"], d["poses_inv"], 1, 1, 1, 0, "zeros")
        so = O.smooth_loss(d["tgt_depth"], d["tgt_img"], d["ref_depths"], d["ref_imgs"])
    assert photo.grad_fn is None
    assert abs(float(photo) - float(po)) <= 1e-5 and abs(float(geom) - float(go)) <= 1e-5 and abs(float(smooth) - float(so)) <= 1e-5


# ------------------------------------------------------------------------------------------------
# 2b. the fp64 instantiations of the SAME kernel sources on the hardware, against the fp64 oracle at 1e-9: what
#     the CPU simulation checks for the logic (tiles, halos, wave shuffles, LDS parking, window atomics, last-block
#     reductions) is checked here with real concurrency
# ------------------------------------------------------------------------------------------------
FLAGS8 = [(1, 1, 1), (1, 1, 0), (1, 0, 1), (1, 0, 0), (0, 1, 1), (0, 1, 0), (0, 0, 1), (0, 0, 0)]


def _rel64(a, b):
    return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-300))


@pytest.mark.parametrize("pad", ["zeros", "border"])
@pytest.mark.parametrize("flags3", FLAGS8)
def test_fp64_pair_kernels_on_hardware(dev, flags3, pad):
    """scsfm_pair_fwd_f64 + the two-pass backward (pair_bwd_photo / pair_bwd_geom), every flag combination."""
    from oracle import scsfm_oracle as O
    from scsfm_hip import _lib, capi, synth
    lib = _lib.get()
    d = synth.make_batch(2, 72, 100, n_ref=1, seed=21, depth="smooth")  # ragged: partial tiles on both axes
    c = lambda x: x.double().contiguous()
    host = [c(d["tgt_img"]), c(d["ref_imgs"][0]), c(d["tgt_depth"][0]), c(d["ref_depths"][0][0]), c(d["poses"][0]),
            c(d["intrinsics"])]
    ti, ri, td, rd, po, K = host
    ssim, mask, auto = flags3
    fl = capi.make_flags(ssim, mask, auto, pad)
    out, ws = capi.pair_fwd(lib, *[t.to(dev) for t in host], fl)
    tdl, rdl, pol = (t.clone().requires_grad_(True) for t in (td, rd, po))
    diff_img, diff_depth, m = O.pairwise_maps(ti, ri, tdl, rdl, pol, K, ssim, mask, auto, pad)
    assert abs(float(out[4]) - float(m.sum())) == 0
    assert abs(float(out[2]) - float((diff_img * m).sum().detach())) < 1e-9
    assert abs(float(out[3]) - float((diff_depth * m).sum().detach())) < 1e-9
    Sm = m.sum()
    gate_g = 1.0 if float(Sm) > 10000 else 0.0
    (0.7 * (diff_img * m).sum() / (3 * Sm) + gate_g * 1.3 * (diff_depth * m).sum() / Sm).backward()
    t = lambda v: torch.tensor([v], dtype=torch.float64, device=dev)
    gt, gr, gp = capi.pair_bwd(lib, *[x.to(dev) for x in host], fl, ws, t(0.7), t(1.3))
    assert _rel64(gt, tdl.grad) < 1e-9 and _rel64(gp, pol.grad) < 1e-9
    if rdl.grad is not None and float(rdl.grad.abs().max()) > 0:
        assert _rel64(gr, rdl.grad) < 1e-9


@pytest.mark.parametrize("H,W,B", [(72, 100, 2), (15, 63, 40), (33, 129, 8), (5, 200, 40)])
@pytest.mark.parametrize("hint,upstream", [((0.7, 1.3), (0.7, 1.3)),   # speculation holds
                                           ((0.7, 1.3), (1.0, 0.5)),   # wrong hint: device-side fallback
                                           (None, (0.7, 1.3))])        # plain forward
def test_fp64_speculative_forward_on_hardware(dev, H, W, B, hint, upstream):
    """scsfm_pairs_fwd_f64 / scsfm_pairs_bwd_f64: the fused speculative forward (+ combine), the fallback passes
    behind their guards and the plain path, at sizes that are no multiple of any tile."""
    from oracle import scsfm_oracle as O
    from scsfm_hip import _lib, capi, synth
    lib = _lib.get()
    d = synth.make_batch(B, H, W, n_ref=2, seed=H * 100 + W, depth="smooth")
    c = lambda x: x.double().contiguous()
    ti, K = c(d["tgt_img"]), c(d["intrinsics"])
    ris = [c(r) for r in d["ref_imgs"]]
    tds, rds = [c(d["tgt_depth"][0])], [[c(r[0])] for r in d["ref_depths"]]
    ps, pis = [c(p) for p in d["poses"]], [c(p) for p in d["poses_inv"]]
    lf = lambda x: x.clone().requires_grad_(True)
    td, rd = [lf(t) for t in tds], [[lf(t) for t in r] for r in rds]
    pp, pi = [lf(p) for p in ps], [lf(p) for p in pis]
    po, go = O.photo_and_geometry_loss(ti, ris, K, td, rd, pp, pi, 1, 1, 1, 1, "zeros")
    (upstream[0] * po + upstream[1] * go).backward()
    g = lambda x: x.to(dev)
    fl = capi.make_flags(1, 1, 1, "zeros")
    dv = dict(ti=g(ti), K=g(K), ris=[g(r) for r in ris], tds=[g(t) for t in tds], rds=[[g(t) for t in r] for r in rds],
              ps=[g(p) for p in ps], pis=[g(p) for p in pis])
    photo, geom, _, ws = capi.photo_geometry_fwd(lib, fl, dv["ti"], dv["K"], dv["ris"], dv["tds"], dv["rds"], dv["ps"],
                                                 dv["pis"], hint=hint)
    assert abs(float(photo) - float(po)) < 1e-11 and abs(float(geom) - float(go)) < 1e-11
    t = lambda v: torch.tensor([v], dtype=torch.float64, device=dev)
    g_td, g_rd, g_p, g_pi = capi.photo_geometry_bwd(lib, fl, dv["ti"], dv["K"], dv["ris"], dv["tds"], dv["rds"], dv["ps"],
                                                    dv["pis"], ws, t(upstream[0]), t(upstream[1]))
    z = lambda x: x.grad if x.grad is not None else torch.zeros_like(x)
    assert _rel64(g_td[0], z(td[0])) < 1e-9
    for i in range(2):
        assert _rel64(g_rd[i][0], z(rd[i][0])) < 1e-9
        assert _rel64(g_p[i], z(pp[i])) < 1e-9 and _rel64(g_pi[i], z(pi[i])) < 1e-9


@pytest.mark.parametrize("hint,upstream", [((1.0, 0.5), (1.0, 0.5)), ((1.0, 0.5), (0.3, 1.1)), (None, (1.0, 0.5))])
def test_fp64_four_scales_read_in_place_on_hardware(dev, hint, upstream):
    """scsfm_pair_desc::depth_shift on the hardware: maps of 4 scales ([B,1,H>>s,W>>s]) read through the nearest
    up-sampling's index map (loss_functions.py:77-82), gradients sum-pooled by the combining kernel; against the
    oracle (F.interpolate under autograd) in fp64."""
    from oracle import scsfm_oracle as O
    from scsfm_hip import _lib, capi, synth
    lib = _lib.get()
    B, H, W = 2, 80, 104
    d = synth.make_batch(B, H, W, n_ref=2, seed=57, depth="smooth", num_scales=4)
    c = lambda x: x.double().contiguous()
    ti, K = c(d["tgt_img"]), c(d["intrinsics"])
    ris = [c(r) for r in d["ref_imgs"]]
    tds, rds = [c(t) for t in d["tgt_depth"]], [[c(t) for t in r] for r in d["ref_depths"]]
    ps, pis = [c(p) for p in d["poses"]], [c(p) for p in d["poses_inv"]]
    lf = lambda x: x.clone().requires_grad_(True)
    td, rd = [lf(t) for t in tds], [[lf(t) for t in r] for r in rds]
    pp, pi = [lf(p) for p in ps], [lf(p) for p in pis]
    po, go = O.photo_and_geometry_loss(ti, ris, K, td, rd, pp, pi, 4, 1, 1, 1, "zeros")
    (upstream[0] * po + upstream[1] * go).backward()
    g = lambda x: x.to(dev)
    fl = capi.make_flags(1, 1, 1, "zeros")
    a = (g(ti), g(K), [g(r) for r in ris], [g(t) for t in tds], [[g(t) for t in r] for r in rds], [g(p) for p in ps],
         [g(p) for p in pis])
    photo, geom, outs, ws = capi.photo_geometry_fwd(lib, fl, *a, hint=hint)
    assert outs.shape[0] == 16
    assert abs(float(photo) - float(po)) < 1e-11 and abs(float(geom) - float(go)) < 1e-11
    t = lambda v: torch.tensor([v], dtype=torch.float64, device=dev)
    g_td, g_rd, g_p, g_pi = capi.photo_geometry_bwd(lib, fl, *a, ws, t(upstream[0]), t(upstream[1]))
    for s in range(4):
        assert g_td[s].shape == tds[s].shape
        assert _rel64(g_td[s], td[s].grad) < 1e-9, s
        for i in range(2):
            assert _rel64(g_rd[i][s], rd[i][s].grad) < 1e-9, (i, s)
    for i in range(2):
        assert _rel64(g_p[i], pp[i].grad) < 1e-9 and _rel64(g_pi[i], pi[i].grad) < 1e-9


def test_large_gradients_bypass_the_fixed_point_window(dev):
    """Twin of tests/test_hostsim_kernels.py's test of the same name on the hardware (ds_add_u32 cells of +-2048 range:
    unscaled per-pixel terms of ~7000 must take the direct fp32 atomics)."""
    from oracle import scsfm_oracle as O
    from scsfm_hip import _lib, capi, synth
    lib = _lib.get()
    B, H, W = 2, 72, 100
    d = synth.make_batch(B, H, W, n_ref=1, seed=61, depth="smooth")
    g = torch.Generator().manual_seed(3)
    tds = [1.2e-3 * (1 + 0.05 * torch.rand(B, 1, H, W, generator=g))]
    rds = [[1.0e-3 * (1.02 + 0.05 * torch.rand(B, 1, H, W, generator=g))]]
    p = torch.zeros(B, 6)
    p[:, 0], p[:, 1] = 4e-7, -3e-7
    ti, K, ris = d["tgt_img"], d["intrinsics"], d["ref_imgs"]
    c = lambda x: x.double()
    lf = lambda x: x.double().clone().requires_grad_(True)
    td64, rd64 = [lf(tds[0])], [[lf(rds[0][0])]]
    po, go = O.photo_and_geometry_loss(c(ti), [c(ris[0])], c(K), td64, rd64, [c(p)], [c(-p)], 1, 1, 1, 0, "zeros")
    (po + 5.0 * go).backward()
    fl = capi.make_flags(1, 1, 0, "zeros")
    v = lambda x: x.to(dev).contiguous()
    a = (v(ti), v(K), [v(ris[0])], [v(tds[0])], [[v(rds[0][0])]], [v(p)], [v(-p)])
    photo, geom, outs, ws = capi.photo_geometry_fwd(lib, fl, *a, hint=(1.0, 5.0))
    assert abs(float(photo) - float(po)) < 1e-5 and abs(float(geom) - float(go)) < 1e-5
    t = lambda x: torch.tensor([x], device=dev)
    g_td, g_rd, _, _ = capi.photo_geometry_bwd(lib, fl, *a, ws, t(1.0), t(5.0))
    assert float(rd64[0][0].grad.abs().max()) * 3 * float(outs[0, 4]) > 2048
    assert _rel64(g_rd[0][0].double(), rd64[0][0].grad) < 1e-3
    assert _rel64(g_td[0].double(), td64[0].grad) < 1e-3


@pytest.mark.parametrize("B,scale,tz,w_geom,hint", [
    (2, 1.0, 1.0, 0.5, (1.0, 0.5)),    # the round-4 review's case: tz = +1, depth 0.1 .. 0.3, 2 x 72 x 100
    (3, 1.0, 1.0, 0.5, (1.0, 0.5)),
    (3, 0.03, 2.0, 0.0, (1.0, 0.0)),   # wrapped 66 cells of the speculative tail before round 5's guard (CPU simulation)
    (2, 0.04, 2.0, 0.5, (1.0, 0.5)),   # geometry gate closed -> fallback passes; wrapped 8 cells of the geometry pass
    (3, 0.04, 2.0, 0.0, None),         # no speculation: fallback passes
])
def test_compressive_warps_do_not_wrap_the_fixed_point_window(dev, B, scale, tz, w_geom, hint):
    """Twin of tests/test_hostsim_kernels.py's test of the same name on the hardware: a scaled-down scene after a
    forward motion of several depths (areal compression ~100, near-cap scatter terms) used to wrap the +-2048-unit
    ds_add_u32 cells silently.  Debug launches count wraps exactly (returning LDS atomics): zero, and dL/d ref_depth
    matches the fp64 oracle."""
    from oracle import scsfm_oracle as O
    from scsfm_hip import _lib, capi, synth
    lib = _lib.get()
    H, W = 72, 100
    d = synth.make_batch(B, H, W, n_ref=1, seed=67, depth="smooth")
    g = torch.Generator().manual_seed(5)
    mk = lambda: (scale * (0.1 + 0.2 * torch.rand(B, 1, H, W, generator=g))).contiguous()
    tds, rds = [mk()], [[mk()]]
    p = torch.zeros(B, 6)
    p[:, 2] = tz * scale
    ti, K, ris = d["tgt_img"], d["intrinsics"], d["ref_imgs"]
    c = lambda x: x.double()
    lf = lambda x: x.double().clone().requires_grad_(True)
    td64, rd64 = [lf(tds[0])], [[lf(rds[0][0])]]
    po, go = O.photo_and_geometry_loss(c(ti), [c(ris[0])], c(K), td64, rd64, [c(p)], [c(-p)], 1, 1, 1, 1, "zeros")
    (po + w_geom * go).backward()
    fl = capi.make_flags(1, 1, 1, "zeros")
    v = lambda x: x.to(dev).contiguous()
    a = (v(ti), v(K), [v(ris[0])], [v(tds[0])], [[v(rds[0][0])]], [v(p)], [v(-p)])
    t = lambda x: torch.tensor([x], device=dev)
    for check in (True, False):  # the debug launch (runtime-flag instantiation, returning atomics) and the product launch
        photo, geom, outs, ws = capi.photo_geometry_fwd(lib, fl, *a, hint=hint, check_window=check and hint is not None)
        assert abs(float(photo) - float(po)) < 1e-5 and abs(float(geom) - float(go)) < 1e-5
        g_td, g_rd, _, _ = capi.photo_geometry_bwd(lib, fl, *a, ws, t(1.0), t(w_geom), check_window=check)
        assert capi.window_overflows(lib, ws, 2, B, H, W, spec=hint is not None) == [0, 0]
        assert _rel64(g_rd[0][0].double(), rd64[0][0].grad) < 1e-4, check
        assert _rel64(g_td[0].double(), td64[0].grad) < 1e-4, check


@pytest.mark.parametrize("auto", [0, 1])
def test_nonuniform_compression_inside_a_large_footprint(dev, auto):
    """Twin of tests/test_hostsim_kernels.py::test_nonuniform_compression_inside_a_large_footprint_fp32 on the hardware
    (round-5 review item 4): every other column of a tile very near (depth << the forward motion: those pixels collapse
    onto a few texels, unscaled scatter terms ~25 each), the rest far (spread over hundreds of cells) -- a compression the
    bounding-box heuristic of round 5 could not see; photo-only upstream gradient.  The unit of the tile's fixed-point cells
    follows from the per-tile bound (csrc/scsfm_geom.h: win_units_of): no wrap in the debug launch (exact count) nor in
    the product launch, dL/d ref_depth within 1e-4 of the fp64 oracle."""
    from _util import nonuniform_case as _nonuniform_case
    from oracle import scsfm_oracle as O
    from scsfm_hip import _lib, capi
    lib = _lib.get()
    ti, K, ris, tds, rds, ps, pis = _nonuniform_case()
    c = lambda x: x.double()
    lf = lambda x: x.double().clone().requires_grad_(True)
    td64, rd64 = [lf(tds[0])], [[lf(rds[0][0])]]
    po, go = O.photo_and_geometry_loss(c(ti), [c(ris[0])], c(K), td64, rd64, [c(ps[0])], [c(pis[0])], 1, 1, 1, auto, "zeros")
    po.backward()
    fl = capi.make_flags(1, 1, auto, "zeros")
    v = lambda x: x.to(dev).contiguous()
    a = (v(ti), v(K), [v(ris[0])], [v(tds[0])], [[v(rds[0][0])]], [v(ps[0])], [v(pis[0])])
    t = lambda x: torch.tensor([x], device=dev)
    for check in (True, False):
        photo, geom, outs, ws = capi.photo_geometry_fwd(lib, fl, *a, hint=(1.0, 0.0), check_window=check)
        assert abs(float(photo) - float(po)) < 1e-5 and abs(float(geom) - float(go)) < 1e-5
        g_td, g_rd, _, _ = capi.photo_geometry_bwd(lib, fl, *a, ws, t(1.0), t(0.0), check_window=check)
        assert capi.window_overflows(lib, ws, 2, 2, 72, 100) == [0, 0]
        assert float(rd64[0][0].grad.abs().max()) * 3 * float(outs[0, 4]) > 2048  # beyond a 2^-20 cell's range on the busiest texel
        assert _rel64(g_rd[0][0].double(), rd64[0][0].grad) < 1e-4, check
        assert _rel64(g_td[0].double(), td64[0].grad) < 1e-4, check


@pytest.mark.parametrize("depth", ["smooth", "iid", "scene"])
def test_no_window_cell_wraps_on_the_bench_inputs(LF, dev, depth, monkeypatch):
    """The exact wrap detector (SCSFM_CHECK_WINDOW=1: returning LDS atomics in the speculative forward and in the fallback
    geometry pass) on the bench's own batches at configs[1] size, through the autograd nodes: no fixed-point cell of any
    of the 12,768 tiles wraps -- with the hinted weights (speculation holds) and with other weights (fallback passes) --,
    and the debug launch's losses equal the product launch's."""
    from scsfm_hip import capi, synth
    d = synth.make_batch(12, 256, 832, n_ref=2, seed=0, depth=depth, image=synth.image_law(depth), dataset="kitti")
    to = lambda t: t.to(dev)
    tgt, K, refs = to(d["tgt_img"]), to(d["intrinsics"]), [to(t) for t in d["ref_imgs"]]

    def step(w_geom):
        mv = lambda t: t.to(dev).clone().requires_grad_(True)
        td, rd = [mv(d["tgt_depth"][0])], [[mv(r[0])] for r in d["ref_depths"]]
        ps, pi = [mv(p) for p in d["poses"]], [mv(p) for p in d["poses_inv"]]
        photo, geom = LF.compute_photo_and_geometry_loss(tgt, refs, K, td, rd, ps, pi, 1, 1, 1, 1, "zeros")
        (photo + w_geom * geom).backward()  # (raises capi.WindowOverflow under SCSFM_CHECK_WINDOW=1 if a cell wrapped)
        return float(photo.detach()), float(geom.detach()), td[0].grad.clone()

    plain = step(0.5)
    monkeypatch.setenv("SCSFM_CHECK_WINDOW", "1")
    checked = step(0.5)     # speculation holds: the forward's tail is what scatters
    step(0.3)               # other weights: the fallback passes scatter
    monkeypatch.delenv("SCSFM_CHECK_WINDOW")
    assert plain[0] == checked[0] and plain[1] == checked[1]
    scale = float(plain[2].abs().max())
    assert float((plain[2] - checked[2]).abs().max()) <= 1e-5 * scale  # (direct atomics of a tile add in another order)


def test_boundary_functions(IW, dev):
    """pixel2cam / cam2pixel / cam2pixel2 / legacy inverse_warp (euler and quat) on the hardware, fp64 and fp32,
    values and gradients against the oracle (tests/_boundary_checks.py; CPU twin: tests/test_boundary_names.py)."""
    import _boundary_checks as BC
    BC.run(IW, dev, torch.float64, 1e-12)
    BC.run(IW, dev, torch.float32, 2e-6)


# ------------------------------------------------------------------------------------------------
# 3. size-independent properties at full size
# ------------------------------------------------------------------------------------------------
def _full(dev, seed=3, B=12):
    from scsfm_hip import synth
    d = synth.make_batch(B, 256, 832, n_ref=1, seed=seed, depth="smooth")
    return [t.to(dev).contiguous() for t in (d["tgt_img"], d["ref_imgs"][0], d["tgt_depth"][0],
                                             d["ref_depths"][0][0], d["poses"][0], d["intrinsics"])]


def test_scales_read_in_place_equal_materialised_upsampling(LF, dev):
    """configs[1] size, 4 scales: handing the coarser maps to the kernels as they are must give the SAME losses, bit
    for bit, as up-sampling them with F.interpolate first (the reads return the same values in the same order), and
    gradients equal to autograd's pooled ones up to the summation order."""
    import torch.nn.functional as F
    from scsfm_hip import synth
    B, H, W = 12, 256, 832
    d = synth.make_batch(B, H, W, n_ref=2, seed=23, depth="smooth", num_scales=4)
    cv = lambda t: t.to(dev).contiguous()
    tgt, refs, K = cv(d["tgt_img"]), [cv(r) for r in d["ref_imgs"]], cv(d["intrinsics"])
    ps, pi = [cv(p) for p in d["poses"]], [cv(p) for p in d["poses_inv"]]

    def run(materialise):
        td = [_leaf(t, dev) for t in d["tgt_depth"]]
        rd = [[_leaf(t, dev) for t in r] for r in d["ref_depths"]]
        pp, pq = [_leaf(p, dev) for p in ps], [_leaf(p, dev) for p in pi]
        up = (lambda t: F.interpolate(t, (H, W), mode="nearest") if t.shape[-1] != W else t) if materialise else (lambda t: t)
        photo, geom = LF.compute_photo_and_geometry_loss(tgt, refs, K, [up(t) for t in td], [[up(t) for t in r] for r in rd],
                                                         pp, pq, 4, 1, 1, 1, "zeros")
        (photo + 0.5 * geom).backward()
        return float(photo.detach()), float(geom.detach()), [t.grad for t in td] + [t.grad for r in rd for t in r], \
            [p.grad for p in pp + pq]

    pa, ga, da, qa = run(False)
    pb, gb, db, qb = run(True)
    assert pa == pb and ga == gb
    for x, y in zip(da, db):
        assert x.shape == y.shape
        assert float((x - y).abs().max()) <= 2e-5 * float(y.abs().max())
    for x, y in zip(qa, qb):
        assert float((x - y).abs().max()) <= 1e-6 * float(y.abs().max())


def test_sums_are_additive_over_batch_shards(dev):
    """Checksum of checksums: the three raw sums of a batch equal the sums over its two halves (this is
    also what the exact data-parallel mode relies on)."""
    from scsfm_hip import _lib, capi
    lib = _lib.get()
    a = _full(dev)
    fl = capi.make_flags(1, 1, 1, "zeros")
    whole, _ = capi.pair_fwd(lib, *a, fl)
    lo, _ = capi.pair_fwd(lib, *[t[:6].contiguous() for t in a], fl)
    hi, _ = capi.pair_fwd(lib, *[t[6:].contiguous() for t in a], fl)
    assert float(whole[4]) == float(lo[4]) + float(hi[4])  # mask counts are integers: exact
    for k in (2, 3):
        assert abs(float(whole[k]) - float(lo[k]) - float(hi[k])) <= 2e-6 * abs(float(whole[k]))


def test_forward_is_deterministic_and_backward_linear(dev):
    from scsfm_hip import _lib, capi
    lib = _lib.get()
    a = _full(dev, seed=4)
    fl = capi.make_flags(1, 1, 1, "zeros")
    o1, ws = capi.pair_fwd(lib, *a, fl)
    o2, _ = capi.pair_fwd(lib, *a, fl)
    assert torch.equal(o1, o2)  # per-block partials + ordered fp64 finalize: bit-reproducible
    one = torch.ones(1, device=dev)
    g1 = capi.pair_bwd(lib, *a, fl, ws, one, one)
    g2 = capi.pair_bwd(lib, *a, fl, ws, 2 * one, 2 * one)
    gp = capi.pair_bwd(lib, *a, fl, ws, one, 0 * one)
    gg = capi.pair_bwd(lib, *a, fl, ws, 0 * one, one)
    for x1, x2, xp, xg in zip(g1, g2, gp, gg):
        s = float(x1.abs().max())
        assert float((x2 - 2 * x1).abs().max()) <= 1e-4 * s       # homogeneity (atomics reorder sums)
        assert float((xp + xg - x1).abs().max()) <= 1e-4 * s      # additivity in the two upstream gradients


def test_batch_permutation_equivariance(dev):
    from scsfm_hip import _lib, capi
    lib = _lib.get()
    a = _full(dev, seed=5, B=4)
    perm = torch.tensor([2, 0, 3, 1], device=dev)
    fl = capi.make_flags(1, 1, 1, "zeros")
    one = torch.ones(1, device=dev)
    o, ws = capi.pair_fwd(lib, *a, fl)
    g = capi.pair_bwd(lib, *a, fl, ws, one, one)
    ap = [t[perm].contiguous() for t in a]
    op, wsp = capi.pair_fwd(lib, *ap, fl)
    gp = capi.pair_bwd(lib, *ap, fl, wsp, one, one)
    assert float(o[4]) == float(op[4]) and abs(float(o[0]) - float(op[0])) <= 1e-6
    assert float((g[0][perm] - gp[0]).abs().max()) <= 1e-6 * float(g[0].abs().max()) + 1e-12
    assert float((g[2][perm] - gp[2]).abs().max()) <= 1e-4 * float(g[2].abs().max())


# ------------------------------------------------------------------------------------------------
# 4. boundary behaviour
# ------------------------------------------------------------------------------------------------
def test_error_behaviour_matches_the_reference(LF, IW, dev):
    from scsfm_hip import synth
    d = synth.make_batch(2, 32, 64, n_ref=1, seed=0)
    img, dep, pose, K = (d["tgt_img"].to(dev), d["tgt_depth"][0].to(dev), d["poses"][0].to(dev), d["intrinsics"].to(dev))
    with pytest.raises(AssertionError, match="wrong size for depth, expected Bx1xHxW"):
        IW.inverse_warp2(img, dep.squeeze(1), dep, pose, K)  # check_sizes, inverse_warp.py:20-26
    with pytest.raises(AssertionError, match="wrong size for pose"):
        IW.inverse_warp2(img, dep, dep, pose[:, :5], K)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        LF.compute_pairwise_loss(img.cpu(), img.cpu(), dep.cpu(), dep.cpu(), pose.cpu(), K.cpu(), 1, 1, 1, "zeros")


def test_ssim_module_and_mean_on_mask(LF, dev):
    from oracle import scsfm_oracle as O
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 64, 96, generator=g)
    y = (x + 0.3 * torch.rand(2, 3, 64, 96, generator=g)).contiguous()
    xd, yd = _leaf(x, dev), _leaf(y, dev)
    out = LF.compute_ssim_loss(xd, yd)
    w = torch.rand(2, 3, 64, 96, generator=g)
    (out * w.to(dev)).sum().backward()
    xc, yc = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    oc = O.ssim_map(xc, yc)
    (oc * w).sum().backward()
    assert float((out.cpu() - oc).abs().max()) <= 3e-5
    _scale_close(xd.grad, xc.grad.numpy(), rel=1e-3, bad=1e-3)
    _scale_close(yd.grad, yc.grad.numpy(), rel=1e-3, bad=1e-3)
    diff = torch.rand(2, 3, 80, 112, generator=g)
    mask = (torch.rand(2, 1, 80, 112, generator=g) > 0.3).float()
    dd = _leaf(diff, dev)
    m = LF.mean_on_mask(dd, mask.to(dev))
    m.backward()
    dc = diff.clone().requires_grad_(True)
    mc = O.mean_on_mask(dc, mask)
    mc.backward()
    assert abs(float(m) - float(mc)) <= 1e-6
    assert float((dd.grad.cpu() - dc.grad).abs().max()) <= 1e-9
    small = LF.mean_on_mask(dd[:, :, :20, :20].contiguous(), mask[:, :, :20, :20].contiguous().to(dev))
    assert float(small) == 0.0  # 800 pixels x 3 channels: below the 10000 gate


def test_legacy_inverse_warp(IW, dev):
    from oracle import scsfm_oracle as O
    from scsfm_hip import synth
    d = synth.make_batch(2, 64, 96, n_ref=1, seed=9)
    img, dep, pose, K = d["ref_imgs"][0], d["tgt_depth"][0], d["poses"][0], d["intrinsics"]
    w, valid = IW.inverse_warp(img.to(dev), dep.squeeze(1).to(dev), pose.to(dev), K.to(dev), "euler", "zeros")
    # oracle: the legacy path has no coordinate overwrite -> same as 'border'-style projection with zero padding
    Kinv = O.inv3x3(K, "explicit")
    cam = O.back_project(dep.squeeze(1), Kinv)
    P = K @ O.pose_vec2mat(pose)
    xn, yn, _ = O.project(cam, P[:, :, :3], P[:, :, 3:], "border")  # 'border' = no overwrite
    ow = O.bilinear_sample(img, xn, yn, "zeros", "explicit")
    ov = torch.maximum(xn.abs(), yn.abs()) <= 1
    assert (valid.cpu() != ov).double().mean() <= 1e-3
    assert_close_frac(w.cpu().numpy(), ow.numpy(), atol=3e-4, max_bad_frac=2e-3)


def test_speculation_holding_failing_and_absent_agree(dev):
    """The three ways a training step can run on the device -- the speculative forward's results stand (hint =
    the upstream weights), they do not (wrong hint: the backward's own passes run behind their guards, with real
    concurrency, which the host simulation cannot show), no speculation at all -- give the same losses and the
    same gradients at full size."""
    from scsfm_hip import _lib, capi, synth
    lib = _lib.get()
    d = synth.make_batch(6, 256, 832, n_ref=2, seed=11, depth="smooth", image="smooth", dataset="kitti")
    to = lambda t: t.to(dev)
    tgt, K, refs = to(d["tgt_img"]), to(d["intrinsics"]), [to(t) for t in d["ref_imgs"]]
    tds, rds = [to(d["tgt_depth"][0])], [[to(r[0])] for r in d["ref_depths"]]
    ps, pis = [to(p) for p in d["poses"]], [to(p) for p in d["poses_inv"]]
    fl = capi.make_flags(1, 1, 1, "zeros")
    w_photo, w_geom = 0.3, 1.1
    gp, gg = torch.full((1,), w_photo, device=dev), torch.full((1,), w_geom, device=dev)
    import numpy as np
    exact = (float(np.float32(w_photo)), float(np.float32(w_geom)))
    results = {}
    for name, hint in (("holds", exact), ("fails", (1.0, 0.5)), ("absent", None)):
        photo, geom, _, ws = capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=hint)
        g_td, g_rd, g_p, g_pi = capi.photo_geometry_bwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, ws, gp, gg)
        results[name] = (float(photo), float(geom), [g_td[0]] + [r[0] for r in g_rd] + list(g_p) + list(g_pi))
    ref = results["absent"]
    for name in ("holds", "fails"):
        got = results[name]
        assert abs(got[0] - ref[0]) <= 1e-6 and abs(got[1] - ref[1]) <= 1e-6, name
        for a, b in zip(got[2], ref[2]):
            scale = float(b.abs().max())
            # The speculative forward (per-wave register pipeline, hat-function tap weights, fixed-point scatter
            # window) and the backward's own two passes (LDS tiles) evaluate the same formulas in different
            # orders: isolated pixels whose discontinuous gates (valid / auto mask, clamps) round the other way
            # differ by their full value, everything else agrees to fp32 round-off; a pose gradient
            # is a sum that a handful of such near pixels can move by ~1 %.
            if a.dim() == 4:
                assert_close_frac(a.cpu().numpy(), b.cpu().numpy(), atol=2e-4 * scale, max_bad_frac=1e-3, what=name)
            else:
                assert float((a - b).abs().max()) <= 1.5e-2 * scale, (name, scale)


@pytest.mark.parametrize("hint,upstream", [((1.0, 0.5), (1.0, 0.5)), ((1.0, 0.5), (0.3, 1.1))])
def test_no_result_depends_on_uninitialised_memory(dev, monkeypatch, hint, upstream):
    """Every torch.empty of the wrappers poisoned with NaN bytes: losses identical, gradients equal up to the
    order of the scatter's atomics (speculation holding, and failing over to the backward's own passes)."""
    from scsfm_hip import _lib, capi, synth
    lib = _lib.get()
    d = synth.make_batch(3, 128, 416, n_ref=2, seed=13, depth="smooth", image="smooth", dataset="kitti")
    to = lambda t: t.to(dev)
    tgt, K, refs = to(d["tgt_img"]), to(d["intrinsics"]), [to(t) for t in d["ref_imgs"]]
    tds, rds = [to(d["tgt_depth"][0])], [[to(r[0])] for r in d["ref_depths"]]
    ps, pis = [to(p) for p in d["poses"]], [to(p) for p in d["poses_inv"]]
    fl = capi.make_flags(1, 1, 1, "zeros")
    gp, gg = torch.full((1,), upstream[0], device=dev), torch.full((1,), upstream[1], device=dev)

    def step():
        photo, geom, _, ws = capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=hint)
        g = capi.photo_geometry_bwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, ws, gp, gg)
        sm, sws = capi.smooth_multi_fwd(lib, tds + [r[0] for r in rds], [tgt] + refs)
        gs = capi.smooth_multi_bwd(lib, tds + [r[0] for r in rds], [tgt] + refs, sws, torch.ones(1, device=dev))
        flat = [g[0][0]] + [r[0] for r in g[1]] + list(g[2]) + list(g[3]) + list(gs)
        return [photo.clone(), geom.clone(), sm.clone()], [t.clone() for t in flat]

    clean_l, clean_g = step()
    real_empty = torch.empty

    def poisoned(*a, **k):
        t = real_empty(*a, **k)
        if t.dtype == torch.uint8:
            t.fill_(255)
        elif t.is_floating_point():
            t.fill_(float("nan"))
        return t

    monkeypatch.setattr(torch, "empty", poisoned)
    dirty_l, dirty_g = step()
    monkeypatch.undo()
    for a, b in zip(dirty_l, clean_l):
        assert torch.equal(a, b)
    for a, b in zip(dirty_g, clean_g):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


def test_single_node_step_equals_the_three_reference_style_calls(LF, dev):
    """compute_total_loss (extension: one autograd node for both losses and the weighted sum) against the
    reference's call structure on the device, values and gradients."""
    from scsfm_hip import synth
    d = synth.make_batch(4, 128, 416, n_ref=2, seed=17, depth="smooth", image="smooth", dataset="kitti")
    to = lambda t: t.to(dev)
    tgt, K, refs = to(d["tgt_img"]), to(d["intrinsics"]), [to(t) for t in d["ref_imgs"]]
    w1, w2, w3 = 1.0, 0.1, 0.5

    def leaves():
        mk = lambda t: to(t).clone().requires_grad_(True)
        return ([mk(t) for t in d["tgt_depth"]], [[mk(t) for t in r] for r in d["ref_depths"]],
                [mk(p) for p in d["poses"]], [mk(p) for p in d["poses_inv"]])

    td, rd, pp, pi = leaves()
    photo, geom = LF.compute_photo_and_geometry_loss(tgt, refs, K, td, rd, pp, pi, 1, 1, 1, 1, "zeros")
    smooth = LF.compute_smooth_loss(td, tgt, rd, refs)
    (w1 * photo + w2 * smooth + w3 * geom).backward()
    td2, rd2, pp2, pi2 = leaves()
    loss, l1, l2, l3 = LF.compute_total_loss(tgt, refs, K, td2, rd2, pp2, pi2, 1, 1, 1, 1, "zeros", w1, w2, w3)
    loss.backward()
    assert float(l1) == float(photo) and float(l2) == float(smooth) and float(l3) == float(geom)
    assert abs(float(loss) - float(w1 * photo + w2 * smooth + w3 * geom)) <= 1e-6
    for a, b in zip(td + [t for r in rd for t in r] + pp + pi, td2 + [t for r in rd2 for t in r] + pp2 + pi2):
        scale = float(a.grad.abs().max())
        assert float((b.grad - a.grad).abs().max()) <= 2e-5 * scale


# ------------------------------------------------------------------------------------------------
# 7. fp32 product path against fp64, entry by entry, away from the gates
# ------------------------------------------------------------------------------------------------
def _unsafe_maps(O, d, n_ref, pad):
    """Per depth map (target, reference 0 .. n_ref - 1): the entries of its gradient that may depend on which side an
    fp32 gate fell (oracle.pairwise_gate_margins in fp64), over every pair-direction that touches the map."""
    c = lambda t: t.double()
    ti, K = c(d["tgt_img"]), c(d["intrinsics"])
    unsafe = [torch.zeros(ti.shape[0], ti.shape[2], ti.shape[3], dtype=torch.bool) for _ in range(1 + n_ref)]
    for i in range(n_ref):
        ri, td, rd = c(d["ref_imgs"][i]), c(d["tgt_depth"][0]), c(d["ref_depths"][i][0])
        for (a_img, b_img, a_d, b_d, pose, ia, ib) in ((ti, ri, td, rd, c(d["poses"][i]), 0, 1 + i),
                                                       (ri, ti, rd, td, c(d["poses_inv"][i]), 1 + i, 0)):
            m = O.pairwise_gate_margins(a_img, b_img, a_d, b_d, pose, K, 1, 1, 1, pad)
            dense, scatter = O.unsafe_gradient_entries(m)
            unsafe[ia] |= dense
            unsafe[ib] |= scatter
    return unsafe


@pytest.mark.parametrize("B,depth,pad", [(12, "smooth", "zeros"), (4, "iid", "zeros"), (4, "smooth", "border"),
                                         (4, "scene", "zeros")])
def test_depth_gradients_entrywise_away_from_the_gates(LF, dev, B, depth, pad):
    """What the fp64 instantiations cannot reach -- the fp32-only code of the product (LDS aliasing, staged taps, the
    fixed-point scatter window) -- judged entry by entry at BASELINE size, with NO outlier allowance.

    Every entry of the three depth gradients of compute_photo_and_geometry_loss whose value cannot hinge on a gate
    decided within fp32 round-off (oracle.pairwise_gate_margins, evaluated in fp64; >= 93 % of the entries, 91 % on iid inputs) must lie
    within ENTRYWISE_MAX_FACTOR[depth] x the worst such entry of the reference's own fp32 arithmetic in the same run -- the entries set aside
    are off by up to a third of it, in the reference's own fp32 arithmetic as much as here (measured,
    tools/diag_gates.py) -- and the error distribution over those entries (median, 99 %, 99.9 %) must be no wider than
    twice that of the reference's fp32 arithmetic against the same fp64 values (99.99 %: 2.5 x).  (fp32 against fp64 cannot be asked
    for more: sigma = E[x^2] - mu^2 and I[x0 + 1] - I[x0] cancel, and the reference's own fp32 entries sit at a median
    of 4e-7, a 99.9 % quantile of 3e-5 and a maximum of 5e-4 of the scale.)"""
    from oracle import scsfm_oracle as O
    from scsfm_hip import synth
    H, W, n_ref = 256, 832, 2
    d = synth.make_batch(B, H, W, n_ref=n_ref, seed=29, depth=depth, image=synth.image_law(depth), dataset="kitti")
    flags = (1, 1, 1, pad)

    def run(device, fn, dtype):
        mv = lambda t: t.to(device=device, dtype=dtype).clone().requires_grad_(True)
        cv = lambda t: t.to(device=device, dtype=dtype)
        td, rd = [mv(d["tgt_depth"][0])], [[mv(r[0])] for r in d["ref_depths"]]
        ps, pi = [mv(p) for p in d["poses"]], [mv(p) for p in d["poses_inv"]]
        photo, geom = fn(cv(d["tgt_img"]), [cv(r) for r in d["ref_imgs"]], cv(d["intrinsics"]), td, rd, ps, pi, 1, *flags)
        (photo + 0.5 * geom).backward()
        return [float(photo.detach()), float(geom.detach())], [g.grad.detach().cpu().double() for g in td + [r[0] for r in rd]]

    vh, gh = run(dev, LF.compute_photo_and_geometry_loss, torch.float32)
    v32, g32 = run("cpu", O.photo_and_geometry_loss, torch.float32)
    v64, g64 = run("cpu", O.photo_and_geometry_loss, torch.float64)
    assert abs(vh[0] - v64[0]) <= 1e-5 and abs(vh[1] - v64[1]) <= 1e-5, (vh, v64)
    unsafe = _unsafe_maps(O, d, n_ref, pad)
    for i, (a, o, c, u) in enumerate(zip(gh, g32, g64, unsafe)):
        a, o, c = a[:, 0], o[:, 0], c[:, 0]
        keep = ~u
        share = float(keep.double().mean())
        scale = float(c.abs().max())
        eh, eo = ((a - c).abs() / scale)[keep], ((o - c).abs() / scale)[keep]
        q = lambda t: [float(torch.quantile(t[::3], p)) for p in (0.5, 0.99, 0.999, 0.9999)]
        qh, qo = q(eh), q(eo)
        # into the session's summary (the driver's log keeps it): judged share, HIP worst / reference-fp32 worst, the quantile
        # ratios, and -- round-5 advisor -- the entries SET ASIDE are not unjudged altogether: their worst error is reported
        # against the reference arithmetic's worst on the same set (both are off by up to a third of the scale there)
        es_h, es_o = ((a - c).abs() / scale)[u], ((o - c).abs() / scale)[u]
        report(f"entrywise [{depth}/{pad} B={B}] map {i}: judged {share:.4f}; worst hip {float(eh.max()):.2e} / ref-fp32 {float(eo.max()):.2e} = "
               f"{float(eh.max()) / max(float(eo.max()), 1e-30):.2f}x; quantile ratios (50, 99, 99.9, 99.99 %) "
               + " ".join(f"{x / max(y, 1e-30):.2f}" for x, y in zip(qh, qo))
               + f"; set aside: worst hip {float(es_h.max()):.2e} / ref-fp32 {float(es_o.max()):.2e}, medians {float(es_h.median()):.2e} / {float(es_o.median()):.2e}")
        # the set-aside entries: a genuine kernel error confined to them would show as a distribution wider than the
        # reference arithmetic's own on the same set (3x on the median, a loose sanity bound)
        assert float(es_h.median()) <= 3.0 * float(es_o.median()) + 1e-7, (i, float(es_h.median()), float(es_o.median()))
        assert share >= ENTRYWISE_MIN_SHARE[depth], (i, share)
        # the worst judged entry: no further from fp64 than ENTRYWISE_MAX_FACTOR x the worst entry of the reference's own
        # fp32 arithmetic in this very run (round 3 used constants: 3e-3, and 1e-1 on iid inputs)
        assert float(eh.max()) <= ENTRYWISE_MAX_FACTOR[depth] * float(eo.max()) + 1e-6, (i, float(eh.max()), float(eo.max()))
        # median, 99 %, 99.9 %: twice the reference arithmetic's; the 99.99 % quantile is the ~27th largest of the ~270 k sampled
        # entries -- a tail statistic: 2.5 x (measured over the four cases, session r05: 0.5 .. 2.0, the 2.0 on the
        # reference-1 map under border padding, whose worst entry is also the suite's largest at 2.9 x)
        for x, y, f in zip(qh, qo, ENTRYWISE_QUANTILE_FACTORS):
            assert x <= f * y + 1e-7, (i, qh, qo)


def test_other_loss_weights_converge_to_the_speculative_path(LF, dev):
    """The reference's call structure with -p 1 -c 0.3 and NO set_weight_hint (a drop-in user of the reference's
    train.py): the first step mis-speculates on the default 1 : 0.5 and runs the backward's own passes; the backward
    leaves the weights it saw on the device (scsfm_pair_desc::hint), and from the second step on the forward's results
    stand: the backward is the guards + the combine.  Gradients equal the no-speculation path's in every step."""
    from scsfm_hip import config, synth
    d = synth.make_batch(12, 256, 832, n_ref=2, seed=5, depth="smooth", image="smooth", dataset="kitti")
    to = lambda t: t.to(dev)
    tgt, K, refs = to(d["tgt_img"]), to(d["intrinsics"]), [to(t) for t in d["ref_imgs"]]
    w1, w3 = 1.0, 0.3

    def step():
        mv = lambda t: t.to(dev).clone().requires_grad_(True)
        td, rd = [mv(d["tgt_depth"][0])], [[mv(r[0])] for r in d["ref_depths"]]
        ps, pi = [mv(p) for p in d["poses"]], [mv(p) for p in d["poses_inv"]]
        photo, geom = LF.compute_photo_and_geometry_loss(tgt, refs, K, td, rd, ps, pi, 1, 1, 1, 1, "zeros")
        loss = w1 * photo + w3 * geom
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        loss.backward()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1), [float(photo.detach()), float(geom.detach())], \
            [t.grad.clone() for t in td + [r[0] for r in rd] + ps + pi]

    config.set_weight_hint(None, None)
    try:
        _, v_ref, g_ref = step()
    finally:
        config.set_weight_hint(1.0, 0.5)  # the package default
    times = []
    for i in range(4):
        t, v, g = step()
        times.append(t)
        assert abs(v[0] - v_ref[0]) <= 1e-6 and abs(v[1] - v_ref[1]) <= 1e-6, (i, v, v_ref)
        hint = config.hint_tensor(torch.device(dev)).tolist()
        assert abs(hint[0] - 1.0) < 1e-12 and abs(hint[1] - float(np.float32(0.3))) < 1e-12, hint
        for a, b in zip(g, g_ref):
            scale = float(b.abs().max())
            if a.dim() == 4:
                assert_close_frac(a.cpu().numpy(), b.cpu().numpy(), atol=2e-4 * scale, max_bad_frac=1e-3, what=f"step {i}")
            else:
                assert float((a - b).abs().max()) <= 1.5e-2 * scale, (i, scale)
    print("backward ms per step (first mis-speculates):", [round(t, 3) for t in times])
    # (informative only: around loss.backward() the host needs ~0.4 ms to enqueue autograd's own kernels, which hides the
    # guards + combine, ~0.03 ms, and part of the fall-back's ~0.65 ms)
    # The same through the library calls, where the workspace can be inspected: sums[8] == 1 after a backward means the
    # forward's speculation stood (the fall-back retires it to 0).
    from scsfm_hip import _lib, capi
    lib = _lib.get()
    tds, rds = [to(d["tgt_depth"][0])], [[to(r[0])] for r in d["ref_depths"]]
    ps, pis = [to(p) for p in d["poses"]], [to(p) for p in d["poses_inv"]]
    fl = capi.make_flags(1, 1, 1, "zeros")
    hint_dev = torch.tensor([1.0, 0.5], dtype=torch.float64, device=dev)
    gp, gg = torch.full((1,), w1, device=dev), torch.full((1,), w3, device=dev)
    ws_bytes, scratch_bytes, _ = capi._sizes(lib, 12, 256, 832)
    held = []
    for _ in range(3):
        _, _, _, ws = capi.photo_geometry_fwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, hint=(1.0, 0.5), hint_dev=hint_dev)
        capi.photo_geometry_bwd(lib, fl, tgt, K, refs, tds, rds, ps, pis, ws, gp, gg, hint_dev=hint_dev)
        stride = ws_bytes + scratch_bytes
        held.append([float(ws[j * stride + 256 * 12 + 64:j * stride + 256 * 12 + 72].view(torch.float64)) for j in range(4)])
    assert held == [[0.0] * 4, [1.0] * 4, [1.0] * 4], held
    config.set_weight_hint(1.0, 0.5)  # (the package default again, on the host and in the device copy)


def test_baseline_size_fixture_recorded_from_the_reference(LF, dev):
    """tests/golden/cfg1_reference.npz: the unmodified reference (CPU, fp32) at BASELINE.json configs[1] size -- 12 x 256 x
    832, 2 refs, SSIM + mask + auto-mask -- recorded in the build container (oracle/make_golden.py: gen_cfg1).  Losses to
    1e-5; every gradient's sum / sum of magnitudes / l2 norm / probe projection; a strided sample of each depth gradient
    entry-wise; the pose gradients in full."""
    from scsfm_hip import synth
    z = load_npz("cfg1_reference.npz")
    d = synth.make_batch(12, 256, 832, n_ref=2, seed=101, depth="smooth", image="smooth", dataset="kitti")
    chk = np.array([float(d["tgt_img"].double().sum()), float(d["tgt_depth"][0].double().sum()), float(d["poses"][0].double().sum())])
    assert np.allclose(chk, z["input_check"], rtol=1e-12, atol=0), "the seeded inputs differ from the recorded ones"
    to = lambda t: t.to(dev)
    td, rd = [_leaf(d["tgt_depth"][0], dev)], [[_leaf(r[0], dev)] for r in d["ref_depths"]]
    ps, pi = [_leaf(p, dev) for p in d["poses"]], [_leaf(p, dev) for p in d["poses_inv"]]
    tgt, refs, K = to(d["tgt_img"]), [to(r) for r in d["ref_imgs"]], to(d["intrinsics"])
    photo, geom = LF.compute_photo_and_geometry_loss(tgt, refs, K, td, rd, ps, pi, 1, 1, 1, 1, "zeros")
    smooth = LF.compute_smooth_loss(td, tgt, rd, refs)
    (1.0 * photo + 0.1 * smooth + 0.5 * geom).backward()
    for nm, a in (("photo", photo), ("geom", geom), ("smooth", smooth)):
        assert abs(float(a) - float(z[nm])) <= 1e-5, (nm, float(a), float(z[nm]))
    probe = torch.cos(0.37 * torch.arange(12 * 256 * 832, dtype=torch.float64)).float().double()
    for name, t in [("g_tgt_depth", td[0])] + [(f"g_ref{i}_depth", rd[i][0]) for i in range(2)]:
        g = t.grad.detach().cpu().double().reshape(-1)
        s, sa, l2, pr, mx = z[f"{name}/checks"]
        got = (float(g.sum()), float(g.abs().sum()), float(g.norm()), float((g * probe).sum()))
        print(f"{name}: sum {got[0]:.6e} (ref {s:.6e})  sum|.| {got[1]:.6e} ({sa:.6e})  l2 {got[2]:.6e} ({l2:.6e})  probe {got[3]:.6e} ({pr:.6e})")
        # a pixel whose gate rounds the other way moves an entry by up to a third of the scale: the
        # checksums are compared at 1e-3 of the sum of magnitudes / of the norm, the sample entry-wise
        assert abs(got[0] - s) <= 1e-3 * sa and abs(got[1] - sa) <= 1e-3 * sa and abs(got[2] - l2) <= 2e-3 * l2 and abs(got[3] - pr) <= 1e-3 * sa, name
        assert_close_frac(g[::997].numpy(), z[f"{name}/sample"].astype(np.float64), atol=2e-3 * mx, rtol=1e-3, max_bad_frac=2e-3, what=name)
    for i in range(2):
        for nm, t in ((f"g_pose{i}", ps[i]), (f"g_pose_inv{i}", pi[i])):
            want = z[nm].astype(np.float64)
            assert float(np.abs(t.grad.cpu().numpy() - want).max()) <= POSE_RTOL * float(np.abs(want).max()), nm


def test_entrywise_against_the_references_own_fp32_gradients_at_baseline_size(LF, dev):
    """Round-5 review: the entry-wise judgement above compares HIP with the ORACLE's fp32 run.  Here the comparand is the
    unmodified reference's own fp32 gradients -- tests/golden/cfg1_reference.npz, every 191st entry of each depth gradient
    of L = photo + 0.1 smooth + 0.5 geometry at 12 x 256 x 832, recorded by oracle/make_golden.py from the imported
    reference -- under the same gate mask and the same bounds: errors against the fp64 oracle, HIP's worst judged entry
    within ENTRYWISE_MAX_FACTOR x the reference's, median / 99 % / 99.9 % within 2 x."""
    from oracle import scsfm_oracle as O
    from scsfm_hip import synth
    z = load_npz("cfg1_reference.npz")
    d = synth.make_batch(12, 256, 832, n_ref=2, seed=101, depth="smooth", image="smooth", dataset="kitti")
    chk = np.array([float(d["tgt_img"].double().sum()), float(d["tgt_depth"][0].double().sum()), float(d["poses"][0].double().sum())])
    assert np.allclose(chk, z["input_check"], rtol=1e-12, atol=0), "the seeded inputs differ from the recorded ones"

    def run(device, fns, dtype):
        mv = lambda t: t.to(device=device, dtype=dtype).clone().requires_grad_(True)
        cv = lambda t: t.to(device=device, dtype=dtype)
        td, rd = [mv(d["tgt_depth"][0])], [[mv(r[0])] for r in d["ref_depths"]]
        ps, pi = [mv(p) for p in d["poses"]], [mv(p) for p in d["poses_inv"]]
        tgt, refs = cv(d["tgt_img"]), [cv(r) for r in d["ref_imgs"]]
        photo, geom = fns[0](tgt, refs, cv(d["intrinsics"]), td, rd, ps, pi, 1, 1, 1, 1, "zeros")
        smooth = fns[1](td, tgt, rd, refs)
        (1.0 * photo + 0.1 * smooth + 0.5 * geom).backward()
        return [g.grad.detach().cpu().double().reshape(-1) for g in td + [r[0] for r in rd]]

    gh = run(dev, (LF.compute_photo_and_geometry_loss, LF.compute_smooth_loss), torch.float32)
    g64 = run("cpu", (O.photo_and_geometry_loss, O.smooth_loss), torch.float64)
    unsafe = _unsafe_maps(O, d, 2, "zeros")
    for i, name in enumerate(("g_tgt_depth", "g_ref0_depth", "g_ref1_depth")):
        ref32 = torch.from_numpy(z[f"{name}/sample191"].astype(np.float64))
        a, c, keep = gh[i][::191], g64[i][::191], ~unsafe[i].reshape(-1)[::191]
        scale = float(g64[i].abs().max())
        eh, eo = ((a - c).abs() / scale)[keep], ((ref32 - c).abs() / scale)[keep]
        q = lambda t: [float(torch.quantile(t, p)) for p in (0.5, 0.99, 0.999)]
        qh, qo = q(eh), q(eo)
        share = float(keep.double().mean())
        report(f"entrywise vs the REFERENCE's recorded fp32 gradients [cfg1, every 191st entry] {name}: judged {share:.4f} of {keep.numel()}; "
               f"worst hip {float(eh.max()):.2e} / reference {float(eo.max()):.2e} = {float(eh.max()) / max(float(eo.max()), 1e-30):.2f}x; "
               "quantile ratios (50, 99, 99.9 %) " + " ".join(f"{x / max(y, 1e-30):.2f}" for x, y in zip(qh, qo)))
        assert share >= ENTRYWISE_MIN_SHARE["smooth"], (name, share)
        assert float(eh.max()) <= ENTRYWISE_MAX_FACTOR["smooth"] * float(eo.max()) + 1e-6, (name, float(eh.max()), float(eo.max()))
        for x, y, f in zip(qh, qo, ENTRYWISE_QUANTILE_FACTORS):
            assert x <= f * y + 1e-7, (name, qh, qo)


def test_inputs_at_odd_element_offsets_give_the_same_results(dev, LF):
    """Every tensor of a step sliced out of a larger allocation at an offset of 1 or 3 elements (4-byte aligned, no
    more): the buffer resources of the image planes, the 16-byte paths of the combining / clearing / smooth kernels and
    their scalar fall-backs must not care where a tensor starts."""
    from scsfm_hip import synth
    B, H, W = 4, 72, 100
    d = synth.make_batch(B, H, W, n_ref=2, seed=41, depth="smooth")

    def shifted(t, k):
        buf = torch.empty(t.numel() + 8, dtype=t.dtype, device=dev)
        v = buf[k:k + t.numel()].view(t.shape)
        v.copy_(t)
        assert v.is_contiguous() and v.data_ptr() % 16 == (buf.data_ptr() + 4 * k) % 16
        return v

    def run(k):
        g = lambda t, r=False: shifted(t.float(), k).requires_grad_(r) if k else t.float().to(dev).requires_grad_(r)
        ti, K = g(d["tgt_img"]), g(d["intrinsics"])
        ris = [g(r) for r in d["ref_imgs"]]
        td, rd = [g(d["tgt_depth"][0], True)], [[g(r[0], True)] for r in d["ref_depths"]]
        pp, pi = [g(p, True) for p in d["poses"]], [g(p, True) for p in d["poses_inv"]]
        photo, geom = LF.compute_photo_and_geometry_loss(ti, ris, K, td, rd, pp, pi, 1, 1, 1, 1, "zeros")
        smooth = LF.compute_smooth_loss(td, ti, rd, ris)
        (photo + 0.1 * smooth + 0.5 * geom).backward()
        return [photo.detach(), geom.detach(), smooth.detach(), td[0].grad, rd[0][0].grad, rd[1][0].grad, pp[0].grad, pi[1].grad]

    ref = run(0)
    for k in (1, 3):
        got = run(k)
        for a, b in zip(got[:3], ref[:3]):
            assert torch.equal(a, b)
        for a, b in zip(got[3:], ref[3:]):
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


# ------------------------------------------------------------------------------------------------
# 8. round 4: the reference's call structure under anomaly mode; iid pose gradients as row statistics over seeds
# ------------------------------------------------------------------------------------------------
def test_reference_call_structure_under_anomaly_mode(LF, dev):
    """The reference switches torch.autograd.set_detect_anomaly(True) on globally (train.py:67) and its loop makes
    three calls, a weighted sum and backward() (train.py:259-282).  The drop-in's autograd nodes must survive that mode
    -- it checks every gradient a node returns for NaN and keeps forward tracebacks -- with a pair whose mask is below
    the 10000-pixel gate (loss_functions.py:123-129: the term is a constant 0 there) in the batch, at configs[1] size,
    and give the very same numbers as without it."""
    from scsfm_hip import synth
    d = synth.make_batch(12, 256, 832, n_ref=2, seed=23, depth="smooth", image="smooth", dataset="kitti")
    # the second reference view is thrown far off: next to none of its pixels land inside the image in either direction
    d["poses"][1] = d["poses"][1].clone(); d["poses_inv"][1] = d["poses_inv"][1].clone()
    d["poses"][1][:, 0] = 500.0; d["poses_inv"][1][:, 0] = -500.0
    to = lambda t: t.to(dev)
    tgt, K, refs = to(d["tgt_img"]), to(d["intrinsics"]), [to(t) for t in d["ref_imgs"]]
    w1, w2, w3 = 1.0, 0.1, 0.5

    def step():
        mv = lambda t: t.to(dev).clone().requires_grad_(True)
        td, rd = [mv(d["tgt_depth"][0])], [[mv(r[0])] for r in d["ref_depths"]]
        ps, pi = [mv(p) for p in d["poses"]], [mv(p) for p in d["poses_inv"]]
        photo, geom = LF.compute_photo_and_geometry_loss(tgt, refs, K, td, rd, ps, pi, 1, 1, 1, 1, "zeros")
        smooth = LF.compute_smooth_loss(td, tgt, rd, refs)
        loss = w1 * photo + w2 * smooth + w3 * geom
        loss.backward()
        leaves = td + [r[0] for r in rd] + ps + pi
        assert all(bool(torch.isfinite(t.grad).all()) for t in leaves)
        return [float(loss.detach()), float(photo.detach()), float(smooth.detach()), float(geom.detach())], \
            [t.grad.clone() for t in leaves]

    v0, g0 = step()
    # the thrown-off pairs are below the gate: their terms are zero (single-pair call, same inputs)
    p1, q1 = LF.compute_pairwise_loss(tgt, refs[1], to(d["tgt_depth"][0]), to(d["ref_depths"][1][0]), to(d["poses"][1]), K,
                                      1, 1, 1, "zeros")
    assert float(p1) == 0.0 and float(q1) == 0.0
    # leaves = [tgt depth, ref depth 0, ref depth 1, pose 0, pose 1, pose_inv 0, pose_inv 1]
    assert float(g0[4].abs().max()) == 0.0 and float(g0[6].abs().max()) == 0.0   # ... and so are their poses' gradients
    assert float(g0[3].abs().max()) > 0.0 and float(g0[5].abs().max()) > 0.0
    prev = torch.is_anomaly_enabled()
    torch.autograd.set_detect_anomaly(True)
    try:
        v1, g1 = step()
        v2, g2 = step()
    finally:
        torch.autograd.set_detect_anomaly(prev)
    assert v0[1:] == v1[1:] == v2[1:], (v0, v1, v2)        # bit-reproducible forward
    assert len(g0) == 7
    for a, b, c in zip(g0, g1, g2):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-5 * scale + 1e-12 and float((a - c).abs().max()) <= 1e-5 * scale + 1e-12


def test_iid_pose_gradients_as_row_statistics_over_seeds(LF, dev):
    """Pose gradients on iid inputs (independent depths 0.1 .. 100 per pixel: a handful of near pixels carry d/d
    translation, and one gate that rounds the other way moves a row by per cents -- in ANY fp32 evaluation, the
    reference's own included, and on different rows in different implementations).  Judged as a distribution, pooled
    over four seeds at 4 x 256 x 832 (64 rows of [B, 6] gradients).  Row error = largest |difference to the fp64
    oracle| in the row; the HIP path's must be no larger than IID_ROW_FACTOR x those of the fp32 oracle (= the
    reference's arithmetic) + 1.5 %:
      * relative to the TENSOR's scale (how every other pose tolerance of this file is expressed): median, 90 %
        quantile and maximum;
      * relative to the ROW's own scale: median and 90 % quantile.  (Not the maximum: it is one row whose own gradient
        nearly cancels -- measured over seeds 17..22, profiles/r04_iid_pose_rows.json: HIP 0.72 on one row of seed 18,
        the reference's fp32 0.54 on one row of seed 22, each at <= 0.11 / 0.13 on the other's row; against the
        tensor's scale the same two rows are 0.33 and 0.24.)"""
    from oracle import scsfm_oracle as O
    from scsfm_hip import synth
    B, H, W, n_ref = 4, 256, 832, 2
    rows = {"hip/tensor": [], "ref/tensor": [], "hip/row": [], "ref/row": []}
    for seed in (17, 18, 19, 20):
        d = synth.make_batch(B, H, W, n_ref=n_ref, seed=seed, depth="iid", image="iid", dataset="kitti")

        def run(device, fn_pg, dtype):
            mv = lambda t: t.to(device=device, dtype=dtype).clone().requires_grad_(True)
            cv = lambda t: t.to(device=device, dtype=dtype)
            td, rd = [mv(d["tgt_depth"][0])], [[mv(r[0])] for r in d["ref_depths"]]
            ps, pi = [mv(p) for p in d["poses"]], [mv(p) for p in d["poses_inv"]]
            photo, geom = fn_pg(cv(d["tgt_img"]), [cv(r) for r in d["ref_imgs"]], cv(d["intrinsics"]), td, rd, ps, pi, 1, 1, 1, 1, "zeros")
            (photo + 0.5 * geom).backward()
            return [p.grad.detach().cpu().double() for p in ps + pi]

        gh = run(dev, LF.compute_photo_and_geometry_loss, torch.float32)
        go = run("cpu", O.photo_and_geometry_loss, torch.float32)
        g64 = run("cpu", O.photo_and_geometry_loss, torch.float64)
        for a, b, c in zip(gh, go, g64):
            eh, eo = (a - c).abs().max(dim=1).values, (b - c).abs().max(dim=1).values
            rows["hip/tensor"].append(eh / c.abs().max()); rows["ref/tensor"].append(eo / c.abs().max())
            rows["hip/row"].append(eh / c.abs().max(dim=1).values); rows["ref/row"].append(eo / c.abs().max(dim=1).values)
    stat = lambda t: (float(t.median()), float(torch.quantile(t, 0.9)), float(t.max()))
    st = {k: stat(torch.cat(v)) for k, v in rows.items()}
    print("iid pose rows (n = 64), median / p90 / max: " + " | ".join(f"{k} {v[0]:.4f} {v[1]:.4f} {v[2]:.4f}" for k, v in st.items()))
    for x, y in zip(st["hip/tensor"], st["ref/tensor"]):
        assert x <= IID_ROW_FACTOR * y + POSE_RTOL, st
    for x, y in zip(st["hip/row"][:2], st["ref/row"][:2]):
        assert x <= IID_ROW_FACTOR * y + POSE_RTOL, st
