"""Gradients with respect to the DATA inputs of the path -- images, intrinsics, a floating-point mask.  train.py never
asks for them ("images, K do not [need grad]"), but the reference's functions are plain autograd code and
give them to whoever does: the drop-in provides them as well (ABI 7: scsfm_pairs_bwd_inputs, scsfm_warp_bwd_inputs,
scsfm_pixel2cam_bwd_intrinsics, scsfm_masked_mean_bwd_mask, scsfm_smooth_multi_bwd_images).  Here: the kernels in the
CPU simulation, fp64, against the oracle's autograd (tests/test_gpu_input_gradients.py repeats it on the hardware
through the public functions)."""
import pytest
import torch

from _util import leaf
from hostsim import harness
from oracle import scsfm_oracle as O
from scsfm_hip import capi, synth


@pytest.fixture(scope="module")
def lib():
    return harness.lib()


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def _batch(B, H, W, seed, n_ref=2):
    d = synth.make_batch(B, H, W, n_ref=n_ref, seed=seed, depth="smooth")
    c = lambda x: x.double().contiguous()
    return dict(ti=c(d["tgt_img"]), K=c(d["intrinsics"]), ris=[c(r) for r in d["ref_imgs"]],
                tds=[c(d["tgt_depth"][0])], rds=[[c(r[0])] for r in d["ref_depths"]],
                ps=[c(p) for p in d["poses"]], pis=[c(p) for p in d["poses_inv"]])


@pytest.mark.parametrize("flags3,pad,hint,upstream", [
    ((1, 1, 1), "zeros", (0.7, 1.3), (0.7, 1.3)),   # the speculative forward's planes stand
    ((1, 1, 1), "zeros", (1.0, 0.5), (0.3, 1.1)),   # wrong hint: the backward ran its own passes
    ((1, 1, 1), "zeros", None, (0.7, 1.3)),         # plain forward
    ((1, 1, 0), "border", (0.7, 1.3), (0.7, 1.3)),
    ((0, 1, 1), "zeros", (1.0, 0.5), (0.3, 1.1)),   # no SSIM
    ((1, 0, 0), "zeros", None, (0.7, 1.3)),
])
def test_photo_geometry_gradients_of_images_and_intrinsics(lib, flags3, pad, hint, upstream):
    """Both after a speculative forward whose planes stand, after one whose hint was wrong (the backward's own
    passes), and after the plain forward."""
    x = _batch(7, 40, 92, seed=5)
    ti, K, ris = leaf(x["ti"]), leaf(x["K"]), [leaf(r) for r in x["ris"]]
    td, rd = [leaf(t) for t in x["tds"]], [[leaf(t) for t in r] for r in x["rds"]]
    pp, pi = [leaf(p) for p in x["ps"]], [leaf(p) for p in x["pis"]]
    po, go = O.photo_and_geometry_loss(ti, ris, K, td, rd, pp, pi, 1, *flags3, pad)
    assert float(po) > 0 and float(go) > 0
    (upstream[0] * po + upstream[1] * go).backward()
    fl = capi.make_flags(*flags3, pad)
    photo, geom, _, ws = capi.photo_geometry_fwd(lib, fl, x["ti"], x["K"], x["ris"], x["tds"], x["rds"], x["ps"], x["pis"],
                                                 hint=hint)
    t = lambda v: torch.tensor([v], dtype=torch.float64)
    g_td, g_rd, g_p, g_pi, g_imgs, g_K = capi.photo_geometry_bwd(
        lib, fl, x["ti"], x["K"], x["ris"], x["tds"], x["rds"], x["ps"], x["pis"], ws, t(upstream[0]), t(upstream[1]),
        need_imgs=[True, True, True], need_K=True)
    assert _rel(g_td[0], td[0].grad) < 1e-10 and _rel(g_p[0], pp[0].grad) < 1e-10  # (what was there before)
    assert _rel(g_K, K.grad) < 1e-9
    assert _rel(g_imgs[0], ti.grad) < 1e-10
    for i in range(2):
        assert _rel(g_imgs[1 + i], ris[i].grad) < 1e-10
    # only some of them wanted
    res = capi.photo_geometry_bwd(lib, fl, x["ti"], x["K"], x["ris"], x["tds"], x["rds"], x["ps"], x["pis"], ws,
                                  t(upstream[0]), t(upstream[1]), need_imgs=[False, True, False], need_K=False)
    assert res[4][0] is None and res[4][2] is None and res[5] is None and _rel(res[4][1], ris[0].grad) < 1e-10


def test_the_same_image_passed_twice_gets_one_gradient_per_argument(lib):
    """A reference frame repeated (or the target among the references): the gradient buffers are keyed by the
    argument's POSITION, so each occurrence receives its own gradient and autograd's sum over the occurrences is the
    oracle's gradient of the shared tensor (keyed by storage, the later occurrence used to take the earlier one's)."""
    x = _batch(3, 40, 92, seed=11)
    shared = x["ris"][0]
    ris_x = [shared, shared]                        # the SAME storage as both references
    ti, K, r = leaf(x["ti"]), leaf(x["K"]), leaf(shared)
    td, rd = [leaf(t) for t in x["tds"]], [[leaf(t) for t in rr] for rr in x["rds"]]
    pp, pi = [leaf(p) for p in x["ps"]], [leaf(p) for p in x["pis"]]
    po, go = O.photo_and_geometry_loss(ti, [r, r], K, td, rd, pp, pi, 1, 1, 1, 1, "zeros")
    (0.7 * po + 1.3 * go).backward()
    fl = capi.make_flags(1, 1, 1, "zeros")
    _, _, _, ws = capi.photo_geometry_fwd(lib, fl, x["ti"], x["K"], ris_x, x["tds"], x["rds"], x["ps"], x["pis"], hint=(0.7, 1.3))
    t = lambda v: torch.tensor([v], dtype=torch.float64)
    res = capi.photo_geometry_bwd(lib, fl, x["ti"], x["K"], ris_x, x["tds"], x["rds"], x["ps"], x["pis"], ws, t(0.7), t(1.3),
                                  need_imgs=[True, True, True], need_K=False)
    g_imgs = res[4]
    assert g_imgs[1].data_ptr() != g_imgs[2].data_ptr()
    assert float(g_imgs[1].abs().max()) > 0 and float(g_imgs[2].abs().max()) > 0
    assert _rel(g_imgs[1] + g_imgs[2], r.grad) < 1e-10
    assert _rel(g_imgs[0], ti.grad) < 1e-10


def test_single_pair_entry_points_give_the_same_input_gradients(lib):
    x = _batch(7, 40, 92, seed=9, n_ref=1)
    ti, ri, K = leaf(x["ti"]), leaf(x["ris"][0]), leaf(x["K"])
    td, rd, po_ = leaf(x["tds"][0]), leaf(x["rds"][0][0]), leaf(x["ps"][0])
    po, go = O.pairwise_loss(ti, ri, td, rd, po_, K, 1, 1, 1, "zeros")
    (0.6 * po + 1.4 * go).backward()
    fl = capi.make_flags(1, 1, 1, "zeros")
    out, ws = capi.pair_fwd(lib, x["ti"], x["ris"][0], x["tds"][0], x["rds"][0][0], x["ps"][0], x["K"], fl)
    t = lambda v: torch.tensor([v], dtype=torch.float64)
    g_td, g_rd, g_pose, g_ti, g_ri, g_K = capi.pair_bwd(lib, x["ti"], x["ris"][0], x["tds"][0], x["rds"][0][0], x["ps"][0],
                                                        x["K"], fl, ws, t(0.6), t(1.4), need_tgt_img=True,
                                                        need_ref_img=True, need_K=True)
    assert _rel(g_ti, ti.grad) < 1e-10 and _rel(g_ri, ri.grad) < 1e-10 and _rel(g_K, K.grad) < 1e-9
    assert _rel(g_td, td.grad) < 1e-10 and _rel(g_pose, po_.grad) < 1e-10


@pytest.mark.parametrize("pad", ["zeros", "border"])
def test_inverse_warp2_gradients_of_the_sampled_image_and_intrinsics(lib, pad):
    x = _batch(2, 24, 40, seed=3, n_ref=1)
    img, depth, ref_depth, pose, K = x["ris"][0], x["tds"][0], x["rds"][0][0], x["ps"][0], x["K"]
    li, ld, lr, lp, lK = leaf(img), leaf(depth), leaf(ref_depth), leaf(pose), leaf(K)
    pi_, valid, pd, cd = O.inverse_warp2(li, ld, lr, lp, lK, pad)
    g = torch.Generator().manual_seed(1)
    gi, gpd, gcd = (torch.randn(s.shape, generator=g, dtype=torch.float64) for s in (pi_, pd, cd))
    ((pi_ * gi).sum() + (pd * gpd).sum() + (cd * gcd).sum()).backward()
    fl = capi.make_flags(0, 0, 0, pad)
    g_depth, g_ref, g_pose, g_src, g_K = capi.warp_bwd(lib, img, depth, ref_depth, pose, K, fl, gi, gpd, gcd,
                                                       need_img=True, need_K=True)
    assert _rel(g_depth, ld.grad) < 1e-10 and _rel(g_pose, lp.grad) < 1e-10
    assert _rel(g_src, li.grad) < 1e-11 and _rel(g_K, lK.grad) < 1e-9


def test_pixel2cam_gradient_of_the_inverse_intrinsics(lib):
    x = _batch(3, 17, 29, seed=2, n_ref=1)
    depth = x["tds"][0][:, 0].contiguous()
    Kinv = leaf(torch.linalg.inv(x["K"]))
    cam = O.back_project(depth, Kinv)
    g = torch.randn(cam.shape, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    (cam * g).sum().backward()
    assert _rel(capi.pixel2cam_bwd_intrinsics(lib, depth, g.contiguous()), Kinv.grad) < 1e-12


@pytest.mark.parametrize("Cm", [1, 3])
@pytest.mark.parametrize("open_gate", [True, False])
def test_mean_on_mask_gradient_of_a_floating_point_mask(lib, Cm, open_gate):
    B, C, H, W = (2, 3, 64, 50) if open_gate else (1, 3, 20, 30)
    g = torch.Generator().manual_seed(7)
    diff = torch.rand(B, C, H, W, generator=g, dtype=torch.float64)
    mask = leaf(0.2 + 0.8 * torch.rand(B, Cm, H, W, generator=g, dtype=torch.float64))
    out = O.mean_on_mask(diff, mask)
    assert (float(out) > 0) == open_gate
    got_out, ws = capi.masked_mean_fwd(lib, diff, mask.detach())
    assert abs(float(got_out) - float(out)) < 1e-12
    got = capi.masked_mean_bwd_mask(lib, diff, mask.shape, ws, torch.tensor([1.7], dtype=torch.float64))
    if open_gate:
        (1.7 * out).backward()
        assert _rel(got, mask.grad) < 1e-11
    else:
        assert float(got.abs().max()) == 0.0


def test_smooth_loss_gradient_of_the_images(lib):
    x = _batch(2, 21, 37, seed=8)
    imgs = [leaf(x["ti"])] + [leaf(r) for r in x["ris"]]
    depths = [x["tds"][0]] + [r[0] for r in x["rds"]]
    L = O.smooth_loss([depths[0]], imgs[0], [[d] for d in depths[1:]], imgs[1:])
    (2.5 * L).backward()
    raw = [x["ti"]] + x["ris"]
    loss, ws = capi.smooth_multi_fwd(lib, depths, raw, keep_edges=False)
    assert abs(float(loss) - float(L)) < 1e-12
    got = capi.smooth_multi_bwd_images(lib, depths, raw, ws, torch.tensor([2.5], dtype=torch.float64), [True, False, True])
    assert got[1] is None
    assert _rel(got[0], imgs[0].grad) < 1e-11 and _rel(got[2], imgs[2].grad) < 1e-11


def test_two_scales_read_in_place_give_the_same_input_gradients(lib):
    """A coarser scale's maps go to the library as they are (scsfm_pair_desc::depth_shift): the image pass and the
    intrinsics reduction run over all 2 x n_ref x n_scales pair-directions."""
    d = synth.make_batch(7, 40, 92, n_ref=2, seed=12, depth="smooth", num_scales=2)
    c = lambda t: t.double().contiguous()
    ti, K, ris = c(d["tgt_img"]), c(d["intrinsics"]), [c(r) for r in d["ref_imgs"]]
    tds, rds = [c(t) for t in d["tgt_depth"]], [[c(t) for t in r] for r in d["ref_depths"]]
    ps, pis = [c(p) for p in d["poses"]], [c(p) for p in d["poses_inv"]]
    lti, lK, lris = leaf(ti), leaf(K), [leaf(r) for r in ris]
    po, go = O.photo_and_geometry_loss(lti, lris, lK, tds, rds, ps, pis, 2, 1, 1, 1, "zeros")
    (po + 0.5 * go).backward()
    fl = capi.make_flags(1, 1, 1, "zeros")
    photo, geom, _, ws = capi.photo_geometry_fwd(lib, fl, ti, K, ris, tds, rds, ps, pis, hint=(1.0, 0.5))
    assert abs(float(photo) - float(po)) < 1e-11
    t = lambda v: torch.tensor([v], dtype=torch.float64)
    res = capi.photo_geometry_bwd(lib, fl, ti, K, ris, tds, rds, ps, pis, ws, t(1.0), t(0.5), need_imgs=[True, True, True],
                                  need_K=True)
    assert _rel(res[5], lK.grad) < 1e-9 and _rel(res[4][0], lti.grad) < 1e-10
    for i in range(2):
        assert _rel(res[4][1 + i], lris[i].grad) < 1e-10
