"""On the MI355X, through the public functions of the drop-in modules: gradients with respect to the DATA inputs --
images, intrinsics, a floating-point mask -- which the reference's autograd gives to a caller that asks
(train.py does not: SURVEY.md 8b).  fp64 against the oracle's autograd at 1e-9; fp32 (what ships) at the tolerances of
the depth gradients.  The CPU twin of this file is tests/test_input_gradients_hostsim.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from scsfm_hip import _lib
    assert _lib.get().path.endswith("libscsfm_hip.so")
    return torch.device("cuda:0")


def _rel(a, b):
    return float((a.detach().cpu().double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))


def _frac_close(a, b, atol_of_max, bad):
    a, b = a.detach().cpu().double(), b.double()
    off = ((a - b).abs() > atol_of_max * b.abs().max() + 1e-3 * b.abs()).double().mean()
    assert float(off) <= bad, float(off)


def _make(B, H, W, seed, dtype, n_ref=2):
    from scsfm_hip import synth
    d = synth.make_batch(B, H, W, n_ref=n_ref, seed=seed, depth="smooth")
    c = lambda x: x.to(dtype).contiguous()
    return dict(ti=c(d["tgt_img"]), K=c(d["intrinsics"]), ris=[c(r) for r in d["ref_imgs"]],
                tds=[c(d["tgt_depth"][0])], rds=[[c(r[0])] for r in d["ref_depths"]],
                ps=[c(p) for p in d["poses"]], pis=[c(p) for p in d["poses_inv"]])


def _leaves(x, to):
    lf = lambda t: t.to(to).clone().requires_grad_(True)
    return dict(ti=lf(x["ti"]), K=lf(x["K"]), ris=[lf(r) for r in x["ris"]], tds=[lf(t) for t in x["tds"]],
                rds=[[lf(t) for t in r] for r in x["rds"]], ps=[lf(p) for p in x["ps"]], pis=[lf(p) for p in x["pis"]])


@pytest.mark.parametrize("weights", [(1.0, 0.5), (0.3, 1.1)])  # the configured hint, and weights it does not expect
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_compute_photo_and_geometry_loss_reaches_images_and_intrinsics(dev, dtype, weights):
    import loss_functions as LF
    from oracle import scsfm_oracle as O
    from scsfm_hip import config
    x64 = _make(4, 72, 100, 31, torch.float64)  # (the fp32 values, widened: both precisions see the same inputs)
    x = _make(4, 72, 100, 31, dtype)
    o = _leaves(x64, "cpu")
    po, go = O.photo_and_geometry_loss(o["ti"], o["ris"], o["K"], o["tds"], o["rds"], o["ps"], o["pis"], 1, 1, 1, 1, "zeros")
    (weights[0] * po + weights[1] * go).backward()
    g = _leaves(x, dev)
    try:
        photo, geom = LF.compute_photo_and_geometry_loss(g["ti"], g["ris"], g["K"], g["tds"], g["rds"], g["ps"], g["pis"],
                                                         1, 1, 1, 1, "zeros")
        (weights[0] * photo + weights[1] * geom).backward()
    finally:
        config.set_weight_hint(1.0, 0.5)
    assert abs(float(photo) - float(po)) <= (1e-10 if dtype == torch.float64 else 1e-5)
    if dtype == torch.float64:
        assert _rel(g["K"].grad, o["K"].grad) < 1e-9 and _rel(g["ti"].grad, o["ti"].grad) < 1e-9
        for a, b in zip(g["ris"], o["ris"]):
            assert _rel(a.grad, b.grad) < 1e-9
        assert _rel(g["tds"][0].grad, o["tds"][0].grad) < 1e-9 and _rel(g["ps"][0].grad, o["ps"][0].grad) < 1e-9
    else:
        # (fp32: the path's gates flip isolated pixels -- same yardstick as the depth gradients; K sums every pixel's
        # geometry term like the pose does)
        assert _rel(g["K"].grad, o["K"].grad) < 3e-2
        _frac_close(g["ti"].grad, o["ti"].grad, 2e-3, 3e-3)
        for a, b in zip(g["ris"], o["ris"]):
            _frac_close(a.grad, b.grad, 2e-3, 3e-3)


@pytest.mark.parametrize("pad", ["zeros", "border"])
def test_inverse_warp2_reaches_the_sampled_image_and_intrinsics(dev, pad):
    import inverse_warp as IW
    from oracle import scsfm_oracle as O
    x = _make(2, 40, 72, 5, torch.float64, n_ref=1)
    o = {k: v.clone().requires_grad_(True) for k, v in dict(img=x["ris"][0], d=x["tds"][0], rd=x["rds"][0][0], p=x["ps"][0],
                                                           K=x["K"]).items()}
    outs = O.inverse_warp2(o["img"], o["d"], o["rd"], o["p"], o["K"], pad)
    gen = torch.Generator().manual_seed(2)
    ups = [torch.randn(t.shape, generator=gen, dtype=torch.float64) for t in outs]
    sum((t * u).sum() for t, u, k in zip(outs, ups, range(4)) if k != 1).backward()
    g = {k: v.detach().to(dev).requires_grad_(True) for k, v in o.items()}
    got = IW.inverse_warp2(g["img"], g["d"], g["rd"], g["p"], g["K"], pad)
    sum((t * u.to(dev)).sum() for t, u, k in zip(got, ups, range(4)) if k != 1).backward()
    for k in o:
        assert _rel(g[k].grad, o[k].grad) < 1e-9, k


@pytest.mark.parametrize("mode", ["euler", "quat"])
def test_legacy_inverse_warp_reaches_the_intrinsics(dev, mode):
    """inverse_warp (inverse_warp.py:157-191) against the oracle's maps assembled the legacy way: no coordinate
    overwrite, either rotation mode."""
    import inverse_warp as IW
    from oracle import scsfm_oracle as O
    x = _make(2, 40, 72, 6, torch.float64, n_ref=1)
    img, depth, pose, K = x["ris"][0], x["tds"][0][:, 0].contiguous(), x["ps"][0], x["K"]
    o = {k: v.clone().requires_grad_(True) for k, v in dict(img=img, d=depth, p=pose, K=K).items()}
    cam = O.back_project(o["d"], torch.linalg.inv(o["K"]))
    M = O.pose_vec2mat(o["p"], mode)
    A = o["K"] @ M[:, :, :3]
    c = (o["K"] @ M[:, :, 3:]).squeeze(-1)
    B, _, H, W = img.shape
    p = A @ cam.reshape(B, 3, -1) + c.unsqueeze(-1)
    Z = p[:, 2].clamp(min=1e-3)
    xn = (2 * (p[:, 0] / Z) / (W - 1) - 1).reshape(B, H, W)
    yn = (2 * (p[:, 1] / Z) / (H - 1) - 1).reshape(B, H, W)
    warped = torch.nn.functional.grid_sample(o["img"], torch.stack([xn, yn], dim=-1), padding_mode="zeros",
                                             align_corners=False)
    up = torch.randn(warped.shape, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    (warped * up).sum().backward()
    g = {k: v.detach().to(dev).requires_grad_(True) for k, v in o.items()}
    got, _ = IW.inverse_warp(g["img"], g["d"], g["p"], g["K"], rotation_mode=mode)
    assert _rel(got, warped.detach()) < 1e-10
    (got * up.to(dev)).sum().backward()
    for k in o:
        assert _rel(g[k].grad, o[k].grad) < 1e-8, k


def test_pixel2cam_mean_on_mask_and_smooth_loss_reach_their_data_inputs(dev):
    import inverse_warp as IW
    import loss_functions as LF
    from oracle import scsfm_oracle as O
    x = _make(2, 64, 90, 8, torch.float64)
    # pixel2cam: intrinsics_inv
    depth = x["tds"][0][:, 0].contiguous()
    ko, kg = torch.linalg.inv(x["K"]).requires_grad_(True), torch.linalg.inv(x["K"]).to(dev).requires_grad_(True)
    up = torch.randn(2, 3, 64, 90, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    (O.back_project(depth, ko) * up).sum().backward()
    (IW.pixel2cam(depth.to(dev), kg) * up.to(dev)).sum().backward()
    assert _rel(kg.grad, ko.grad) < 1e-11
    # mean_on_mask: a floating-point mask
    gen = torch.Generator().manual_seed(4)
    diff = torch.rand(2, 3, 64, 90, generator=gen, dtype=torch.float64)
    m0 = 0.2 + 0.8 * torch.rand(2, 1, 64, 90, generator=gen, dtype=torch.float64)
    mo, mg = m0.clone().requires_grad_(True), m0.to(dev).requires_grad_(True)
    do, dg = diff.clone().requires_grad_(True), diff.to(dev).requires_grad_(True)
    O.mean_on_mask(do, mo).backward()
    LF.mean_on_mask(dg, mg).backward()
    assert _rel(mg.grad, mo.grad) < 1e-11 and _rel(dg.grad, do.grad) < 1e-11
    # compute_smooth_loss: the images
    o, g = _leaves(x, "cpu"), _leaves(x, dev)
    O.smooth_loss(o["tds"], o["ti"], o["rds"], o["ris"]).backward()
    LF.compute_smooth_loss(g["tds"], g["ti"], g["rds"], g["ris"]).backward()
    assert _rel(g["ti"].grad, o["ti"].grad) < 1e-10 and _rel(g["ris"][1].grad, o["ris"][1].grad) < 1e-10
    assert _rel(g["tds"][0].grad, o["tds"][0].grad) < 1e-10


def test_single_node_step_gives_the_same_input_gradients_as_the_three_calls(dev):
    """compute_total_loss (this package's extension: both losses and the weighted sum behind one autograd node) against
    compute_photo_and_geometry_loss + compute_smooth_loss + the weighted sum, fp64."""
    import loss_functions as LF
    x = _make(4, 72, 100, 33, torch.float64)
    a, b = _leaves(x, dev), _leaves(x, dev)
    w = (0.9, 0.2, 0.6)
    p, g = LF.compute_photo_and_geometry_loss(a["ti"], a["ris"], a["K"], a["tds"], a["rds"], a["ps"], a["pis"], 1, 1, 1, 1, "zeros")
    s = LF.compute_smooth_loss(a["tds"], a["ti"], a["rds"], a["ris"])
    (w[0] * p + w[1] * s + w[2] * g).backward()
    loss = LF.compute_total_loss(b["ti"], b["ris"], b["K"], b["tds"], b["rds"], b["ps"], b["pis"], 1, 1, 1, 1, "zeros", *w)[0]
    loss.backward()
    pairs = [(a["ti"], b["ti"]), (a["K"], b["K"]), (a["tds"][0], b["tds"][0]), (a["ps"][1], b["ps"][1])] + list(zip(a["ris"], b["ris"]))
    for u, v in pairs:
        assert _rel(v.grad, u.grad.cpu()) < 1e-10
