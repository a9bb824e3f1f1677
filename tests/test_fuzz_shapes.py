"""Seeded sweep of the pair-loss entry points over shapes nobody chose: image sizes that are no multiple of a tile, a
wave or a strip in either direction (down to images lower than a strip and narrower than a wave), 1-3 references, every
flag combination of compute_pairwise_loss (loss_functions.py:95-119), both paddings (inverse_warp.py:262,267), and the
three states of the speculation (hint right / hint wrong -> the backward's own passes / no hint -> plain forward), fp64
instantiations against the fp64 oracle.  The same cases run on the kernel simulator (CPU CI) and on the hardware.

The cases are drawn once from a fixed seed, so a failure names a reproducible case; the batch is sized so that the
10000-pixel gates of mean_on_mask (loss_functions.py:125) are open in most cases and closed in a few (both must hold)."""
import random

import pytest
import torch

from oracle import scsfm_oracle as O
from scsfm_hip import capi, synth


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        H = rng.choice([3, 5, 9, 14, 15, 17, 29, 31, 33, 47, 61])
        W = rng.choice([5, 33, 61, 62, 63, 64, 65, 67, 125, 127, 129, 190])
        # most cases above the gate (B*H*W > 10000 and a fair share of valid pixels), every fourth below it
        want = 3000 if i % 4 == 3 else 26000
        B = max(1, min(48, -(-want // (H * W))))
        n_ref = rng.choice([1, 1, 2, 3])
        flags3 = (rng.randint(0, 1), rng.randint(0, 1), rng.randint(0, 1))
        pad = rng.choice(["zeros", "zeros", "border"])
        state = rng.choice(["holds", "wrong", "plain"])
        up = (rng.choice([1.0, 0.7, 0.25]), rng.choice([0.5, 1.3, 0.0]))
        out.append((H, W, B, n_ref, flags3, pad, state, up, 1000 + i))
    return out


CASES = _cases(20, seed=20260930)
IDS = [f"{H}x{W}x{B}-r{n}-f{''.join(map(str, f))}-{pad}-{state}" for H, W, B, n, f, pad, state, up, seed in CASES]


def _rel(a, b):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-300)) if float(b.abs().max()) > 0 else float(a.abs().max())


def _run_case(lib, case, device):
    H, W, B, n_ref, flags3, pad, state, up, seed = case
    d = synth.make_batch(B, H, W, n_ref=n_ref, seed=seed, depth="smooth")
    c = lambda x: x.double().contiguous()
    ti, K = c(d["tgt_img"]), c(d["intrinsics"])
    ris = [c(r) for r in d["ref_imgs"]]
    tds, rds = [c(d["tgt_depth"][0])], [[c(r[0])] for r in d["ref_depths"]]
    ps, pis = [c(p) for p in d["poses"]], [c(p) for p in d["poses_inv"]]
    lf = lambda x: x.clone().requires_grad_(True)
    td, rd = [lf(t) for t in tds], [[lf(t) for t in r] for r in rds]
    pp, pi = [lf(p) for p in ps], [lf(p) for p in pis]
    po, go = O.photo_and_geometry_loss(ti, ris, K, td, rd, pp, pi, 1, *flags3, pad)
    tot = up[0] * po + up[1] * go
    if tot.requires_grad:
        tot.backward()
    hint = {"holds": up, "wrong": (up[0] * 0.5 + 0.1, up[1] + 0.3), "plain": None}[state]
    g = lambda x: x.to(device)
    fl = capi.make_flags(*flags3, pad)
    a = dict(ti=g(ti), K=g(K), ris=[g(r) for r in ris], tds=[g(t) for t in tds], rds=[[g(t) for t in r] for r in rds],
             ps=[g(p) for p in ps], pis=[g(p) for p in pis])
    photo, geom, _, ws = capi.photo_geometry_fwd(lib, fl, a["ti"], a["K"], a["ris"], a["tds"], a["rds"], a["ps"], a["pis"], hint=hint)
    po, go = po.detach(), go.detach()
    assert abs(float(photo) - float(po)) < 1e-11 * max(1.0, abs(float(po))), (float(photo), float(po))
    assert abs(float(geom) - float(go)) < 1e-11 * max(1.0, abs(float(go))), (float(geom), float(go))
    t = lambda v: torch.tensor([v], dtype=torch.float64, device=device)
    g_td, g_rd, g_p, g_pi = capi.photo_geometry_bwd(lib, fl, a["ti"], a["K"], a["ris"], a["tds"], a["rds"], a["ps"], a["pis"], ws,
                                                    t(up[0]), t(up[1]))
    z = lambda x: x.grad if x.grad is not None else torch.zeros_like(x)
    assert _rel(g_td[0], z(td[0])) < 1e-9
    for i in range(n_ref):
        assert _rel(g_rd[i][0], z(rd[i][0])) < 1e-9, i
        assert _rel(g_p[i], z(pp[i])) < 1e-9 and _rel(g_pi[i], z(pi[i])) < 1e-9, i
    return float(po), float(go)


def test_the_sweep_covers_what_it_says():
    """The drawn cases contain every state of the speculation, both paddings, 1-3 references, images lower than a strip
    (H < 4) or a tile's interior (H < 14) and narrower / wider than a wave, and both settings of every flag."""
    states = {c[6] for c in CASES}
    assert states == {"holds", "wrong", "plain"} and {c[5] for c in CASES} == {"zeros", "border"}
    assert {c[3] for c in CASES} >= {1, 2} and any(c[0] < 14 for c in CASES) and any(c[1] < 64 for c in CASES) and any(c[1] > 64 for c in CASES)
    for k in range(3):
        assert {c[4][k] for c in CASES} == {0, 1}


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_seeded_shapes_on_the_simulator(case):
    from hostsim import harness
    _run_case(harness.lib(), case, torch.device("cpu"))


def test_some_gate_is_open_and_some_closed_in_the_sweep():
    """(what the two batch sizes of the draw are for: the sweep must see open gates -- positive losses -- AND closed ones)"""
    from hostsim import harness
    lib = harness.lib()
    open_case = next(c for i, c in enumerate(CASES) if i % 4 != 3 and c[4][1] == 0)   # no weight mask, big batch
    closed_case = next(c for i, c in enumerate(CASES) if i % 4 == 3)
    po, _ = _run_case(lib, open_case, torch.device("cpu"))
    pc, gc = _run_case(lib, closed_case, torch.device("cpu"))
    assert po > 0 and pc == 0.0 and gc == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_seeded_shapes_on_the_hardware(case):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from scsfm_hip import _lib
    lib = _lib.get()
    assert lib.path.endswith("libscsfm_hip.so")  # the hipcc build, not the simulator
    _run_case(lib, case, torch.device("cuda:0"))


# ---- the other public callables of the path at seeded odd shapes: compute_smooth_loss (loss_functions.py:132-159), inverse_warp2
# (inverse_warp.py:230-269) and the SSIM module (loss_functions.py:11-42), values and gradients, fp64, through the host mirror
def _aux_cases(n, seed):
    rng = random.Random(seed)
    return [(rng.choice([2, 3, 5, 9, 16, 17, 31, 47]), rng.choice([2, 5, 33, 63, 64, 65, 129, 190]), rng.choice([1, 2, 3]),
             rng.choice(["zeros", "border"]), 2000 + i) for i in range(n)]


AUX_CASES = _aux_cases(8, seed=930)
AUX_IDS = [f"{H}x{W}x{B}-{pad}" for H, W, B, pad, seed in AUX_CASES]


def _run_aux_case(LF, IW, case, device):
    H, W, B, pad, seed = case
    d = synth.make_batch(B, H, W, n_ref=2, seed=seed, depth="smooth")
    c = lambda x: x.double().contiguous()
    lf = lambda x: c(x).clone().requires_grad_(True)
    dv = lambda x: c(x).to(device).clone().requires_grad_(True)
    gen = torch.Generator().manual_seed(seed)
    # compute_smooth_loss over the three frames
    td, rds = [lf(d["tgt_depth"][0])], [[lf(r[0])] for r in d["ref_depths"]]
    so = O.smooth_loss(td, c(d["tgt_img"]), rds, [c(r) for r in d["ref_imgs"]])
    so.backward()
    tdv, rdv = [dv(d["tgt_depth"][0])], [[dv(r[0])] for r in d["ref_depths"]]
    sh = LF.compute_smooth_loss(tdv, c(d["tgt_img"]).to(device), rdv, [c(r).to(device) for r in d["ref_imgs"]])
    sh.backward()
    assert abs(float(sh.detach()) - float(so.detach())) < 1e-11 * max(1.0, abs(float(so.detach())))
    assert _rel(tdv[0].grad, td[0].grad) < 1e-9 and all(_rel(a[0].grad, b[0].grad) < 1e-9 for a, b in zip(rdv, rds))
    # inverse_warp2: four maps and the gradients of a random functional of them
    img, dep, rdep, pose, K = c(d["ref_imgs"][0]), lf(d["tgt_depth"][0]), lf(d["ref_depths"][0][0]), lf(d["poses"][0]), c(d["intrinsics"])
    wts = [torch.rand(B, 3, H, W, generator=gen, dtype=torch.float64), torch.rand(B, 1, H, W, generator=gen, dtype=torch.float64),
           torch.rand(B, 1, H, W, generator=gen, dtype=torch.float64)]
    o = O.inverse_warp2(img, dep, rdep, pose, K, pad)
    (o[0] * wts[0]).sum().add((o[2] * wts[1]).sum()).add((o[3] * wts[2]).sum()).backward()
    depv, rdepv, posev = dv(d["tgt_depth"][0]), dv(d["ref_depths"][0][0]), dv(d["poses"][0])
    h = IW.inverse_warp2(img.to(device), depv, rdepv, posev, K.to(device), pad)
    (h[0] * wts[0].to(device)).sum().add((h[2] * wts[1].to(device)).sum()).add((h[3] * wts[2].to(device)).sum()).backward()
    for a, b in zip(h, o):
        assert a.shape == b.shape and float((a.detach().cpu().double() - b.detach().double()).abs().max()) < 1e-9
    assert _rel(depv.grad, dep.grad) < 1e-9 and _rel(rdepv.grad, rdep.grad) < 1e-9 and _rel(posev.grad, pose.grad) < 1e-9
    # SSIM module (needs a 3x3 window's reflection: H, W >= 2)
    x = torch.rand(B, 3, H, W, generator=gen, dtype=torch.float64)
    y = (x + 0.3 * torch.rand(B, 3, H, W, generator=gen, dtype=torch.float64)).contiguous()
    w = torch.rand(B, 3, H, W, generator=gen, dtype=torch.float64)
    xc, yc = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    oc = O.ssim_map(xc, yc)
    (oc * w).sum().backward()
    xd, yd = x.to(device).requires_grad_(True), y.to(device).requires_grad_(True)
    od = LF.compute_ssim_loss(xd, yd)
    (od * w.to(device)).sum().backward()
    assert float((od.detach().cpu() - oc.detach()).abs().max()) < 1e-10
    assert _rel(xd.grad, xc.grad) < 1e-9 and _rel(yd.grad, yc.grad) < 1e-9


@pytest.mark.parametrize("case", AUX_CASES, ids=AUX_IDS)
def test_seeded_shapes_of_the_other_callables_on_the_simulator(case, monkeypatch):
    import inverse_warp as IW
    import loss_functions as LF
    from hostsim import harness
    from scsfm_hip import _lib, ops
    lib = harness.lib()
    monkeypatch.setattr(_lib, "get", lambda: lib)
    monkeypatch.setattr(ops, "_need_cuda", lambda *a: None)
    _run_aux_case(LF, IW, case, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("case", AUX_CASES, ids=AUX_IDS)
def test_seeded_shapes_of_the_other_callables_on_the_hardware(case):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import inverse_warp as IW
    import loss_functions as LF
    from scsfm_hip import _lib
    assert _lib.get().path.endswith("libscsfm_hip.so")
    _run_aux_case(LF, IW, case, torch.device("cuda:0"))


# ---- tensors as a caller may hand them over: the reference is plain torch and takes any strides (images of a channels_last
# net, a depth map that is one channel of a wider tensor, an expanded intrinsics matrix); the drop-in must give the same
# numbers as for their contiguous copies
def _run_strided_case(LF, device, dtype):
    B, H, W = 3, 47, 67
    d = synth.make_batch(B, H, W, n_ref=2, seed=77, depth="smooth")
    c = lambda x: x.to(device=device, dtype=dtype).contiguous()

    def strided_depth(x):  # [B,1,H,W] as channel 1 of a [B,3,H,W] tensor
        wide = torch.zeros(B, 3, H, W, dtype=dtype, device=device)
        wide[:, 1:2] = c(x)
        return wide[:, 1:2]

    def run(img_f, depth_f, K_f):
        td = [depth_f(d["tgt_depth"][0]).detach().requires_grad_(True)]
        rd = [[depth_f(r[0]).detach().requires_grad_(True)] for r in d["ref_depths"]]
        ps = [c(p).requires_grad_(True) for p in d["poses"]]
        pi = [c(p).requires_grad_(True) for p in d["poses_inv"]]
        tgt, refs = img_f(d["tgt_img"]), [img_f(r) for r in d["ref_imgs"]]
        photo, geom = LF.compute_photo_and_geometry_loss(tgt, refs, K_f(d["intrinsics"]), td, rd, ps, pi, 1, 1, 1, 1, "zeros")
        smooth = LF.compute_smooth_loss(td, tgt, rd, refs)
        (photo + 0.1 * smooth + 0.5 * geom).backward()
        return ([float(photo.detach()), float(geom.detach()), float(smooth.detach())],
                [t.grad.detach().cpu() for t in td + [r[0] for r in rd] + ps + pi])

    v0, g0 = run(c, c, c)
    nhwc = lambda x: c(x).contiguous(memory_format=torch.channels_last)
    expanded_K = lambda K: c(K)[:1].expand(B, 3, 3) if bool((K == K[:1]).all()) else c(K).transpose(1, 2).contiguous().transpose(1, 2)
    v1, g1 = run(nhwc, strided_depth, expanded_K)
    assert not nhwc(d["tgt_img"]).is_contiguous() and not strided_depth(d["tgt_depth"][0]).is_contiguous()
    assert v0 == v1, (v0, v1)  # the same kernels on the same values: bit for bit
    tol = 1e-12 if dtype == torch.float64 else 2e-5  # (fp32: the scatter's atomics may arrive in another order)
    for a, b in zip(g0, g1):
        assert a.shape == b.shape and _rel(b, a) <= tol


def test_strided_inputs_give_the_numbers_of_their_contiguous_copies_on_the_simulator(monkeypatch):
    import loss_functions as LF
    from hostsim import harness
    from scsfm_hip import _lib, ops
    lib = harness.lib()
    monkeypatch.setattr(_lib, "get", lambda: lib)
    monkeypatch.setattr(ops, "_need_cuda", lambda *a: None)
    _run_strided_case(LF, torch.device("cpu"), torch.float64)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_strided_inputs_give_the_numbers_of_their_contiguous_copies_on_the_hardware(dtype):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import loss_functions as LF
    from scsfm_hip import _lib
    assert _lib.get().path.endswith("libscsfm_hip.so")
    _run_strided_case(LF, torch.device("cuda:0"), dtype)
