"""The CPU oracle (oracle/scsfm_oracle.py) must reproduce every golden fixture that
oracle/make_golden.py recorded from the unmodified reference (fp32, torch CPU)."""
import numpy as np
import pytest
import torch

from _util import assert_close_frac, grad_stats, leaf, load_inputs, load_npz
from oracle import scsfm_oracle as O

FLAGS = [(1, 1, 1), (1, 1, 0), (1, 0, 1), (1, 0, 0), (0, 1, 1), (0, 1, 0), (0, 0, 1), (0, 0, 0)]
FULL = {"111_zeros", "110_border", "010_zeros"}


@pytest.mark.parametrize("impl", ["explicit", "aten"])
@pytest.mark.parametrize("name", ["smooth", "iid", "tiny"])
def test_pairwise_loss_and_grads(name, impl):
    d = load_inputs(name)
    gold = load_npz(f"pair_{name}.npz")
    for ssim, mask, auto in FLAGS:
        for pad in ("zeros", "border"):
            key = f"{ssim}{mask}{auto}_{pad}"
            dt, dr, pose = leaf(d["tgt_depth"][0]), leaf(d["ref_depths"][0][0]), leaf(d["poses"][0])
            photo, geom = O.pairwise_loss(d["tgt_img"], d["ref_imgs"][0], dt, dr, pose, d["intrinsics"],
                                          ssim, mask, auto, pad, impl=impl)
            tol = 5e-7 if impl == "aten" else 2e-6  # reduction order depends on the thread count
            assert abs(float(photo.detach()) - float(gold[f"{key}/photo"])) <= tol, key
            assert abs(float(geom.detach()) - float(gold[f"{key}/geom"])) <= tol, key
            L = 1.0 * photo + 0.5 * geom
            if not L.requires_grad:
                assert name == "tiny"
                assert float(gold[f"{key}/photo"]) == 0.0 or float(gold[f"{key}/geom"]) == 0.0
                continue
            L.backward()
            gp = pose.grad if pose.grad is not None else torch.zeros_like(pose)
            assert_close_frac(gp.numpy(), gold[f"{key}/g_pose"], atol=1e-5, rtol=1e-4, what=key + " g_pose")
            for nm, t in (("g_tgt_depth", dt), ("g_ref_depth", dr)):
                g = t.grad if t.grad is not None else torch.zeros_like(t)
                st = gold[f"{key}/{nm}_stats"]
                mine = grad_stats(g)
                assert abs(mine[1] - st[1]) <= 1e-4 * st[1] + 1e-9, (key, nm, mine, st)
                assert abs(mine[2] - st[2]) <= 1e-4 * st[1] + 1e-9, (key, nm, mine, st)
                if key in FULL:
                    assert_close_frac(g.numpy(), gold[f"{key}/{nm}"], atol=2e-8, rtol=1e-4,
                                      max_bad_frac=1e-3, what=f"{key} {nm}")


@pytest.mark.parametrize("impl", ["explicit", "aten"])
@pytest.mark.parametrize("name", ["smooth", "iid"])
def test_warp_maps(name, impl):
    d = load_inputs(name)
    gold = load_npz(f"maps_{name}.npz")
    for pad in ("zeros", "border"):
        w, v, pd, cd = O.inverse_warp2(d["ref_imgs"][0], d["tgt_depth"][0], d["ref_depths"][0][0],
                                       d["poses"][0], d["intrinsics"], pad, impl=impl)
        bad = 0.0 if impl == "aten" else 1e-3
        assert (v.numpy().astype(np.uint8) != gold[f"{pad}/valid_mask"]).mean() <= bad
        assert_close_frac(w.numpy(), gold[f"{pad}/projected_img"], atol=2e-5, max_bad_frac=bad, what="img")
        assert_close_frac(pd.numpy(), gold[f"{pad}/projected_depth"], atol=1e-5, rtol=1e-5, max_bad_frac=bad,
                          what="proj depth")
        assert_close_frac(cd.numpy(), gold[f"{pad}/computed_depth"], atol=1e-5, rtol=1e-5, what="comp depth")


@pytest.mark.parametrize("name", ["smooth", "iid"])
def test_total_loss_and_grads(name):
    d = load_inputs(name)
    gold = load_npz(f"total_{name}.npz")
    for n_scales in (1, 2):
        for ssim, mask, auto, pad in ((1, 1, 1, "zeros"), (1, 1, 0, "border")):
            key = f"s{n_scales}_{ssim}{mask}{auto}_{pad}"
            td = [leaf(x) for x in d["tgt_depth"]]
            rd = [[leaf(x) for x in r] for r in d["ref_depths"]]
            ps = [leaf(p) for p in d["poses"]]
            pi = [leaf(p) for p in d["poses_inv"]]
            photo, geom = O.photo_and_geometry_loss(d["tgt_img"], d["ref_imgs"], d["intrinsics"], td, rd, ps, pi,
                                                    n_scales, ssim, mask, auto, pad)
            smooth = O.smooth_loss(td, d["tgt_img"], rd, d["ref_imgs"])
            assert abs(float(photo.detach()) - float(gold[f"{key}/photo"])) <= 5e-6
            assert abs(float(geom.detach()) - float(gold[f"{key}/geom"])) <= 5e-6
            assert abs(float(smooth.detach()) - float(gold[f"{key}/smooth"])) <= 5e-6
            (1.0 * photo + 0.1 * smooth + 0.5 * geom).backward()
            for s in range(2):
                g = td[s].grad if td[s].grad is not None else torch.zeros_like(td[s])
                assert_close_frac(g.numpy(), gold[f"{key}/g_tgt_depth_s{s}"], atol=5e-8, rtol=1e-4,
                                  max_bad_frac=1e-3, what=f"{key} tgt s{s}")
                for i in range(2):
                    g = rd[i][s].grad if rd[i][s].grad is not None else torch.zeros_like(rd[i][s])
                    assert_close_frac(g.numpy(), gold[f"{key}/g_ref{i}_depth_s{s}"], atol=5e-8, rtol=1e-4,
                                      max_bad_frac=1e-3, what=f"{key} ref{i} s{s}")
            for i in range(2):
                assert_close_frac(ps[i].grad.numpy(), gold[f"{key}/g_pose{i}"], atol=2e-5, rtol=1e-4)
                assert_close_frac(pi[i].grad.numpy(), gold[f"{key}/g_pose_inv{i}"], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("name", ["smooth", "iid"])
def test_smooth_only(name):
    d = load_inputs(name)
    gold = load_npz(f"total_{name}.npz")
    td = [leaf(x) for x in d["tgt_depth"]]
    rd = [[leaf(x) for x in r] for r in d["ref_depths"]]
    loss = O.smooth_loss(td, d["tgt_img"], rd, d["ref_imgs"])
    loss.backward()
    assert abs(float(loss.detach()) - float(gold["smooth_only/loss"])) <= 1e-6
    assert_close_frac(td[0].grad.numpy(), gold["smooth_only/g_tgt_depth"], atol=1e-9, rtol=1e-4)
    for i in range(2):
        assert_close_frac(rd[i][0].grad.numpy(), gold[f"smooth_only/g_ref{i}_depth"], atol=1e-9, rtol=1e-4)


def test_pose_and_errors():
    gold = load_npz("misc.npz")
    vec = torch.from_numpy(gold["pose/vec"])
    r = torch.from_numpy(gold["pose/probe"])
    for mode in ("euler", "quat"):
        v = leaf(vec)
        M = O.pose_vec2mat(v, mode)
        (M * r).sum().backward()
        np.testing.assert_allclose(M.detach().numpy(), gold[f"pose/{mode}/mat"], atol=1e-6)
        np.testing.assert_allclose(v.grad.numpy(), gold[f"pose/{mode}/g_vec"], atol=2e-6)
    for ds in ("kitti", "nyu"):
        out = O.depth_errors(torch.from_numpy(gold[f"errors/{ds}/gt"]), torch.from_numpy(gold[f"errors/{ds}/pred"]), ds)
        np.testing.assert_allclose(out, gold[f"errors/{ds}/out"], rtol=1e-6, atol=1e-7)


def test_oracle_reproduces_the_reference_at_baseline_size():
    """cfg1_reference.npz: the unmodified reference at BASELINE.json configs[1] size (12 x 256 x 832, 2 refs; recorded by
    oracle/make_golden.py: gen_cfg1).  The oracle in its ATen mode calls the same CPU kernels: losses, gradient
    checksums and samples agree to fp32 round-off of the sums."""
    import os
    import numpy as np
    import torch
    from oracle import scsfm_oracle as O
    from scsfm_hip import synth
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_reference.npz"))
    d = synth.make_batch(12, 256, 832, n_ref=2, seed=101, depth="smooth", image="smooth", dataset="kitti")
    chk = np.array([float(d["tgt_img"].double().sum()), float(d["tgt_depth"][0].double().sum()), float(d["poses"][0].double().sum())])
    assert np.allclose(chk, z["input_check"], rtol=1e-12, atol=0), "the seeded inputs differ from the recorded ones"
    leaf = lambda t: t.clone().requires_grad_(True)
    td, rd = [leaf(d["tgt_depth"][0])], [[leaf(r[0])] for r in d["ref_depths"]]
    ps, pi = [leaf(p) for p in d["poses"]], [leaf(p) for p in d["poses_inv"]]
    photo, geom = O.photo_and_geometry_loss(d["tgt_img"], d["ref_imgs"], d["intrinsics"], td, rd, ps, pi, 1, 1, 1, 1, "zeros", impl="aten")
    smooth = O.smooth_loss(td, d["tgt_img"], rd, d["ref_imgs"])
    (1.0 * photo + 0.1 * smooth + 0.5 * geom).backward()
    assert abs(float(photo) - float(z["photo"])) <= 2e-6 and abs(float(geom) - float(z["geom"])) <= 2e-6
    assert abs(float(smooth) - float(z["smooth"])) <= 1e-6
    for name, t in [("g_tgt_depth", td[0])] + [(f"g_ref{i}_depth", rd[i][0]) for i in range(2)]:
        g = t.grad.double().reshape(-1)
        want = z[f"{name}/checks"]
        assert abs(float(g.sum()) - want[0]) <= 1e-5 * want[1] and abs(float(g.abs().sum()) - want[1]) <= 1e-5 * want[1], name
        assert np.allclose(t.grad.reshape(-1)[::997].numpy(), z[f"{name}/sample"], rtol=1e-3, atol=1e-6 * want[4]), name
        # round 6: the denser sample of the reference's own gradients (the comparand of the GPU suite's entry-wise judgement)
        assert np.allclose(t.grad.reshape(-1)[::191].numpy(), z[f"{name}/sample191"], rtol=1e-3, atol=1e-6 * want[4]), name
    for i in range(2):
        assert np.allclose(ps[i].grad.numpy(), z[f"g_pose{i}"], rtol=1e-3, atol=1e-4 * np.abs(z[f"g_pose{i}"]).max())


def test_the_gate_margins_of_the_gradient_judgement_are_frozen():
    """Round-5 review: the entry-wise gradient judgement of tests/test_gpu_parity.py sets aside the entries whose value can
    hinge on a gate decided within fp32 round-off (oracle.pairwise_gate_margins).  Its three margins were widened once
    (round 5: eps_slope_px introduced, with the root-cause analysis of profiles/r05_iid_worst_entries.json); they are
    frozen here -- a further widening is a change of the parity criterion and has to show up as an edit of THIS test --
    together with the shares of entries that must remain judged and the bounds they are held to.

    End of round 6, the one edit since, in the NARROWING direction (round-5 advisor: 5e-4 px was ~8x the coordinate error it
    stands for): the slope-aware margin is two ulp of the image's largest coordinate (oracle.slope_margin_px: 1.2e-4 px at
    W = 832) instead of the constant 5e-4 px -- measured on the hardware over 5e-4 ... 6e-5 px, every worst-entry ratio is
    unchanged (profiles/r06_margin_sensitivity.json) -- and the judged shares rise accordingly: MORE entries judged under
    the same bounds; the constant value margin eps_val was halved (2e-4 -> 1e-4) on the same evidence (ratios identical down to 5e-5).
    Floors of the judged shares: iid 0.79 -> 0.91, elsewhere 0.90 -> 0.93 (what was measured minus two points)."""
    import inspect
    from oracle import scsfm_oracle as O
    sig = inspect.signature(O.pairwise_gate_margins)
    assert {k: sig.parameters[k].default for k in ("eps_px", "eps_val", "eps_slope_px")} == \
        {"eps_px": 2e-3, "eps_val": 1e-4, "eps_slope_px": None}
    # None = two ulp of the largest pixel coordinate in fp32: narrower than round 5's 5e-4 at every BASELINE size
    assert O.slope_margin_px(256, 832) == 2.0 ** -13 and O.slope_margin_px(256, 320) == 2.0 ** -14 and O.slope_margin_px(128, 416) == 2.0 ** -14
    assert O.slope_margin_px(256, 832) < 5e-4 / 4
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("_gpu_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_parity.py"))
    src = open(spec.origin).read()
    ns = {}
    for line in src.split("\n"):
        if line.startswith(("ENTRYWISE_MAX_FACTOR =", "ENTRYWISE_MIN_SHARE =", "ENTRYWISE_QUANTILE_FACTORS =", "HOT_PATH_BAD_SHARE")):
            exec(line, ns)
    assert ns["ENTRYWISE_MAX_FACTOR"] == {"smooth": 4.0, "iid": 4.0, "scene": 4.0}
    assert ns["ENTRYWISE_MIN_SHARE"] == {"smooth": 0.93, "iid": 0.91, "scene": 0.93}
    assert ns["ENTRYWISE_QUANTILE_FACTORS"] == (2.0, 2.0, 2.0, 2.5)
    # the coarser statistical test (HIP fp32 against the oracle's fp32, outlier share): cut from 2e-3 / 8e-3 to a few times the measured
    assert ns["HOT_PATH_BAD_SHARE"] == 3e-4 and ns["HOT_PATH_BAD_SHARE_COARSE_FACTOR"] == 6
