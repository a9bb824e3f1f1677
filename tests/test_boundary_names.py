"""The drop-in `inverse_warp` module exposes exactly the reference's namespace, and every name runs (CPU: kernels
through tests/hostsim, fp64 against the oracle; the GPU twin is tests/test_gpu_parity.py::test_boundary_functions)."""
import os

import pytest
import torch

import _boundary_checks as BC

REF = "/root/reference/inverse_warp.py"


def test_star_import_exposes_the_reference_namespace():
    ns = {}
    exec("from inverse_warp import *", ns)
    mine = {k for k in ns if not k.startswith("__")}
    assert BC.REFERENCE_NAMES <= mine, BC.REFERENCE_NAMES - mine
    # nothing of this package's plumbing leaks into the star import beyond what the reference's own would bring
    assert mine == BC.REFERENCE_NAMES, mine ^ BC.REFERENCE_NAMES
    if os.path.exists(REF):  # the list above is the reference's
        import importlib.util
        spec = importlib.util.spec_from_file_location("_ref_inverse_warp", REF)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        assert {k for k in vars(ref) if not k.startswith("__")} == BC.REFERENCE_NAMES


def test_every_public_name_runs_and_matches_the_oracle(monkeypatch):
    import inverse_warp as IW
    from hostsim import harness
    from scsfm_hip import _lib, ops
    lib = harness.lib()
    monkeypatch.setattr(_lib, "get", lambda: lib)
    monkeypatch.setattr(ops, "_need_cuda", lambda *a: None)
    BC.run(IW, torch.device("cpu"), torch.float64, 1e-12)
    BC.run(IW, torch.device("cpu"), torch.float32, 2e-6)
    # the remaining names
    IW.set_id_grid(torch.zeros(1, 4, 5))
    assert tuple(IW.pixel_coords.shape) == (1, 3, 4, 5)
    ang = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float64)
    from oracle import scsfm_oracle as O
    assert float((IW.euler2mat(ang) - O.rot_from_euler(ang)).abs().max()) < 1e-14
    assert float((IW.quat2mat(ang) - O.rot_from_quat(ang)).abs().max()) < 1e-14
    with pytest.raises(AssertionError, match="wrong size for depth"):
        IW.pixel2cam(torch.zeros(2, 1, 4, 5), torch.zeros(2, 3, 3))


def test_cam2pixel_rejects_matrices_the_kernels_would_misread():
    """cam2pixel / cam2pixel2 index the rotation as 9 and the translation as 3 contiguous scalars per batch element: a
    [B,3,4] matrix (the shape the reference's docstring names), a [B,3] translation or another batch size must raise
    like check_sizes does -- before anything reaches the library (no device needed for this)."""
    import pytest
    import torch
    import inverse_warp as IW
    cam = torch.zeros(2, 3, 4, 5)
    rot, tr = torch.eye(3).repeat(2, 1, 1), torch.zeros(2, 3, 1)
    for fn in (IW.cam2pixel, IW.cam2pixel2):
        with pytest.raises(AssertionError, match="wrong size for proj_c2p_rot"):
            fn(cam, torch.zeros(2, 3, 4), tr, "zeros")
        with pytest.raises(AssertionError, match="wrong size for proj_c2p_rot"):
            fn(cam, torch.eye(3).repeat(3, 1, 1), tr, "zeros")
        with pytest.raises(AssertionError, match="wrong size for proj_c2p_tr"):
            fn(cam, rot, torch.zeros(2, 3), "zeros")
        with pytest.raises(AssertionError, match="wrong size for cam_coords"):
            fn(torch.zeros(2, 4, 5), rot, tr, "zeros")
