"""The C ABI: every function include/scsfm_hip.h declares must be exported, with the same name, by
the product library (hipcc, gfx950) and by the host-simulation build used in CPU-only CI.  No
compute calls here -- loading and symbol resolution only."""
import os
import shutil

import pytest

from scsfm_hip import _lib

EXPECTED = {
    "scsfm_abi_version", "scsfm_source_id", "scsfm_profile_begin", "scsfm_profile_end",
    "scsfm_step_total_f32", "scsfm_step_total_f64", "scsfm_step_weights_f32", "scsfm_step_weights_f64",
    "scsfm_pair_ws_bytes", "scsfm_pair_bwd_scratch_bytes", "scsfm_pair_fwd_f32", "scsfm_pair_bwd_f32", "scsfm_pair_refinalize_f32", "scsfm_pair_fwd_spec_f32",
    "scsfm_pair_fwd_spec_f64",
    "scsfm_pair_fwd_f64", "scsfm_pair_bwd_f64", "scsfm_pair_refinalize_f64",
    "scsfm_pairs_fwd_f32", "scsfm_pairs_bwd_f32", "scsfm_pairs_fwd_f64", "scsfm_pairs_bwd_f64",
    "scsfm_smooth_multi_fwd_f32", "scsfm_smooth_multi_bwd_f32", "scsfm_smooth_multi_fwd_f64",
    "scsfm_smooth_multi_bwd_f64",
    "scsfm_warp_ws_bytes", "scsfm_warp_fwd_f32", "scsfm_warp_bwd_f32", "scsfm_warp_fwd_f64", "scsfm_warp_bwd_f64",
    "scsfm_pose_vec2mat_fwd_f32", "scsfm_pose_vec2mat_bwd_f32", "scsfm_pose_vec2mat_fwd_f64",
    "scsfm_pose_vec2mat_bwd_f64",
    "scsfm_smooth_ws_bytes", "scsfm_smooth_fwd_f32", "scsfm_smooth_bwd_f32", "scsfm_smooth_fwd_f64",
    "scsfm_smooth_bwd_f64",
    "scsfm_ssim_fwd_f32", "scsfm_ssim_bwd_f32", "scsfm_ssim_fwd_f64", "scsfm_ssim_bwd_f64",
    "scsfm_masked_mean_ws_bytes", "scsfm_masked_mean_fwd_f32", "scsfm_masked_mean_bwd_f32",
    "scsfm_masked_mean_fwd_f64", "scsfm_masked_mean_bwd_f64", "scsfm_augment_u8_f32",
    "scsfm_pixel2cam_fwd_f32", "scsfm_pixel2cam_bwd_f32", "scsfm_pixel2cam_fwd_f64", "scsfm_pixel2cam_bwd_f64",
    "scsfm_cam2pixel_fwd_f32", "scsfm_cam2pixel_bwd_f32", "scsfm_cam2pixel_fwd_f64", "scsfm_cam2pixel_bwd_f64",
    # ABI 7: gradients of the data inputs (images, intrinsics, a floating-point mask)
    "scsfm_pairs_bwd_inputs_f32", "scsfm_pairs_bwd_inputs_f64", "scsfm_warp_bwd_inputs_f32", "scsfm_warp_bwd_inputs_f64",
    "scsfm_pixel2cam_bwd_intrinsics_f32", "scsfm_pixel2cam_bwd_intrinsics_f64",
    "scsfm_masked_mean_bwd_mask_f32", "scsfm_masked_mean_bwd_mask_f64",
    "scsfm_smooth_multi_bwd_images_f32", "scsfm_smooth_multi_bwd_images_f64",
}


def test_header_declares_the_expected_entry_points():
    decls = _lib.parse_header()
    assert set(decls) == EXPECTED
    # raw pointers and sizes only: the parser maps every argument to int / unsigned / size_t / void*
    restype, argtypes = decls["scsfm_pair_fwd_f32"]
    assert len(argtypes) == 13
    assert len(decls["scsfm_pair_bwd_f32"][1]) == 18


def test_hostsim_build_exports_every_symbol():
    from hostsim import harness
    lib = harness.lib()
    assert set(lib.decls) == EXPECTED
    assert lib.size("scsfm_pair_ws_bytes", 2, 32, 64) > 0
    assert lib.size("scsfm_pair_ws_bytes", 0, 32, 64) == 0


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="no hipcc on this machine")
def test_product_library_builds_and_exports_every_symbol():
    from scsfm_hip import build
    path = build.build(verbose=False)
    assert path.endswith("libscsfm_hip.so") and os.path.exists(path)
    lib = _lib.CLib(path)  # resolves every declared symbol or raises
    assert set(lib.decls) == EXPECTED
    # argument validation happens before any launch, so it can be exercised without a GPU
    assert lib._dll.scsfm_pair_fwd_f32(0, 8, 8, None, None, None, None, None, None, 0, None, None, None) == -1
    assert lib._dll.scsfm_smooth_fwd_f32(1, 1, 8, None, None, None, None, None) == -1
    assert lib.size("scsfm_smooth_ws_bytes", 12, 256, 832) > 0


def test_product_loader_never_points_at_the_simulator():
    assert _lib.LIB_PATH.endswith(os.path.join("scsfm_hip", "libscsfm_hip.so"))
    src = open(_lib.__file__).read()
    assert "hostsim" not in src and "oracle" not in src


def test_graph_capture_needs_a_device():
    """scsfm_hip.graphs has no CPU mode either."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    from scsfm_hip.graphs import GraphedStep
    with pytest.raises(RuntimeError):
        GraphedStep(lambda: None)
