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
    # ABI 8: the smooth loss's depth gradients added by the pair backward's combining pass
    "scsfm_pairs_bwd_smooth_f32", "scsfm_pairs_bwd_smooth_f64",
    "scsfm_smooth_multi_fwd_step_f32", "scsfm_smooth_multi_fwd_step_f64",
    # ABI 9: the smooth loss riding in the speculative forward (scsfm_pair_desc::smooth_ws); the step total in its finalize launch
    "scsfm_pairs_fwd_step_f32", "scsfm_pairs_fwd_step_f64",
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


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="no hipcc on this machine")
def test_entry_points_reject_sizes_their_32_bit_offsets_cannot_address():
    """csrc/scsfm_common.h: dims_ok.  The kernels address the three colour planes of one image with 32-bit byte
    offsets and put (pair, batch element) on the grid's z axis: an image of 3 * H * W * sizeof(T) >= 4 GiB or a batch
    beyond 8191 is refused with SCSFM_ERR_ARG before anything is launched (so this runs without a GPU; the pointers only
    have to be non-null)."""
    import ctypes
    from scsfm_hip import build, capi
    lib = _lib.CLib(build.build(verbose=False))
    buf = ctypes.create_string_buffer(4096)
    p = ctypes.addressof(buf)
    big_h, big_w = 20000, 20000          # 3 * 4e8 * 4 B = 4.8 GB per image in fp32
    assert 3 * big_h * big_w * 4 >= 2 ** 32
    d = (capi.PairDesc * 1)()
    for f in ("tgt_img", "ref_img", "tgt_depth", "ref_depth", "pose", "ws", "out", "total"):
        setattr(d[0], f, p)
    call = lambda name, *a: lib._fn[name](*a)
    assert call("scsfm_pairs_fwd_f32", 1, ctypes.addressof(d), 1, big_h, big_w, p, 7, 1.0, 0.5, None) == -1
    assert call("scsfm_pairs_fwd_f32", 1, ctypes.addressof(d), 8192, 16, 16, p, 7, 1.0, 0.5, None) == -1   # grid z
    assert call("scsfm_pairs_fwd_f64", 1, ctypes.addressof(d), 1, 16384, 16384, p, 7, 1.0, 0.5, None) == -1  # fp64: 6.4 GB
    assert call("scsfm_warp_fwd_f32", 1, big_h, big_w, p, p, p, p, p, 0, p, p, p, p, p, None) == -1
    ptrs = (ctypes.c_void_p * 1)(p)
    assert call("scsfm_smooth_multi_fwd_f32", 1, ptrs, ptrs, 1, big_h, big_w, p, None, p, None) == -1
    assert call("scsfm_ssim_fwd_f32", 1, big_h, big_w, p, p, p, None) == -1


def test_bench_quotes_hbm_counters_only_for_the_library_they_were_collected_on(tmp_path, monkeypatch):
    """bench.py: roofline.traffic comes from profiles/pmc_latest.json, which names the library (source id) the
    rocprofv3 --pmc passes ran on; a different loaded library gets None and the reason, not stale counters."""
    import argparse
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    key = "scsfm::pair_fwd_spec_kernel<float, true, 7u, false, false>|gz48"
    json.dump({"_library_source_id": "aaaa", key: {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 24.0}}, open(prof / "pmc_latest.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    a = argparse.Namespace(batch=12, height=256, width=832, n_ref=2, depth="smooth")
    t, why = bench.pmc_traffic(a, 4, "aaaa")
    # the two counters separately (round 6), raw; and calibrated where profiles/r06_fetch_calibration.json is there
    assert t["fetch_bytes_raw"] == 1000 * 1024 and t["write_bytes_raw"] == 24 * 1024 and "QUOTED" in why and "fetch_bytes" not in t
    json.dump({"kernels": {"read_kernel<unsigned int>": {"FETCH_SIZE_over_known": 0.5}, "gather_b64_kernel": {"FETCH_SIZE_over_known": 1.0},
                           "write_kernel<unsigned int>": {"WRITE_SIZE_over_known": 1.0}, "atomic_f32_kernel": {"WRITE_SIZE_over_known": 1.0}}},
              open(prof / "r06_fetch_calibration.json", "w"))
    t, _ = bench.pmc_traffic(a, 4, "aaaa")
    assert t["fetch_bytes_range"] == [1000 * 1024, 2000 * 1024] and t["write_bytes"] == 24 * 1024
    got, why = bench.pmc_traffic(a, 4, "bbbb")
    assert got is None and "aaaa" in why and "bbbb" in why
    a.depth = "iid"
    assert bench.pmc_traffic(a, 4, "aaaa")[0] is None


def test_bench_measures_hbm_counters_through_child_passes_and_says_so(tmp_path, monkeypatch):
    """bench.py --pmc-live: roofline.traffic is MEASURED by two child runs of the loss-path leg under `rocprofv3 --pmc`, one
    counter per pass; every failure (no tool, a pass that fails, a database without the kernel, a profiler already around
    this process) gives None and the reason, so that the caller quotes the committed counters instead.  The profiler is a
    stand-in here that writes the database layout pmc_summary.py reads (pmc_events / kernels)."""
    import argparse
    import stat
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    for k in [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_"))]:
        monkeypatch.delenv(k)
    monkeypatch.setenv("RANK", "0")  # (a launcher's variables must not reach the children)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))  # (no calibration file: raw counters only)
    a = argparse.Namespace(batch=2, height=128, width=416, n_ref=2, depth="smooth", dataset="kitti")
    bindir = tmp_path / "bin"
    bindir.mkdir()
    monkeypatch.setenv("PATH", str(bindir))
    assert bench.pmc_traffic_live(a, 4, 30.0) == (None, "no rocprofv3 on PATH")
    fake = bindir / "rocprofv3"
    fake.write_text(textwrap.dedent(f"""\
        #!{sys.executable}
        import os, sqlite3, sys
        a = sys.argv[1:]
        c, d, o, child = a[a.index("--pmc") + 1], a[a.index("-d") + 1], a[a.index("-o") + 1], a[a.index("--") + 1:]
        assert "--kernel-trace" in a and len([x for x in a[:a.index("--")] if x.isupper()]) == 1  # one counter, no other domain
        assert child[1].endswith("bench.py") and child[child.index("--pmc-live") + 1] == "0" and child[child.index("--e2e") + 1] == "0"
        assert child[child.index("--batch") + 1] == "2" and child[child.index("--width") + 1] == "416" and "RANK" not in os.environ
        mode = os.environ.get("FAKE_MODE", "ok")
        if mode == "rc":
            sys.exit(3)
        db = sqlite3.connect(os.path.join(d, o + "_results.db"))
        db.execute("create table kernels(dispatch_id, name, grid_z)")
        db.execute("create table pmc_events(dispatch_id, counter_name, counter_value, duration)")
        name = "void scsfm::pair_fwd_spec_kernel<float, true, 7u, false, false>(scsfm::PairBatch<float>)"
        rows = [(1, name, 8, 100.0), (2, name, 8, 300.0), (3, name, 4, 7000.0), (4, "void scsfm::pairs_combine_kernel<float>()", 8, 9000.0)]
        for i, n, gz, v in rows:
            if mode == "absent" and "spec" in n:
                continue
            db.execute("insert into kernels values (?,?,?)", (i, n, gz))
            db.execute("insert into pmc_events values (?,?,?,?)", (i, c, v * (2 if c == "WRITE_SIZE" else 1), 5000.0))
        db.commit()
        """))
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    t, why = bench.pmc_traffic_live(a, 4, 30.0)
    # the average over the dispatches of THIS workload's grid (z = pairs x batch = 8), KiB -> bytes; the other grid and kernel ignored
    assert t["fetch_bytes_raw"] == 200 * 1024 and t["write_bytes_raw"] == 400 * 1024 and why.startswith("measured in this run")
    assert t["passes"]["FETCH_SIZE"]["launches"] == 2 and t["passes"]["WRITE_SIZE"]["avg_launch_us_under_the_counter"] == 5.0
    monkeypatch.setenv("FAKE_MODE", "rc")
    t, why = bench.pmc_traffic_live(a, 4, 30.0)
    assert t is None and "rc 3" in why
    monkeypatch.setenv("FAKE_MODE", "absent")
    t, why = bench.pmc_traffic_live(a, 4, 30.0)
    assert t is None and "no launch of the dominant kernel" in why
    monkeypatch.setenv("FAKE_MODE", "ok")
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    assert bench.pmc_traffic_live(a, 4, 30.0) == (None, "this process already runs under a profiler")


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


def test_tuning_flags_change_the_source_id():
    """build(extra=...) binaries carry an id of their own (round-4 advisor finding: bench.py's counter check took a
    tuning variant for the default library)."""
    from scsfm_hip import build
    base = build.source_id()
    assert build.source_id(()) == base
    v = build.source_id(("-DSCSFM_WIN_W=80",))
    assert v != base and len(v) == 16 and build.source_id(("-DSCSFM_WIN_W=80",)) == v
    assert build.source_id(("-DSCSFM_WIN_W=64",)) != v
