"""bench.py's contract on hardware, at a reduced size: one JSON line whose `value` is the training rate
(BASELINE.json's metric) with the hot-path figures, roofline and CPU baseline beside it; and the two-rank control
flow (both ranks on the one GPU, gloo) with DistributedDataParallel around the nets -- the path that died in
round 1 -- reporting the number of ranks its collective spanned."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--loss-steps", "3", "--loss-warmup", "1", "--kernel-iters", "2", "--batch", "2",
         "--height", "128", "--width", "416"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


MIOPEN_TOKENS = ("GemmBwdRest", "MIOpen", "miopen")  # what the one crash this retry exists for leaves on stderr
RETRIED = []  # (command, stderr tail) of every retry of this session: the last test of the module reports them


def _run(cmd, env, timeout=900):
    """bench.py in a subprocess.  One retry ONLY for the known crash: on one box of the pool a MIOpen backward solver
    (GemmBwdRest, requested with a null workspace at this reduced shape) took the process down with 'Memory access
    fault ... address (nil)' once in six sessions of round 4 -- inside the nets' backward, not in this library.  The retry
    needs BOTH signatures on stderr (round-5 review: the nil-address line alone names no library, a null-address fault
    of this library's own kernels must not be retried away): a MIOpen token AND the nil-address fault.  Any other death
    by signal is not retried: the caller's assertion on the return code fails with its stderr.  Every retry is
    recorded and surfaces as an xfail in the summary (test_no_bench_subprocess_had_to_be_retried)."""
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    if out.returncode < 0 and "address (nil)" in out.stderr and any(k in out.stderr for k in MIOPEN_TOKENS):
        import warnings
        RETRIED.append((" ".join(cmd[-12:]), out.stderr[-400:]))
        warnings.warn(f"bench.py died with signal {-out.returncode} inside MIOpen (known, null-workspace solver); "
                      f"retrying once.  stderr tail: {out.stderr[-400:]}")
        out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    return out


def _last_json(out):
    lines = [l for l in out.stdout.strip().split("\n") if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-4000:]
    return json.loads(lines[-1])


def test_single_gpu_line_has_the_contract_fields():
    env = dict(os.environ, SCSFM_CUDNN_BENCHMARK="0", MIOPEN_FIND_MODE="FAST")
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--cpu-seconds", "4"], env)
    r = _last_json(out)
    assert r["metric"].startswith("train images/sec") and r["unit"] == "images/s" and r["n_gpus"] == 1
    assert r["steps"] == 2 and r["warmup"] == 1 and r["higher_is_better"] is True and r["vs_baseline"] is None
    assert r["value"] > 0 and abs(r["value"] - 2 * 2 / (r["ms_per_step"] * 2e-3)) < 1e-2 * r["value"]
    assert r["train"]["final_loss"] == r["train"]["final_loss"]
    assert 0 < r["warp_loss_ms_per_step"] < r["ms_per_step"]
    rf = r["roofline"]
    # `frac` is against the HBM peak (the judged figure); what bounds the kernel is named truthfully beside it
    assert rf["bound"] == "valu_issue" and rf["frac_is_against"] == "hbm" and rf["peak"] == 8000.0 and rf["launches_timed"] == 3
    assert rf["smooth_loss_rides_in_the_kernel"] is True and rf["smooth_edge_plane_bytes_per_launch"] == 4 * 2 * 128 * 416 * 3
    assert abs(rf["kernel_own_frac"] - rf["kernel_own_algorithmic_bytes_per_launch"] / rf["avg_launch_us"] / 1e3 / 8000.0) < 1e-3
    if rf["issue_bound_us"] is not None:  # (quoted for the library the static cost was computed on)
        # (the three values are rounded separately -- 0.1 us, 0.01 us, 1e-4 -- and at this reduced shape the launch is a few
        #  microseconds long, so the tolerance carries their rounding steps instead of a flat 1e-3)
        ratio = rf["issue_bound_us"] / rf["avg_launch_us"]
        assert abs(rf["frac_of_issue_bound"] - ratio) < 1e-3 + ratio * (0.05 / rf["issue_bound_us"] + 0.005 / rf["avg_launch_us"])
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / rf["avg_launch_us"] / 1e3) < 0.01 * rf["achieved"]
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and len(cb["variants"]) >= 3
    assert {v["threads"] for v in cb["variants"]} >= {1} and any(v["anomaly_mode"] for v in cb["variants"])
    # pinned, in blocks, with the fastest step beside the median (round 6)
    assert cb["min_ms_per_step"] <= cb["ms_per_step"] and cb["block_medians_ms"] and all(v["omp_proc_bind"] == "close" for v in cb["variants"])
    assert "configs0_end_to_end" in cb  # (None at this reduced shape: the end-to-end CPU step is run at configs[0]'s size only)
    lib = r["library"]
    assert lib["path"].endswith("scsfm_hip/libscsfm_hip.so") and lib["abi_version"] == 9 and not lib["env_override"]
    # the binary names the sources it was built from, and they are the tree's
    assert lib["source_id_in_binary"] == lib["source_sha256_16"] and len(lib["source_sha256_16"]) == 16
    assert r["collective_backend"] is None and r["rccl_ranks"] == 1


def test_hbm_counters_are_measured_in_the_run():
    """`roofline.traffic` MEASURED by the run itself (round 6, --pmc-live): two child runs of the loss-path leg under
    `rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE`, on this box and the loaded library; here at a reduced shape (--pmc-live 2: the
    default takes the leg at configs[1] only).  The counters must be of the right size: the kernel writes its dense and
    scatter planes and the frames' edge planes, and reads every frame at least once (FETCH_SIZE counts between half and all
    of the bytes: profiles/r06_fetch_calibration.json)."""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("no rocprofv3 on this box")
    env = dict(os.environ, SCSFM_CUDNN_BENCHMARK="0", MIOPEN_FIND_MODE="FAST")
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--cpu-seconds", "0", "--e2e", "0", "--other-laws", "0",
                "--pmc-live", "2"], env)
    r = _last_json(out)
    rf = r["roofline"]
    if rf["traffic_is"] is None or not rf["traffic_is"].startswith("measured in this run"):
        # the profiler could not run on this box (the line says why and quotes the committed counters instead): that is the
        # documented fallback, not a defect of the library -- reported, not failed (the logic of the leg itself is held by
        # tests/test_abi.py::test_bench_measures_hbm_counters_through_child_passes_and_says_so)
        pytest.skip(f"live counters unavailable on this box: {rf['traffic_source']}")
    d = rf["traffic_detail"]
    n_px = 2 * 128 * 416
    assert rf["traffic"] == d["fetch_bytes_raw"] + d["write_bytes_raw"] and d["passes"]["FETCH_SIZE"]["launches"] >= 3
    # writes: 8 B/px dense + scatter planes per pair-direction and the 4 B/px edge planes of 3 frames, plus flush atomics
    assert 0.9 * (4 * 8 + 3 * 4) * n_px <= d["write_bytes_raw"] <= 3.0 * (4 * 8 + 3 * 4) * n_px
    # reads: 32 B/px per pair-direction, counted between half and all, some of it served by the caches
    assert 0.25 * 4 * 32 * n_px <= d["fetch_bytes_raw"] <= 1.5 * 4 * 32 * n_px


def test_two_ranks_sharing_the_gpu_report_a_training_rate():
    env = dict(os.environ, SCSFM_BENCH_SHARED_GPU="1", SCSFM_CUDNN_BENCHMARK="0", MIOPEN_FIND_MODE="FAST")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--cpu-seconds", "0"]
    out = _run(cmd, env)
    r = _last_json(out)
    # (gloo between the two ranks that share the GPU: the line says so and does not call them RCCL ranks)
    assert r["n_gpus"] == 2 and r["collective_backend"] == "gloo" and r["collective_ranks"] == 2 and "rccl_ranks" not in r
    assert r["config"]["global_batch"] == 4
    assert r["value"] > 0 and r["train"]["final_loss"] == r["train"]["final_loss"]
    assert "ddp2" in r["config"]["parallelism"]


@pytest.mark.parametrize("exact", [0, 1])
def test_one_rank_over_rccl(exact):
    """The multi-GPU path with an RCCL communicator under it, as far as one GPU can take it: process group on the nccl
    backend (world size 1, device_id), both nets in DistributedDataParallel (bucketed gradient all-reduce on HIP tensors,
    gradient_as_bucket_view), training steps in the default and in the exact mask-normalisation mode (all-reduce of the
    pairs' raw sums on a HIP tensor + re-finalisation), the hot path eager and as a HIP-graph replay (--graph 2)."""
    env = dict(os.environ, SCSFM_CUDNN_BENCHMARK="0", MIOPEN_FIND_MODE="FAST", MASTER_PORT=str(_free_port()))
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--cpu-seconds", "0", "--force-dist", "nccl",
                "--exact", str(exact), "--graph", "2"], env)
    r = _last_json(out)
    assert r["collective_backend"] == "nccl" and r["rccl_ranks"] == 1 and r["n_gpus"] == 1
    assert r["exact_mask_normalisation"] == bool(exact)
    assert r["value"] > 0 and r["train"]["final_loss"] == r["train"]["final_loss"]
    assert "ddp1" in r["config"]["parallelism"]
    assert r["warp_loss"]["graph_error"] is None


def test_gpus_2_without_a_launcher_launches_itself():
    """`python bench.py --gpus 2` started WITHOUT torch.distributed.run -- the form the driver's N = 1 command has -- must
    not die on launch: it re-executes itself under the launcher (127.0.0.1, a free port), rank 0 prints the one JSON
    line and the ranks' exit status comes back.  Two ranks on the box's one GPU (gloo), as above."""
    env = dict(os.environ, SCSFM_BENCH_SHARED_GPU="1", SCSFM_CUDNN_BENCHMARK="0", MIOPEN_FIND_MODE="FAST")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--e2e", "1", *SMALL, "--cpu-seconds", "0"], env)
    r = _last_json(out)
    assert r["n_gpus"] == 2 and r["collective_ranks"] == 2 and r["config"]["global_batch"] == 4
    assert r["steps"] == 2 and r["value"] > 0 and "ddp2" in r["config"]["parallelism"]
    assert "re-executing as" in out.stderr


def test_eight_ranks_on_the_one_gpu_run_the_drivers_scaling_command():
    """The driver's 8-GPU command -- `python bench.py --gpus 8 ...`, no launcher -- as far as a 1-GPU box can take it: eight
    ranks that share the GPU (SCSFM_BENCH_SHARED_GPU=1: gloo between them), self-launch, rendezvous on 127.0.0.1, the
    nets in DistributedDataParallel, rank 0's one JSON line with n_gpus = collective_ranks = 8 and a global batch of
    8 x per-rank.  No 8-GPU node was available to any round: this is what stands in for it."""
    env = dict(os.environ, SCSFM_BENCH_SHARED_GPU="1", SCSFM_CUDNN_BENCHMARK="0", MIOPEN_FIND_MODE="FAST", OMP_NUM_THREADS="4")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    small = ["--steps", "2", "--warmup", "1", "--loss-steps", "2", "--loss-warmup", "1", "--kernel-iters", "1", "--batch", "1",
             "--height", "128", "--width", "416", "--other-laws", "0"]
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", *small, "--cpu-seconds", "0"], env, timeout=1500)
    r = _last_json(out)
    assert r["n_gpus"] == 8 and r["collective_ranks"] == 8 and r["collective_backend"] == "gloo" and "rccl_ranks" not in r
    assert r["config"]["global_batch"] == 8 and "ddp8" in r["config"]["parallelism"] and r["scaling"] == "weak"
    assert r["steps"] == 2 and r["value"] > 0 and r["train"]["final_loss"] == r["train"]["final_loss"]
    assert "cpu_baseline" not in r  # rank 0 at N = 1 only
    assert "re-executing as" in out.stderr


def test_no_bench_subprocess_had_to_be_retried():
    """Last in the module: a retry of the known MIOpen crash does not fail the suite, but it must be visible."""
    if RETRIED:
        pytest.xfail(f"{len(RETRIED)} bench subprocess(es) died inside MIOpen and were retried once: {RETRIED}")
