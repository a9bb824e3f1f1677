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


def _last_json(out):
    lines = [l for l in out.stdout.strip().split("\n") if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-4000:]
    return json.loads(lines[-1])


def test_single_gpu_line_has_the_contract_fields():
    env = dict(os.environ, SCSFM_CUDNN_BENCHMARK="0", MIOPEN_FIND_MODE="FAST")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--cpu-seconds", "4"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    r = _last_json(out)
    assert r["metric"].startswith("train images/sec") and r["unit"] == "images/s" and r["n_gpus"] == 1
    assert r["steps"] == 2 and r["warmup"] == 1 and r["higher_is_better"] is True and r["vs_baseline"] is None
    assert r["value"] > 0 and abs(r["value"] - 2 * 2 / (r["ms_per_step"] * 2e-3)) < 1e-2 * r["value"]
    assert r["train"]["final_loss"] == r["train"]["final_loss"]
    assert 0 < r["warp_loss_ms_per_step"] < r["ms_per_step"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["launches_timed"] == 3
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / rf["avg_launch_us"] / 1e3) < 0.01 * rf["achieved"]
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and len(cb["variants"]) >= 3
    assert {v["threads"] for v in cb["variants"]} >= {1} and any(v["anomaly_mode"] for v in cb["variants"])
    lib = r["library"]
    assert lib["path"].endswith("scsfm_hip/libscsfm_hip.so") and lib["abi_version"] == 5 and not lib["env_override"]


def test_two_ranks_sharing_the_gpu_report_a_training_rate():
    env = dict(os.environ, SCSFM_BENCH_SHARED_GPU="1", SCSFM_CUDNN_BENCHMARK="0", MIOPEN_FIND_MODE="FAST")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--cpu-seconds", "0"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    r = _last_json(out)
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["config"]["global_batch"] == 4
    assert r["value"] > 0 and r["train"]["final_loss"] == r["train"]["final_loss"]
    assert "ddp2" in r["config"]["parallelism"]
