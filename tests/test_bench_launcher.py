"""bench.py --gpus N started without a launcher (the driver's N = 1 command form with a larger N) turns itself into a
torch.distributed.run job instead of exiting: the argument handling, without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_torchrun_command_shape():
    import bench
    cmd = bench.torchrun_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]  # the arguments travel unchanged
    # a free port is picked when none is given, and differs from call to call often enough not to be a constant
    ports = {bench.torchrun_command(2, [])[bench.torchrun_command(2, []).index("--master-port") + 1] for _ in range(4)}
    assert all(1024 < int(p) < 65536 for p in ports)


def test_self_launch_runs_the_ranks_and_returns_their_status(monkeypatch):
    """self_launch() with the launcher stubbed by a command that records what it was given: exit status propagates."""
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    assert bench.self_launch(4, ["--gpus", "4", "--e2e", "0"]) == 7
    assert "--nproc-per-node=4" in seen["cmd"] and seen["cmd"][-4:] == ["--gpus", "4", "--e2e", "0"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_ranks_of_the_launched_job_do_not_launch_again():
    """Inside the job RANK / WORLD_SIZE are set: the rank path is taken (here it stops at the missing GPU, AFTER the
    launcher decision), never a second launcher."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29998")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert "re-executing" not in out.stderr
    import torch
    if not torch.cuda.is_available():
        assert out.returncode != 0 and "needs a HIP device" in out.stderr


def test_one_rank_under_a_launcher_is_not_reported_as_n_gpus():
    """Round-5 advisor finding: `torchrun --nproc-per-node=1 bench.py --gpus 4` (or a stale RANK in the environment) used
    to run silently on one GPU.  With RANK / WORLD_SIZE set and WORLD_SIZE = 1 the command is refused."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29997")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE is 1" in out.stderr and "re-executing" not in out.stderr
    assert out.stdout.strip() == ""  # no JSON line
