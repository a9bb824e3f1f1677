"""Two ranks of train.train_step under DistributedDataParallel on ONE MI355X (both processes on device 0, gloo
between them: a 1-GPU box cannot host two RCCL ranks), product library, real DispResNet18 / PoseResNet18 --
the multi-process training step of SURVEY.md §8(e) on hardware.  Compared, step by step, with a single-process
emulation of the same data-parallel step (tests/_ddp_steps.py), in the default (per-shard masked means) and
the exact (whole-batch masked means, one all-reduce of the raw sums) modes."""
import socket

import pytest
import torch
import torch.multiprocessing as mp

import _ddp_steps as S

pytestmark = pytest.mark.gpu
B, H, W, STEPS = 2, 128, 416, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("exact", [False, True])
def test_two_ranks_on_one_gpu_train_step(exact):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=S.ddp_worker, args=(r, 2, port, exact, STEPS, B, H, W, "cuda", False, q)) for r in range(2)]
    for p in procs:
        p.start()
    S._paths()
    from scsfm_hip import _lib, config as hip_config
    assert _lib.get().path.endswith("libscsfm_hip.so")
    hip_config.set_weight_hint(S.W1, S.W3)
    ref_losses, ref_snaps = S.emulate(exact, 2, STEPS, B, H, W, torch.device("cuda", 0))
    results = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert len(results[0]["losses"]) == STEPS
    assert results[0]["losses"][0]["rank"][3] > 0  # geometry gate open
    worst = S.compare(results, ref_losses, ref_snaps, exact, loss_tol=1e-4, grad_tol=5e-2, later_loss_tol=2e-2)
    print(f"exact={exact}: worst parameter-update mismatch {worst:.2e} of the update")
