"""train.train_step under DistributedDataParallel at world 2 on CPU (gloo, kernels through tests/hostsim) with
the real DispResNet18 / PoseResNet18: the step that crashed in round 1 (DDP's default buffer broadcast rewrote
BatchNorm statistics that earlier forwards of the same step had saved for backward), and the gradient semantics
of both data-parallel modes (per-shard means; exact whole-batch masked means with the smooth term unscaled)."""
import socket

import pytest
import torch.multiprocessing as mp

import _ddp_steps as S

B, H, W, STEPS = 3, 64, 96, 2  # global 2 x 3 x 64 x 96 = 36864 pixels: above the 10000-pixel gates in exact mode


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("exact", [False, True])
def test_two_rank_train_step_matches_the_single_process_emulation(exact, monkeypatch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=S.ddp_worker, args=(r, 2, port, exact, STEPS, B, H, W, "cpu", True, q)) for r in range(2)]
    for p in procs:
        p.start()
    # the emulation in this process, on the same host-simulation library
    S._paths()
    import torch
    from hostsim import harness
    from scsfm_hip import _lib, config as hip_config, ops
    lib = harness.lib()
    monkeypatch.setattr(_lib, "get", lambda: lib)
    monkeypatch.setattr(ops, "_need_cuda", lambda *a: None)
    hip_config.set_weight_hint(S.W1, S.W3)
    torch.set_num_threads(2)
    ref_losses, ref_snaps = S.emulate(exact, 2, STEPS, B, H, W, torch.device("cpu"))
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    if exact:  # the configuration is meant to exercise the geometry term too
        assert results[0]["losses"][0]["rank"][3] > 0
    worst = S.compare(results, ref_losses, ref_snaps, exact, loss_tol=2e-5, grad_tol=2e-2)
    print(f"exact={exact}: worst parameter-update mismatch {worst:.2e} of the update")
