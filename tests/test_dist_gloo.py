"""world_size-2 data-parallel checks on CPU (gloo): batch sharding and the exact-normalisation mode
(one all-reduce of the [n_pairs, 3] raw sums, then the gates / divisions on the global sums).  The
kernels run through tests/hostsim; on the GPU the same capi code path runs with backend nccl (RCCL)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, exact, q):
    for p in (ROOT, os.path.join(ROOT, "sc-sfmlearner-release_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from hostsim import harness
    from oracle import scsfm_oracle as O
    from scsfm_hip import capi, dist as sdist, synth
    torch.set_num_threads(1)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = sdist.init_process_group_from_env(backend="gloo")
    assert (r, lr, w) == (rank, rank, world) and dist.is_initialized()
    lib = harness.lib()
    # B=4 at 72x104: each 2-sample shard (14976 px) is above the photo gate but, with the auto mask,
    # below the geometry gate -- the global batch is above both.  Exact mode must follow the global gate.
    full = synth.make_batch(4, 72, 104, n_ref=1, seed=31, depth="smooth")
    shard = sdist.shard_batch(full, rank, world)
    assert shard["tgt_img"].shape[0] == 2 and torch.equal(shard["tgt_img"], full["tgt_img"][2 * rank:2 * rank + 2])
    flags = capi.make_flags(1, 1, 1, "zeros")
    group = None
    if exact:
        sdist.enable_exact_normalisation()
        group = sdist.exact_group()
        assert group is not None
    tdepth = [shard["tgt_depth"][0]]
    rdepth = [[shard["ref_depths"][0][0]]]
    photo, geom, outs, wss = capi.photo_geometry_fwd(lib, flags, shard["tgt_img"], shard["intrinsics"], shard["ref_imgs"],
                                                     tdepth, rdepth, shard["poses"], shard["poses_inv"], group=group)
    one = torch.ones(1)
    g_td, g_rd, g_p, g_pi = capi.photo_geometry_bwd(lib, flags, shard["tgt_img"], shard["intrinsics"], shard["ref_imgs"],
                                                    tdepth, rdepth, shard["poses"], shard["poses_inv"], wss, one, one)
    # single-process oracle on the concatenated batch (exact) or on the shard (default)
    src = full if exact else shard
    lf = lambda t: t.clone().requires_grad_(True)
    td, rd = [lf(src["tgt_depth"][0])], [[lf(src["ref_depths"][0][0])]]
    ps, pi = [lf(src["poses"][0])], [lf(src["poses_inv"][0])]
    po, go = O.photo_and_geometry_loss(src["tgt_img"], src["ref_imgs"], src["intrinsics"], td, rd, ps, pi, 1, 1, 1, 1,
                                       "zeros")
    (po + go).backward()
    sl = slice(2 * rank, 2 * rank + 2) if exact else slice(None)
    res = {
        "photo": abs(float(photo) - float(po)), "geom": abs(float(geom) - float(go)), "geom_val": float(go),
        "g_td": float((g_td[0] - td[0].grad[sl]).abs().max() / td[0].grad.abs().max()),
        "g_rd": float((g_rd[0][0] - rd[0][0].grad[sl]).abs().max() / rd[0][0].grad.abs().max()),
        "g_pose": float((g_p[0] - ps[0].grad[sl]).abs().max() / ps[0].grad.abs().max()),
        "local_geom_gate_open": bool(float(outs[0, 4]) > 10000) if not exact else None,
    }
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exact", [True, False])
def test_two_rank_data_parallel(exact):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, exact, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in (0, 1):
        r = results[rank]
        assert r["photo"] <= 2e-6 and r["geom"] <= 2e-6, r
        assert r["g_td"] <= 5e-3 and r["g_rd"] <= 5e-3 and r["g_pose"] <= 5e-3, r
    if exact:
        # the global geometry gate is open although each shard alone is below it
        assert results[0]["geom_val"] > 0


def test_env_world_defaults(monkeypatch):
    from scsfm_hip import dist as sdist
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    assert sdist.env_world() == (0, 0, 1)
    assert sdist.exact_group() is None
    with pytest.raises(ValueError):
        sdist.shard_batch({"x": torch.zeros(3, 2)}, 0, 2)
