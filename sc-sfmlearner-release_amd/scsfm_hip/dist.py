"""Data-parallel helpers: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL on
ROCm, over xGMI; "gloo" in the CPU tests).

The loss path shards over the batch dimension with no collective in the default mode (each rank
normalises its masked means by its own shard -- what any DDP port of the reference does; the
reference itself runs the whole loss on GPU 0, train.py:168-169).  The optional *exact* mode
reproduces the reference's whole-batch semantics (the masked means are ratios of global sums and
the 10000-pixel gate is on the global count, loss_functions.py:123-129): one tiny all-reduce of
the [n_pairs, 3] raw sums between the forward reduction and the backward kernels.  In that mode
every rank returns the GLOBAL photo / geometry losses with gradients w.r.t. its own shard; if the
parameter gradients are subsequently *averaged* (DistributedDataParallel), multiply THOSE TWO terms
by the world size before ``backward()`` so that the averaged gradient equals the single-process
gradient: ``world * (w1 * l1 + w3 * l3) + w2 * l2``.  The smooth term is a per-shard mean whose
average over equally sized shards already is the global one -- scaling it too would weight it
world times too heavily (train.py: train_step does exactly this).
"""
from __future__ import annotations

import os

import torch

_exact_group = None
_exact_enabled = False


def enable_exact_normalisation(group=None):
    """Turn on the global-sum exchange for compute_photo_and_geometry_loss.  ``group=None`` means
    the default (world) process group."""
    import torch.distributed as dist
    global _exact_group, _exact_enabled
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    _exact_group = group if group is not None else dist.group.WORLD
    _exact_enabled = True


def disable_exact_normalisation():
    global _exact_group, _exact_enabled
    _exact_group, _exact_enabled = None, False


def exact_group():
    return _exact_group if _exact_enabled else None


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process if absent)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend, rank, world, device=None, force=False):
    """The one place a process group is created (train.py and bench.py both come through here).  ``nccl`` (= RCCL on
    ROCm) is bound to ``device`` (``device_id``: the communicator is created eagerly on that GPU and collectives never
    have to guess the device); ``gloo`` serves the CPU tests and ranks that share one GPU.  ``force``: also at world
    size 1 (bench.py --force-dist).  Rendezvous defaults to 127.0.0.1 (one node; the container's hostname may not
    resolve).  Returns True when a group was created."""
    import torch.distributed as dist
    if (world <= 1 and not force) or dist.is_initialized():
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(device)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return True


def init_process_group_from_env(backend=None):
    """Initialise torch.distributed for one-process-per-GPU runs launched by torchrun.  Backend defaults to nccl (RCCL)
    when a GPU is visible, gloo otherwise.  Returns (rank, local_rank, world).  The HIP library is loaded here as well
    -- and built, once per node under scsfm_hip.build's file lock, if the tree has none -- so that no rank meets a
    compiler inside its first training step."""
    rank, local_rank, world = env_world()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    device = torch.device("cuda", local_rank) if backend == "nccl" else None
    init_process_group(backend, rank, world, device)
    if torch.cuda.is_available():
        from . import _lib
        _lib.get()
    return rank, local_rank, world


def shard_batch(batch, rank, world):
    """Slice every tensor of a synth.make_batch-style dict along dim 0 into this rank's shard
    (global batch must be divisible by the world size)."""
    def sl(t):
        n = t.shape[0]
        if n % world:
            raise ValueError(f"global batch {n} is not divisible by world size {world}")
        k = n // world
        return t[rank * k:(rank + 1) * k].contiguous()

    def rec(x):
        if isinstance(x, torch.Tensor):
            return sl(x)
        if isinstance(x, (list, tuple)):
            return [rec(y) for y in x]
        return x

    return {k: rec(v) for k, v in batch.items()}


def rank_cpu_block(local_rank, local_world, allowed):
    """This rank's share of the host cores: the ``local_rank``-th of ``local_world`` contiguous, equally sized blocks of
    the sorted CPU ids the job may use (contiguous ids share a NUMA node / L3 on the hosts of the pool; the remainder of
    an uneven division is left to the operating system).  Fewer cores than ranks -> every rank keeps them all."""
    allowed = sorted(allowed)
    if local_world <= 1 or len(allowed) < local_world:
        return allowed
    k = len(allowed) // local_world
    return allowed[local_rank * k:(local_rank + 1) * k]


def pin_rank_to_its_cores(local_rank=None, local_world=None, max_threads=8):
    """One process per GPU on one node: confine this rank -- and the data-loader workers it forks later, which inherit
    the mask -- to its block of the host's cores, and size torch's intra-op pool to it (at most ``max_threads``: the
    host side of a training step is launch-bound, not compute-bound).  Eight unpinned ranks with the default pool (one
    thread per core of a 256-core host each) oversubscribe the host 8x and migrate across NUMA nodes.  Returns the block
    (a list of CPU ids), or None where the platform has no affinity call or nothing was to be done."""
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", 1)))
    if local_world <= 1 or not hasattr(os, "sched_getaffinity"):
        return None
    block = rank_cpu_block(local_rank, local_world, os.sched_getaffinity(0))
    try:
        os.sched_setaffinity(0, set(block))
    except OSError:
        return None
    torch.set_num_threads(max(1, min(max_threads, len(block))))
    return block


def loader_workers_for_rank(requested, block):
    """Data-loader workers of one rank: what was asked for (the reference's -j, default 4), but no more than the rank's
    cores minus one for the training process itself."""
    if block is None:
        return requested
    return max(0, min(requested, len(block) - 1))
