"""HIP-graph capture of a static-shape step.

The loss path of a training step is ~25 short launches (the library's kernels plus the scalar arithmetic
and gradient accumulation of autograd); once the kernels are fast the step is bound by launching them
from Python (~0.5 ms, measured with tools/host_overhead.py).  Every launch of this package goes to torch's
current stream and nothing synchronises or reads back, so the whole step -- forward, ``backward()``
included -- can be captured once into a HIP graph (``torch.cuda.CUDAGraph``) and replayed with a single
launch.  The usual rules of graph capture apply: fixed shapes, inputs updated in place in the same
tensors (``tensor.copy_``), results read from the same output tensors after ``replay()``.
"""
from __future__ import annotations

import torch


class GraphedStep:
    """``step = GraphedStep(fn)`` runs ``fn()`` a few times eagerly on a side stream (allocator / autotuner
    warm-up), captures one more call, and keeps what it returned; ``step.replay()`` relaunches the
    captured work and returns those same (now refreshed) tensors.

    ``fn`` must leave gradients in ``.grad`` of tensors whose ``.grad`` was None before the call (set
    them to None first, as ``zero_grad(set_to_none=True)`` does): the captured backward then owns the
    gradient buffers and every replay rewrites them in place."""

    def __init__(self, fn, warmup: int = 3):
        if not torch.cuda.is_available():
            raise RuntimeError("scsfm_hip.graphs: HIP graphs need a HIP device")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn()

    def replay(self):
        self.graph.replay()
        return self.outputs
