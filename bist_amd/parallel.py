"""Batch-data-parallel host logic of the training path: the per-clip forward shards by
clip with no data-path collective; the only exchange is one sum all-reduce of the flat gradient buffer
(RCCL over xGMI on the GPUs, any torch.distributed backend in tests) followed by a 1/world scaling that is
folded into the Adam kernel's ``grad_scale``."""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def clip_range(rank: int, world: int, global_batch: int) -> Tuple[int, int]:
    """Clips [lo, hi) of a global batch owned by ``rank`` (contiguous, sizes differ by at most one)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def exchange_gradients(flat_grad: torch.Tensor, group=None, bucket_elems: int = 0) -> float:
    """Sum ``flat_grad`` over all ranks in place; returns the scale (1/world) the optimiser must apply.

    ``bucket_elems`` > 0 splits the buffer into equal chunks issued back to back (xGMI is point-to-point:
    a few large messages keep all 7 links busy; per-tensor collectives do not)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world == 1:
        return 1.0
    if bucket_elems <= 0 or bucket_elems >= flat_grad.numel():
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    else:
        works = [dist.all_reduce(flat_grad[o:o + bucket_elems], op=dist.ReduceOp.SUM, group=group, async_op=True)
                 for o in range(0, flat_grad.numel(), bucket_elems)]
        for w in works:
            w.wait()
    return 1.0 / world


def chunk_bounds(n: int, chunks: int, align: int = 1) -> List[Tuple[int, int]]:
    """[lo, hi) element ranges that cut n elements into at most `chunks` pieces whose starts are multiples of `align`."""
    step = -(-n // max(1, chunks))
    step = -(-step // align) * align
    return [(lo, min(n, lo + step)) for lo in range(0, n, step)]


def exchange_gradients_async(flat_grad: torch.Tensor, bounds: List[Tuple[int, int]], group=None):
    """Issue one summing all-reduce per [lo, hi) piece of ``flat_grad`` (in place, in order) and return the work handles;
    ``handle.wait()`` orders the caller's stream after that piece (RCCL) or blocks the host (gloo)."""
    return [dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True) for lo, hi in bounds]


def noam_rate(step: int, d_model: int, factor: float = 1.0, warmup: int = 4000) -> float:
    """NoamOpt.rate (model/optimize.py:28-34)."""
    return factor * (d_model ** -0.5 * min(step ** -0.5, step * warmup ** -1.5))
