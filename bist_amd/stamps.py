"""Development aid: device-side timestamps at named points of the streams of a step (scripts/stamp_step.py).

A profiler trace adds ~2 us per dispatch and its timeline of a replayed hipGraph is not the un-profiled one; what bounds the
replayed training step is the longest chain of DEPENDENT launches across its three streams, which only a timeline taken inside the
replay shows.  ``mark(name)`` queues a one-thread launch (bist_dev_timestamp) on the current stream that writes the device's 100 MHz
wall clock into the next slot; ``through(x, name)`` does so in the forward pass and -- as an identity autograd node -- again when the
gradient comes back through that point.  Captured with the step, every replay refreshes the slots.  Off (the default) both are a
single attribute test; nothing in the product path enables them.
"""
from __future__ import annotations

import torch

from ._lib import check, lib
from .ops import _stream

ENABLED = False
BUF = None           # int64 [capacity] device buffer of clock values
NAMES = []           # slot -> (name, stream id)
LAYER = 0            # layer index the layer loop is in (names of per-layer points)


def enable(capacity: int = 16384, device="cuda") -> None:
    global ENABLED, BUF
    BUF = torch.zeros(capacity, dtype=torch.int64, device=device)
    NAMES.clear()
    ENABLED = True


def disable() -> None:
    global ENABLED
    ENABLED = False


def mark(name: str) -> None:
    if not ENABLED:
        return
    slot = len(NAMES)
    if slot >= BUF.numel():
        return
    NAMES.append((name, torch.cuda.current_stream().cuda_stream))
    check(lib.bist_dev_timestamp(BUF.data_ptr() + 8 * slot, _stream()), "bist_dev_timestamp")


class _Through(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, name):
        ctx.name = name
        mark(name + " f")
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        mark(ctx.name + " b")
        return g, None


def lmark(name: str) -> None:
    if ENABLED:
        mark("L%d %s" % (LAYER, name))


def through(x, name: str, layer: bool = False):
    """x unchanged; a timestamp now and, under autograd, one more when x's gradient passes.  layer: prefix the name with the layer index."""
    if not ENABLED:
        return x
    if layer:
        name = "L%d %s" % (LAYER, name)
    if torch.is_grad_enabled() and torch.is_tensor(x) and x.requires_grad:
        return _Through.apply(x, name)
    mark(name + " f")
    return x


def read(first_slot: int = 0):
    """[(name, stream, microseconds since the earliest stamp)] of the slots written since the buffer was last cleared, in time order."""
    torch.cuda.synchronize()
    vals = BUF.cpu().tolist()
    rows = [(NAMES[i][0], NAMES[i][1], vals[i]) for i in range(first_slot, len(NAMES)) if vals[i] != 0]
    if not rows:
        return []
    t0 = min(v for _, _, v in rows)
    return sorted(((n, s, (v - t0) / 100.0) for n, s, v in rows), key=lambda r: r[2])
