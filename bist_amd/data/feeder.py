"""Host -> HBM input side of the hot path (reference: data/dataset.py:107-127, 146-174).

The reference collates a batch on the host (zero-padded fp32 features ``[B,T,S,C]``, int64 token ids), derives
``temporal_mask`` on the host (dataset.py:79) and then copies everything with blocking ``.cuda()`` calls
(``Batch.move_to_cuda``) inside the training loop (train.py:30).  At B=16 the features are 205 MB per step in fp32
-- 3.3 ms on a 63 GB/s PCIe link if nothing overlaps it, a fifth of this build's training step.

``DeviceFeeder`` keeps that contract (it yields ``bist_amd.data.Batch`` objects on the device) and changes how the
bytes move:

  * two pinned staging slots and two device slots per field, filled by a copy stream, so that the transfer of batch
    i+1 overlaps the compute of batch i; the consumer stream waits on an event, never on the host.  A producer that
    writes its collated batch straight into ``pinned_like()`` buffers skips the staging copy;
  * features cross PCIe in the dtype the producer has (the reference's fp32 ``.npy`` data, or bf16 if stored so) and
    are cast to the compute dtype on the device by ``bist_cast`` on the copy stream -- a host-side cast of 51 M
    elements per step would cost more than the whole training step;
  * ``temporal_mask`` is derived on the device from the features that just landed (``bist_temporal_mask``,
    bit-identical to dataset.py:79), so the host never reduces the feature tensor.

Nothing here is on the timed path of ``bench.py`` (features resident in HBM, as the metric prescribes);
``scripts/bench_feed.py`` measures the PCIe-inclusive rate quoted in DESIGN.md.
"""
from __future__ import annotations

from typing import Iterable, Iterator, Optional

import torch

from .._lib import check, lib
from ..ops import dtype_code
from .batch import PAD, Batch

_FIELDS = ("query", "his", "cap", "trg", "trg_y")


class HostBatch:
    """What a collate function hands over: CPU tensors (ids int64, features fp32 or bf16 ``[B,T,S,C]``)."""

    def __init__(self, query, his, fts, cap, trg, trg_y=None):
        self.query, self.his, self.fts, self.cap, self.trg, self.trg_y = query, his, fts, cap, trg, trg_y


class DeviceFeeder:
    """Iterate device-resident ``Batch`` objects over an iterable of ``HostBatch``.

    feature_dtype: dtype of the features in HBM (bf16 for the throughput path, fp32 for the parity path).
    The Batch yielded for step i is valid until the next one is requested (its slot is then refilled with step i+2,
    after the work queued on the consumer stream up to that point has finished).
    """

    def __init__(self, source: Iterable[HostBatch], device="cuda", feature_dtype: torch.dtype = torch.bfloat16, pad: int = PAD):
        if not torch.cuda.is_available():
            raise RuntimeError("bist_amd.data.DeviceFeeder needs the MI355X (there is no CPU path)")
        self.source, self.device, self.feature_dtype, self.pad = source, torch.device(device), feature_dtype, pad
        # The copy stream must not share a hardware QUEUE with a chain of the split-graph executor: its cast launch sits behind a 4 ms
        # host-to-device copy, and a chain queued behind THAT stands still for as long (measured: 17.5 ms per fed step from a pool stream,
        # 11.6 ms sharing the caption chain's queue, against 8.4 ms resident).
        from .. import functional as _Fn
        self.copy_stream = _Fn.copy_stream() or torch.cuda.Stream(device=self.device)
        self._pinned = [{}, {}]
        self._dev = [{}, {}]
        self._free = [None, None]          # event: the slot's previous consumer is done (recorded when its Batch is replaced)
        self._copied = [None, None]        # event: the slot's last H2D copies have finished READING their host buffers

    # -- staging ---------------------------------------------------------------------------------
    def _slot(self, table, name, like: torch.Tensor, dtype, pinned: bool):
        buf = table.get(name)
        if buf is None or buf.shape != like.shape or buf.dtype != dtype:
            buf = (torch.empty(like.shape, dtype=dtype, pin_memory=True) if pinned
                   else torch.empty(like.shape, dtype=dtype, device=self.device))
            table[name] = buf
        return buf

    @staticmethod
    def pinned_like(shape, dtype) -> torch.Tensor:
        """A page-locked host tensor for producers that collate in place (then no staging copy is made)."""
        return torch.empty(shape, dtype=dtype, pin_memory=True)

    def _stage(self, hb: HostBatch, slot: int):
        """host tensors -> (pinned staging) -> device slot -> compute dtype, all on the copy stream;
        returns (device tensors, ready event)."""
        pin, dev = self._pinned[slot], self._dev[slot]
        out = {}
        if self._copied[slot] is not None:
            # The H2D copies of batch i-2 read this slot's pinned staging buffers asynchronously (they may still be queued behind
            # the consumer's work when the host runs ahead): the HOST must not rewrite those buffers before they are done.
            self._copied[slot].synchronize()
        if self._free[slot] is not None:
            self.copy_stream.wait_event(self._free[slot])          # the device slot is still being read by step i-2
        with torch.cuda.stream(self.copy_stream):
            for name in _FIELDS + ("fts",):
                t = getattr(hb, name)
                if t is None:
                    out[name] = None
                    continue
                if not t.is_pinned():
                    p = self._slot(pin, name, t, t.dtype, True)
                    p.copy_(t)
                    t = p
                d = self._slot(dev, name, t, t.dtype, False)
                d.copy_(t, non_blocking=True)
                if name == "fts" and t.dtype != self.feature_dtype:
                    c = self._slot(dev, "fts_cast", t, self.feature_dtype, False)
                    check(lib.bist_cast(d.data_ptr(), c.data_ptr(), d.numel(), dtype_code(d.dtype), dtype_code(self.feature_dtype),
                                        self.copy_stream.cuda_stream), "bist_cast")
                    d = c
                d._bist_generation = getattr(d, "_bist_generation", 0) + 1      # the slot tensor is reused: tell consumers that cache by identity
                out[name] = d
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        self._copied[slot] = ready
        self._last_slot = slot
        return out, ready

    def host_buffers_reusable(self, batches_ago: Optional[int] = None) -> None:
        """Block the host until staged copies have read their host buffers.  Producers that hand over their OWN pinned tensors and reuse
        them must call this before overwriting them; pageable producers need nothing (their staging copy is guarded inside the feeder).
        Default (None): BOTH slots' copies -- safe whichever buffers the producer is about to rewrite.  batches_ago = k (1 = the latest
        staged batch, 2 = the one before, ...): only that batch's slot (two slots alternate: odd k is the latest slot, even k the other)."""
        last = getattr(self, "_last_slot", None)
        if last is None:
            return                                   # nothing staged yet
        if batches_ago is None:
            for ev in self._copied:
                if ev is not None:
                    ev.synchronize()
            return
        if batches_ago < 1:
            raise ValueError("batches_ago counts back from the latest staged batch: 1, 2, ...")
        ev = self._copied[last if batches_ago % 2 == 1 else last ^ 1]
        if ev is not None:
            ev.synchronize()

    def _batch(self, tensors, ready) -> Batch:
        torch.cuda.current_stream(self.device).wait_event(ready)
        return Batch(tensors["query"], tensors["his"], tensors["fts"], tensors["cap"], tensors["trg"], tensors["trg_y"], pad=self.pad)

    # -- iteration -------------------------------------------------------------------------------
    def __iter__(self) -> Iterator[Batch]:
        it = iter(self.source)
        slot = 0
        try:
            pending = self._stage(next(it), slot)
        except StopIteration:
            return
        while pending is not None:
            cur, cur_slot = pending, slot
            slot ^= 1
            try:
                nxt = next(it)
            except StopIteration:
                nxt = None
            pending = self._stage(nxt, slot) if nxt is not None else None      # batch i+1 starts crossing PCIe now
            b = self._batch(*cur)
            yield b
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))              # consumer of batch i is done with slot cur_slot
            self._free[cur_slot] = done
