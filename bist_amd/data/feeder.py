"""Host -> HBM input side of the hot path (reference: data/dataset.py:107-127, 146-174).

The reference collates a batch on the host (zero-padded fp32 features ``[B,T,S,C]``, int64 token ids), derives
``temporal_mask`` on the host (dataset.py:79) and then copies everything with blocking ``.cuda()`` calls
(``Batch.move_to_cuda``) inside the training loop (train.py:30).  At B=16 the features are 205 MB per step in fp32
-- 3.3 ms on a 63 GB/s PCIe link if nothing overlaps it, a fifth of this build's training step.

``DeviceFeeder`` keeps that contract (it yields ``bist_amd.data.Batch`` objects on the device) and changes how the
bytes move:

  * three pinned staging slots and three device slots per field, filled by a copy stream, so that the transfer of batch
    i+1 overlaps the compute of batch i.  The copy stream carries NOTHING BUT COPIES: no wait ahead of them (the host
    waits for the slot's previous reader, two steps back), no event or kernel behind them (the host waits for the copy
    stream before it yields the batch: the copy was queued a step earlier) -- a packet that sits on the copy stream's
    hardware queue while a 3.6 ms copy runs makes the chain of the replayed step that shares that queue's dispatch pipe
    pay a threefold launch gap for as long (the PCIe-inclusive step went from 10.7 to 8.2 ms, 7.97
    resident).  A producer that writes its collated batch straight into ``pinned_like()`` buffers skips the staging copy;
  * features cross PCIe in the dtype the producer has (the reference's fp32 ``.npy`` data, or bf16 if stored so) and
    are cast to the compute dtype on the device by ``bist_cast`` on the CONSUMER's stream, at the head of the step that
    reads them -- a host-side cast of 51 M elements per step would cost more than the whole training step;
  * ``temporal_mask`` is derived on the device from the features that just landed (``bist_temporal_mask``,
    bit-identical to dataset.py:79), so the host never reduces the feature tensor.

Nothing here is on the timed path of ``bench.py``'s ``value`` (features resident in HBM, as the metric prescribes); its ``fed``
side line is the PCIe-inclusive rate through this class.
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


SLOTS = max(2, int(__import__("os").environ.get("BIST_FEEDER_SLOTS", "3")))      # device / pinned staging slots (2: the round-3 double buffer)


class DeviceFeeder:
    """Iterate device-resident ``Batch`` objects over an iterable of ``HostBatch``.

    feature_dtype: dtype of the features in HBM (bf16 for the throughput path, fp32 for the parity path).
    The Batch yielded for step i is valid until the next one is requested (its slot is refilled with step i+3 -- three slots --
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
        # THREE slots: the slot refilled for step i+1 was read by step i-2, which is over when the host gets here (it runs at most one step
        # ahead: it waits for that step below).  With two slots the copy stream stood in a wait for step i-1 -- a packet resident on its
        # queue for the whole step, and the chain that shares the queue's dispatch pipe paid the threefold launch gap for as long: the
        # fed step took 10.7 ms where the copy itself (3.6 ms, a step ahead) costs 0.1.
        self.slots = SLOTS
        self._pinned = [{} for _ in range(self.slots)]
        self._dev = [{} for _ in range(self.slots)]
        self._free = [None] * self.slots          # event: the slot's previous consumer is done (recorded when its Batch is replaced)
        self._in_flight = False                   # the batch staged last may still be crossing PCIe

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
        """host tensors -> (pinned staging) -> device slot, on the copy stream: COPIES ONLY.  Nothing is queued behind them -- an event or
        a kernel behind a 3.6 ms host-to-device copy is a packet that sits on the copy stream's hardware queue for as long, and the chain of
        the replayed step that shares that queue's dispatch pipe pays the threefold launch gap meanwhile (1.1-1.4 ms per step, measured:
        scripts/probe_h2d_interference.py); _finish() completes the batch one step later.  Returns the device tensors (features as copied)."""
        pin, dev = self._pinned[slot], self._dev[slot]
        out = {}
        if self._free[slot] is not None:
            # the device slot was read by the step before the last (three slots): the HOST waits for it -- normally over already -- so the
            # copy stream never holds a wait either; this is also what keeps the host from running more than a step ahead.  (The slot's
            # pinned staging buffers were read by copies that _finish() has waited for.)
            self._free[slot].synchronize()
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
                out[name] = d
        self._in_flight = True
        self._last_slot = slot
        return out, slot

    def _finish(self, tensors, slot: int):
        """The staged batch becomes a device batch: the HOST waits for the copy stream (its copies were queued a step ago: normally
        over) -- no event, the consumer's stream has nothing to wait for -- and the features are cast to the compute dtype ON THE CONSUMER'S
        stream, i.e. at the head of the step that reads them (behind the copy, on the copy stream, the cast measured 0.6 ms per step more)."""
        self.copy_stream.synchronize()
        self._in_flight = False
        dev = self._dev[slot]
        d = tensors["fts"]
        if d is not None and d.dtype != self.feature_dtype:
            c = self._slot(dev, "fts_cast", d, self.feature_dtype, False)
            check(lib.bist_cast(d.data_ptr(), c.data_ptr(), d.numel(), dtype_code(d.dtype), dtype_code(self.feature_dtype),
                                torch.cuda.current_stream(self.device).cuda_stream), "bist_cast")
            tensors["fts"] = c
        for t in tensors.values():
            if t is not None:
                t._bist_generation = getattr(t, "_bist_generation", 0) + 1      # the slot tensors are reused: tell consumers that cache by identity
        return (tensors,)

    def host_buffers_reusable(self, batches_ago: Optional[int] = None) -> None:
        """Block the host until staged copies have read their host buffers.  Producers that hand over their OWN pinned tensors and reuse
        them must call this before overwriting them; pageable producers need nothing (their staging copy is guarded inside the feeder).
        Only the batch staged last can still be crossing PCIe (the feeder waits for every earlier one before it yields it), so whatever
        ``batches_ago`` names, the host waits for the copy stream -- where nothing but copies is ever queued ahead of it."""
        if batches_ago is not None and batches_ago < 1:
            raise ValueError("batches_ago counts back from the latest staged batch: 1, 2, ...")
        if getattr(self, "_in_flight", False):
            self.copy_stream.synchronize()
            self._in_flight = False

    def _batch(self, tensors) -> Batch:
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
            slot = (slot + 1) % self.slots
            try:
                nxt = next(it)
            except StopIteration:
                nxt = None
            fin = self._finish(*cur)                                           # batch i has crossed (host wait), cast, event
            pending = self._stage(nxt, slot) if nxt is not None else None      # batch i+1 starts crossing PCIe now
            b = self._batch(*fin)
            yield b
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))              # consumer of batch i is done with slot cur_slot
            self._free[cur_slot] = done
