"""The ``Batch`` contract the hot path consumes (reference: data/dataset.py:59-131).

Only the fields and masks that ``MTN.forward`` / the generators read are built here; loading
AVSD json/.npy files is out of scope.  Unlike the reference's Batch the
masks can be derived on the device: ``temporal_mask`` comes from a HIP reduction over the
features (``ops.temporal_mask``) instead of a host-side ``fts.sum(2).sum(-1) != 0``.
"""
from __future__ import annotations

from typing import Optional

import torch

PAD = 1  # '<blank>' (data/data_handler.py:23; collate pads with 1, dataset.py:184-193)


def subsequent_mask(size: int, device=None) -> torch.Tensor:
    """[1,size,size] bool, True where a position may be attended (data/data_utils.py:14-18)."""
    m = torch.ones(1, size, size, dtype=torch.bool, device=device).tril_()
    m._bist_causal = size          # content tag: consumers may cache what they derive from a pure causal mask by its size alone
    return m


class Batch:
    """Same attribute names as the reference Batch (dataset.py:59-99).

    query/his/cap/trg/trg_y: int64 [B, L*]; fts: float [B,T,S,C] (f32 or bf16).
    """

    def __init__(self, query, his, fts, cap, trg, trg_y=None, pad: int = PAD, vids=None, qa_ids=None):
        self.vids, self.qa_ids = vids, qa_ids
        self.query, self.his, self.cap, self.trg, self.trg_y = query, his, cap, trg, trg_y
        self.fts = fts
        self.query_mask = (query != pad).unsqueeze(-2)              # dataset.py:66
        self.query_mask2 = torch.cat([self.query_mask, self.query_mask], dim=0)   # the same mask for the stacked (t2s, s2t) batch of a reasoning layer
        self.his_mask = (his != pad).unsqueeze(-2)                  # dataset.py:67
        self.cap_mask = (cap != pad).unsqueeze(-2) if cap is not None else None   # dataset.py:92
        self.temporal_mask: Optional[torch.Tensor] = None           # dataset.py:79 (device-side, see below)
        self.trg_mask = self.make_std_mask(trg, pad)                # dataset.py:96
        self.trg_mean_mask = (trg_y != pad) if trg_y is not None else None
        self.ntokens = (trg_y != pad).sum() if trg_y is not None else None   # dataset.py:98
        self.qntokens = (query != pad).sum()                        # dataset.py:99
        self.audio_fts = None
        self.audio_mask = None
        if fts is not None:
            self._derive_temporal_mask()

    def _derive_temporal_mask(self):
        if self.fts.is_cuda:
            from .. import ops
            self.temporal_mask = ops.temporal_mask(self.fts)
        else:   # host-side batches (before move_to_cuda) follow the reference expression literally
            self.temporal_mask = (self.fts.sum(2).sum(-1) != 0).unsqueeze(-2)

    @staticmethod
    def make_std_mask(tgt, pad):
        """pad mask AND causal mask -> [B,Lt,Lt] (dataset.py:101-105)."""
        return (tgt != pad).unsqueeze(-2) & subsequent_mask(tgt.size(-1), tgt.device)

    def move_to_cuda(self):
        """dataset.py:107-127."""
        for name in ("query", "his", "cap", "trg", "trg_y", "query_mask", "query_mask2", "his_mask", "cap_mask", "trg_mask",
                     "trg_mean_mask", "fts", "temporal_mask"):
            v = getattr(self, name)
            if v is not None:
                setattr(self, name, v.to("cuda", non_blocking=True))
        return self
