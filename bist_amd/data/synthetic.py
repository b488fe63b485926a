"""Synthetic batches of the benchmark shape: N(0,1) features with trailing all-zero
temporal rows on some clips (which is what drives ``temporal_mask``), uniform token ids in [4,V) with
trailing pad (=1) on a quarter of the rows.  Generated on the host with a seeded numpy stream, moved
to the device once."""
from __future__ import annotations

import numpy as np
import torch

from .batch import PAD, Batch

SOS, EOS = 2, 3


def synthetic_batch(B: int, T: int = 32, S: int = 49, C: int = 2048, Lq: int = 20, Lh: int = 60, Lc: int = 25,
                    Lt: int = 20, vocab: int = 3000, seed: int = 1234, device="cuda", dtype=torch.bfloat16,
                    ragged: bool = True) -> Batch:
    rs = np.random.RandomState(seed)
    fts = torch.from_numpy(rs.standard_normal((B, T, S, C)).astype(np.float32))
    if ragged:
        for i in range(B):
            fts[i, int(rs.randint(T // 2, T + 1)):] = 0.0

    def ids(L, pad_tail):
        x = rs.randint(4, vocab, size=(B, L)).astype(np.int64)
        if pad_tail:
            for i in range(B):
                if rs.rand() < 0.25 and L > 2:
                    x[i, int(rs.randint(L // 2, L)):] = PAD
        return torch.from_numpy(x)

    query, his, cap = ids(Lq, ragged), ids(Lh, ragged), ids(Lc, ragged)
    full = ids(Lt + 1, False)
    full[:, 0] = SOS
    trg, trg_y = full[:, :-1].clone(), full[:, 1:].clone()
    if ragged:
        for i in range(B):
            if rs.rand() < 0.25 and Lt > 2:
                cut = int(rs.randint(Lt // 2, Lt))
                trg_y[i, cut:] = PAD
                trg[i, cut + 1:] = PAD
                trg_y[i, cut - 1] = EOS
    dev = torch.device(device)
    return Batch(query.to(dev), his.to(dev), fts.to(dev).to(dtype), cap.to(dev), trg.to(dev), trg_y.to(dev))
