"""Beam search over the HIP model (reference: model/decode.py:53-104).

The host-side bookkeeping (hypothesis lists, numpy argsort tie order, length penalty) follows the
reference step for step so that n-best lists are identical; what changes is the device work: the
reference re-runs all visual/caption reasoning layers for every hypothesis of every step although
they do not depend on the partial response -- here they are computed once per turn
(``MultimodalDecoder8.REASONING_CACHE``), and the hypotheses of a step, which all have the same length, go
through the decoder layers and the generator as ONE batch of ``beam`` rows instead of ``beam`` separate
``model.decode`` calls (decode.py:62-66 loops over them; the rows are independent, the results identical).
"""
from __future__ import annotations

import numpy as np
import torch

import types

from ..data.batch import subsequent_mask

BATCH_HYPOTHESES = True      # False: one model.decode call per hypothesis, like the reference


def _rows(t, n):
    """[1, ...] -> contiguous [n, ...] copies (the kernels want dense batches)."""
    return t.expand(n, *t.shape[1:]).contiguous()


def _turn_for_rows(batch, ft, n, cache):
    """The turn's static inputs replicated for n hypotheses: ids / masks of the dialogue and the encoded text plus the
    per-layer reasoning results (never the video tensor).  Built once per row count."""
    hit = cache.get(n)
    if hit is None:
        b = types.SimpleNamespace(**{k: v for k, v in vars(batch).items()})
        for name in ("query", "his", "cap", "query_mask", "his_mask", "cap_mask"):
            v = getattr(batch, name, None)
            setattr(b, name, _rows(v, n) if v is not None else None)
        b.fts = None
        f = {k: _rows(ft[k], n) for k in ("encoded_query", "encoded_his", "encoded_cap") if ft.get(k) is not None}
        f["_bist_reasoning"] = [{k: _rows(v, n) for k, v in layer.items()} for layer in ft["_bist_reasoning"]]
        hit = cache[n] = (b, f)
    return hit


def beam_search_decode(model, batch, max_len, start_symbol, unk_symbol, end_symbol, pad_symbol, beam=5, penalty=1.0,
                       nbest=5, min_len=1, train_args=None, dec_eos=False):
    dev = batch.query.device
    ft = model.encode(batch)
    hyplist = [([], 0.0, torch.full((1, 1), start_symbol, dtype=torch.long, device=dev))]
    best_state, comp_hyplist = None, []
    rows_cache = {}
    for l in range(max_len):
        new_hyplist, argmin = [], 0
        lp_rows = None
        if BATCH_HYPOTHESES and len(hyplist) > 1 and "_bist_reasoning" in ft:
            n = len(hyplist)
            bn, fn = _turn_for_rows(batch, ft, n, rows_cache)
            bn.trg = torch.cat([st for _, _, st in hyplist], dim=0)
            bn.trg_mask = subsequent_mask(bn.trg.size(1), dev)
            fn = model.decode(bn, fn)
            step = dict(fn)
            step["decoded_text"] = fn["decoded_text"][:, -1:].contiguous()
            step["encoded_tgt"] = fn["encoded_tgt"][:, -1:].contiguous()
            lp_rows = model.generator(step, bn, train_args).float().cpu().numpy()        # [n, 1, V]
        for idx, (out, lp, st) in enumerate(hyplist):
            if lp_rows is not None:
                lp_vec = np.squeeze(lp_rows[idx:idx + 1] + lp)
            else:
                batch.trg = st
                batch.trg_mask = subsequent_mask(st.size(1), dev)
                ft = model.decode(batch, ft)
                step = dict(ft)
                step["decoded_text"] = ft["decoded_text"][:, -1:].contiguous()
                step["encoded_tgt"] = ft["encoded_tgt"][:, -1:].contiguous()
                logp = model.generator(step, batch, train_args)
                lp_vec = np.squeeze(logp.float().cpu().numpy() + lp)
            if l >= min_len:
                new_lp = lp_vec[end_symbol] + penalty * (len(out) + 1)
                comp_hyplist.append((out, new_lp))
                if best_state is None or best_state < new_lp:
                    best_state = new_lp
            for o in np.argsort(lp_vec)[::-1]:
                if o == unk_symbol or (not dec_eos and o == end_symbol):
                    continue
                new_lp = lp_vec[o]
                if len(new_hyplist) == beam:
                    if new_hyplist[argmin][1] < new_lp:
                        new_st = torch.cat([st, torch.full((1, 1), int(o), dtype=torch.long, device=dev)], dim=1)
                        new_hyplist[argmin] = (out + [o], new_lp, new_st)
                        argmin = min(enumerate(new_hyplist), key=lambda e: e[1][1])[0]
                    else:
                        break
                else:
                    new_st = torch.cat([st, torch.full((1, 1), int(o), dtype=torch.long, device=dev)], dim=1)
                    new_hyplist.append((out + [o], new_lp, new_st))
                    if len(new_hyplist) == beam:
                        argmin = min(enumerate(new_hyplist), key=lambda e: e[1][1])[0]
        hyplist = new_hyplist
    if comp_hyplist:
        return sorted(comp_hyplist, key=lambda e: -e[1])[:nbest], best_state
    return [([], 0)], None
