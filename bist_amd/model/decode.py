"""Beam search over the HIP model (reference: model/decode.py:53-104).

The host-side bookkeeping (hypothesis lists, numpy argsort tie order, length penalty) follows the
reference step for step so that n-best lists are identical; what changes is the device work: the
reference re-runs all visual/caption reasoning layers for every hypothesis of every step although
they do not depend on the partial response -- here ``DecodeCache`` computes them
once per turn.
"""
from __future__ import annotations

import numpy as np
import torch

from ..data.batch import subsequent_mask


def beam_search_decode(model, batch, max_len, start_symbol, unk_symbol, end_symbol, pad_symbol, beam=5, penalty=1.0,
                       nbest=5, min_len=1, train_args=None, dec_eos=False):
    dev = batch.query.device
    ft = model.encode(batch)
    hyplist = [([], 0.0, torch.full((1, 1), start_symbol, dtype=torch.long, device=dev))]
    best_state, comp_hyplist = None, []
    for l in range(max_len):
        new_hyplist, argmin = [], 0
        for out, lp, st in hyplist:
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
