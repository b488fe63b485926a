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

import os
import types

from .. import functional as Fn
from ..data.batch import subsequent_mask

BATCH_HYPOTHESES = True      # False: one model.decode call per hypothesis, like the reference


def _rows(t, n, lazy=False):
    """[1, ...] -> contiguous [n, ...] copies (the kernels want dense batches); lazy: the stride-0 view, no launch."""
    v = t.expand(n, *t.shape[1:])
    return v if lazy else v.contiguous()


def _turn_for_rows(batch, ft, n, cache, lazy=False):
    """The turn's static inputs replicated for n hypotheses: ids / masks of the dialogue and the encoded text plus the
    per-layer reasoning results (never the video tensor).  Built once per row count.  lazy: stride-0 views instead of copies (~40
    launches per turn) -- for a caller that only hands them to _TurnBuffers.load, which copies them when a step graph needs rows."""
    hit = cache.get(n)
    if hit is None:
        b = types.SimpleNamespace(**{k: v for k, v in vars(batch).items()})
        for name in ("query", "his", "cap", "query_mask", "his_mask", "cap_mask"):
            v = getattr(batch, name, None)
            setattr(b, name, _rows(v, n, lazy) if v is not None else None)
        b.fts = None
        f = {k: _rows(ft[k], n, lazy) for k in ("encoded_query", "encoded_his", "encoded_cap") if ft.get(k) is not None}
        f["_bist_reasoning"] = [{k: _rows(v, n, lazy) for k, v in layer.items()} for layer in ft["_bist_reasoning"]]
        if "_bist_turn" in ft:
            f["_bist_turn"] = ft["_bist_turn"]
        f["_bist_shared_rows"] = True          # every row is the same dialogue (the decoder's persistent kernel relies on it)
        hit = cache[n] = (b, f)
    return hit


STEP_GRAPHS = True           # replay one hipGraph per (rows, prefix length) for the batched decode step


class _TurnBuffers:
    """Static device copies of a turn's inputs for `n` hypothesis rows (ids, masks, encoded text, per-layer reasoning).
    Every step graph of this geometry reads THESE tensors, so a new turn costs one round of small copies, not a capture."""

    def __init__(self, bn, fn, decoder=None):
        self.decoder = decoder
        dense = lambda v: torch.empty(v.shape, device=v.device, dtype=v.dtype).copy_(v)       # (v may be a stride-0 view)
        self.b = types.SimpleNamespace(**vars(bn))
        for name in ("query", "his", "cap", "query_mask", "his_mask", "cap_mask"):
            v = getattr(bn, name, None)
            setattr(self.b, name, dense(v) if v is not None else None)
        self.f = {k: dense(v) for k, v in fn.items() if torch.is_tensor(v)}
        self.f["_bist_shared_rows"] = True
        self.f["_bist_reasoning"] = [{k: dense(v) for k, v in layer.items()} for layer in fn["_bist_reasoning"]]
        self.loaded = None               # the (bn, fn) pair whose memories the decoder's caches hold
        self.rows_of = None              # the (bn, fn) pair whose rows the static buffers hold

    def load(self, bn, fn, need_rows=True):
        """need_rows=False: the step graph about to be replayed reads nothing of these buffers (persistent decoder kernel on its
        key / value caches + the pointer heads on their per-turn constants: both were projected by the turn's first-step graph), so
        the ~40 row copies are skipped when the decoder's caches already hold this turn."""
        dec = self.decoder
        fresh = dec is not None and hasattr(dec, "decode_cache_is") and dec.decode_cache_is(fn.get("_bist_turn"))
        if (need_rows or not fresh) and self.rows_of is not fn:
            self._copy_rows(bn, fn)
            self.rows_of = fn
        if self.loaded is fn:
            return
        self.loaded = fn
        if dec is not None:          # a new turn: re-project the memories' keys / values (from the static buffers) unless the first step did
            if dec._fused_decode_ok(self.b, self.f, self.f["encoded_query"][:, :1]):
                # (the turn's first step -- the replayed encode + first-step graph -- has usually projected them already: same turn token)
                dec.prepare_decode_cache(self.b, self.f, src=fn["_bist_reasoning"], turn=fn.get("_bist_turn"))

    def _copy_rows(self, bn, fn):
        dsts, srcs = [], []
        for name in ("query", "his", "cap", "query_mask", "his_mask", "cap_mask"):
            v = getattr(bn, name, None)
            if v is not None:
                dsts.append(getattr(self.b, name)); srcs.append(v)
        for k, v in fn.items():
            if torch.is_tensor(v):
                dsts.append(self.f[k]); srcs.append(v)
        for dst, src in zip(self.f["_bist_reasoning"], fn["_bist_reasoning"]):
            for k, v in src.items():
                dsts.append(dst[k]); srcs.append(v)
        # ~40 small tensors per turn: grouped by dtype into multi-tensor copies (one launch per group instead of one per tensor)
        groups = {}
        for d_, s_ in zip(dsts, srcs):
            groups.setdefault((d_.dtype, s_.dtype), ([], []))
            groups[(d_.dtype, s_.dtype)][0].append(d_); groups[(d_.dtype, s_.dtype)][1].append(s_.expand_as(d_) if s_.shape != d_.shape else s_)
        for (dd, sd), (ds, ss) in groups.items():
            if dd == sd and len(ds) > 1:
                torch._foreach_copy_(ds, ss)
            else:
                for d_, s_ in zip(ds, ss):
                    d_.copy_(s_)


def _descending(lp_vec, k):
    """Token ids in the order ``np.argsort(lp_vec)[::-1]`` visits them (decode.py:88), as far as the beam loop can look: it takes
    at most ``beam`` candidates per hypothesis and skips at most two symbols (<unk>, <eos>), so the k = beam + 2 largest are
    enough.  A partial selection orders equal values differently from a full sort, so whenever the k+1 largest values are not
    all distinct the full sort is used -- the visiting order is then the reference's in every case."""
    n = lp_vec.shape[0]
    if n <= 4 * k:
        return np.argsort(lp_vec)[::-1]
    top = np.argpartition(lp_vec, n - (k + 1))[n - (k + 1):]
    vals = lp_vec[top]
    if np.unique(vals).size != vals.size:
        return np.argsort(lp_vec)[::-1]
    return top[np.argsort(vals)[::-1]][:k]


def _graph_store(model):
    """The model's captured step graphs, dropped when the parameter VALUES changed since they were captured: the graphs bake the
    addresses of derived operands (packed / fragment-ordered weights), whose contents are refreshed only by eager code."""
    from .. import ops
    key = (ops.WEIGHTS_EPOCH, sum(p._version for p in ops.module_parameters(model)))
    if model.__dict__.get("_bist_step_graphs_key") != key:
        model.__dict__["_bist_step_graphs"] = {}
        model.__dict__["_bist_step_graphs_key"] = key
    return model.__dict__["_bist_step_graphs"]


# Every dialogue geometry (token lengths of query / history / caption, frame count) holds one first-step graph, up to max_len step graphs,
# their static buffers and the decoder's key / value caches (~13 graphs; first sight costs 45-150 ms of captures against ~8 ms per replayed
# turn: scripts/decode_geometries.py).  When more than this many are held, all of them are dropped.  (Measured over a test-set-like sweep,
# scripts/decode_eval_sweep.py: ~25 MiB of buffers and ~0.2 GiB of reserved graph pools per geometry.)
MAX_GEOMETRIES = int(os.environ.get("BIST_DECODE_MAX_GEOMETRIES", "192"))


def _drop_graphs(model):
    """Forget every captured decode graph of the model together with the buffers whose addresses they hold."""
    model.__dict__["_bist_step_graphs"] = {}
    dec = getattr(model, "mutlimodal_decoder", None)
    if dec is not None:
        dec.__dict__.pop("_bist_dec_state", None)


def _graph_step(model, bn, fn, trg, train_args):
    """decode + generator for the n hypothesis rows of one step, replayed from a hipGraph (captured once per geometry:
    row count, prefix length, dialogue lengths, dtype); returns the log-probs [n, 1, V] as a numpy array."""
    dev = bn.query.device
    n, Lt = trg.shape
    geom = (n, tuple(bn.query.shape), tuple(bn.his.shape), None if bn.cap is None else tuple(bn.cap.shape), fn["encoded_query"].dtype,
            len(fn["_bist_reasoning"]))
    store = model.__dict__.setdefault("_bist_step_graphs", {})          # (checked against the weights once per turn: _graph_first_step)
    tb = store.get(("turn",) + geom)
    if tb is None:
        tb = store[("turn",) + geom] = _TurnBuffers(bn, fn, getattr(model, "mutlimodal_decoder", None))
    tb.load(bn, fn)
    g = store.get((Lt,) + geom)
    if g is None:
        strg = torch.zeros((n, Lt), dtype=torch.long, device=dev)
        tb.b.trg = strg
        tb.b.trg_mask = subsequent_mask(Lt, dev)

        def run():
            f2 = model.decode(tb.b, dict(tb.f))
            step = dict(f2)
            step["decoded_text"] = f2["decoded_text"][:, -1:].contiguous()
            step["encoded_tgt"] = f2["encoded_tgt"][:, -1:].contiguous()
            return model.generator(step, tb.b, train_args).float()
        strg.copy_(trg)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = Fn.Graph()
        with Fn.capture_graph(graph):   # other threads (RCCL watchdog) may touch the runtime during capture
            out = run()
        STATS["captures"] += 1
        g = store[(Lt,) + geom] = (graph, strg, tb.b.trg_mask, out)
    graph, strg, _, out = g
    strg.copy_(trg)
    graph.replay()
    return out.cpu().numpy()


INCREMENTAL = os.environ.get("BIST_INCREMENTAL_DECODE", "1") != "0"      # tuning aid: 0 = every step recomputes all prefix rows, like the reference


def _graph_step_incr(model, bn, fn, new_tokens, pos, slot0, mask_np, train_args, shared=None):
    """One decode step for the NEW position only (``pos``) of n hypotheses: the decoder stack's persistent kernel appends the rows'
    self-attention keys / values to its per-layer pools at slots slot0 .. and attends, per hypothesis, the slots named by mask_np
    [n, LkS] (its ancestors' rows from the earlier steps and its own).  The decoder is causal, so the earlier positions' rows are
    what a recompute of the whole prefix (decode.py:62-66) would produce again.  One hipGraph per (n, pos).
    shared = (tokens [n,1] int64, mask [n,LkS] uint8) device tensors: the graph reads THESE (the device-side beam update of the
    previous step has written them, bist_beam_step) and the call returns the log-probs as a DEVICE tensor -- no host round trip."""
    dev = bn.query.device
    n = shared[0].shape[0] if shared is not None else new_tokens.shape[0]
    LkS = shared[1].shape[1] if shared is not None else mask_np.shape[1]
    geom = (n, tuple(bn.query.shape), tuple(bn.his.shape), None if bn.cap is None else tuple(bn.cap.shape), fn["encoded_query"].dtype,
            len(fn["_bist_reasoning"]))
    store = model.__dict__.setdefault("_bist_step_graphs", {})
    tb = store.get(("turn",) + geom)
    if tb is None:
        tb = store[("turn",) + geom] = _TurnBuffers(bn, fn, getattr(model, "mutlimodal_decoder", None))
    key = ("incr", pos, slot0, LkS, None if shared is None else (shared[0].data_ptr(), shared[1].data_ptr())) + geom
    g = store.get(key)
    tb.load(bn, fn, need_rows=g is None or not g[4])
    if g is None:
        strg = shared[0] if shared is not None else torch.zeros((n, 1), dtype=torch.long, device=dev)
        smask = shared[1] if shared is not None else torch.zeros((n, LkS), dtype=torch.uint8, device=dev)
        sb = types.SimpleNamespace(**vars(tb.b))
        sb.trg, sb.trg_mask = strg, None

        flags = {}

        def run():
            f = dict(tb.f)
            f["_bist_incr"] = (slot0, smask)
            f2 = model.decode(sb, f, pos)
            out = model.generator(f2, sb, train_args).float()
            # neither the decoder layers nor the heads read the rows of tb: the replays of this graph do not need them refreshed
            flags["self_contained"] = "_bist_turn_consts" in f2 and bool(f2.get("_bist_ptr_fast"))
            return out
        if shared is None:
            strg.copy_(new_tokens)
            smask.copy_(torch.from_numpy(mask_np))
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = Fn.Graph()
        with Fn.capture_graph(graph):
            out = run()
        STATS["captures"] += 1
        g = store[key] = (graph, strg, smask, out, bool(flags.get("self_contained")))
    graph, strg, smask, out, _ = g
    if shared is None:
        strg.copy_(new_tokens)
        smask.copy_(torch.from_numpy(mask_np))
    graph.replay()
    return out if shared is not None else out.cpu().numpy()


_TURN_FIELDS = ("query", "his", "cap", "fts", "query_mask", "query_mask2", "his_mask", "cap_mask", "temporal_mask")


_BUCKETED_FIELDS = {"query": "pad", "his": "pad", "cap": "pad", "query_mask": 0, "query_mask2": 0, "his_mask": 0, "cap_mask": 0}


def _staged_shape(f, v, bucket):
    """Shape of field f in the static buffers of a turn's graphs: the token tensors and their masks reach up to the next multiple of the
    length bucket (see BUCKET: padded positions carry the pad id and a False mask, so nothing reads them; dataset.py:66-67, 92 build the masks from the pad id)."""
    shp = tuple(v.shape)
    if f not in _BUCKETED_FIELDS or not bucket:
        return shp
    L = shp[-1]
    if bucket == "class":
        # the lengths the decoder kernel's caches and the pointer heads come in anyway (their padded lengths): a query goes to 32 (64, 96, ..),
        # a history to 32 / 64 / 128 / 256 / 512 (then multiples of 64), a caption -- it only shapes the first-step graph -- to a multiple of 8
        if f.startswith("his"):
            P = 32
            while P < L and P < 512:
                P *= 2
            P = P if L <= P else -(-L // 64) * 64
        elif f.startswith("query"):
            P = -(-L // 32) * 32
        else:
            P = -(-L // 8) * 8
    else:
        P = L if bucket <= 1 else -(-L // bucket) * bucket
    return shp[:-1] + (P,)


def _graph_first_step(model, batch, start_symbol, train_args, host=True, pad_symbol=None):
    """model.encode + the first decode step (prefix = <sos>) of a turn, replayed from one hipGraph per dialogue geometry:
    ~1000 launches of reasoning at B=1 are launch-bound when issued from Python.  Returns (ft, log-probs [1,1,V] numpy, the turn's
    batch as the graphs see it); ft (encoded text, per-layer reasoning) lives in the graph's static outputs until the next turn of
    this geometry.  pad_symbol (not None: length buckets): the dialogue's fields go into the graph's static buffers -- token tensors and
    masks padded to their bucket on the way -- in ONE launch (bist_stage_inputs)."""
    from .. import ops
    dev = batch.query.device
    bucket = BUCKET if pad_symbol is not None else 0
    geom = tuple((f, None if getattr(batch, f, None) is None else (_staged_shape(f, getattr(batch, f), bucket), getattr(batch, f).dtype))
                 for f in _TURN_FIELDS)

    def stage(sb):
        jobs = []
        for f in _TURN_FIELDS:
            v = getattr(batch, f, None)
            if v is not None:
                pad = _BUCKETED_FIELDS.get(f, 0)
                jobs.append((v if v.is_contiguous() else v.contiguous(), getattr(sb, f), pad_symbol if pad == "pad" else pad))
        ops.stage_inputs(jobs)
    # A geometry seen before: stage and launch first, check the graphs against the parameters' values while the device runs (the walk
    # over ~900 parameters costs 0.2 ms of host time with the device idle otherwise).  Stale graphs -- the parameters were written
    # since the capture -- are dropped by _graph_store; what the replay left behind is overwritten by the fresh turn below.
    held = model.__dict__.get("_bist_step_graphs")
    g = held.get(("first",) + geom) if held is not None else None
    launched = False
    if g is not None:
        stage(g[1])
        g[0].replay()
        launched = True
    store = _graph_store(model)
    if store is not held:
        if launched:
            torch.cuda.synchronize(dev)      # (the void replay has finished before its graph objects go)
        g, launched = None, False
    if g is None:
        if sum(1 for k in store if isinstance(k, tuple) and k and k[0] == "first") >= MAX_GEOMETRIES:
            _drop_graphs(model)              # bounded memory over a test set of many dialogue lengths: start over (re-captured on demand)
            store = _graph_store(model)
        sb = types.SimpleNamespace(**vars(batch))
        for f in _TURN_FIELDS:
            v = getattr(batch, f, None)
            setattr(sb, f, torch.empty(_staged_shape(f, v, bucket), dtype=v.dtype, device=dev) if v is not None else None)
        sb.trg = torch.full((1, 1), start_symbol, dtype=torch.long, device=dev)
        sb.trg_mask = subsequent_mask(1, dev)
        stage(sb)

        def run():
            f2 = model.decode(sb, model.encode(sb))
            step = dict(f2)
            step["decoded_text"] = f2["decoded_text"][:, -1:].contiguous()
            step["encoded_tgt"] = f2["encoded_tgt"][:, -1:].contiguous()
            return f2, model.generator(step, sb, train_args).float()
        # At B = 1 every launch of the first step is a few microseconds: several chains buy little and every cross-chain edge costs two
        # sync launches under the split executor (measured 7.85 ms per turn with four chains, 7.15 ms through the runtime's executor --
        # whose replays of this very graph have crashed).  FIRST_STEP_STREAMS = 1 captures it on ONE stream: the
        # runtime's single-queue path, 0.3 us of host time per node.
        conc = Fn.CONCURRENT
        if FIRST_STEP_STREAMS == 1:
            Fn.CONCURRENT = False
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                run()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = Fn.Graph()
            with Fn.capture_graph(graph):   # other threads (RCCL watchdog) may touch the runtime during capture
                f2, out = run()
        finally:
            Fn.CONCURRENT = conc
        STATS["captures"] += 1
        fused = "_bist_fused_first" in f2
        g = store[("first",) + geom] = (graph, sb, f2, out, fused)
    graph, sb, f2, out, _ = g
    if not launched:
        graph.replay()
    seen = types.SimpleNamespace(**vars(batch))          # the turn's batch as its graphs see it (padded fields): what the later steps replicate
    for f in _TURN_FIELDS:
        setattr(seen, f, getattr(sb, f))
    ft = {k: v for k, v in f2.items() if k not in ("_bist_reasoning", "_bist_incr", "_bist_fused_first")}
    ft["_bist_reasoning"] = [dict(layer) for layer in f2["_bist_reasoning"]]
    # the replay projected this turn's memories into the decoder's key / value caches (and left position 0's self-attention
    # keys / values in slot 0 of the pools): the step graphs of this turn need not project them again
    ft["_bist_turn"] = object()
    dec = getattr(model, "mutlimodal_decoder", None)
    if hasattr(dec, "select_decode_cache"):
        # True: the replayed first step ran the decoder layers through the persistent kernel (position 0 sits in slot 0 of its pools)
        ft["_bist_pool_ready"] = bool(g[4]) and dec.select_decode_cache(ft, ft["_bist_turn"])
    return ft, (out.cpu().numpy() if host else out), seen


# What the turns of this process did (bench.py reports it next to the turn times): turns decided on the device, turns the device handed back
# to the host loop (ties / NaN / too few candidates: the turn is then run TWICE), turns on the host loop from the start, hipGraph captures.
STATS = {"turns": 0, "device": 0, "host_redo": 0, "host": 0, "captures": 0}


FIRST_STEP_STREAMS = int(os.environ.get("BIST_DECODE_FIRST_STREAMS", "0"))      # tuning aid: 1 = the turn's first step captured on one stream; 0 = the model's streams


DEVICE_BEAM = os.environ.get("BIST_DEVICE_BEAM", "1") != "0"      # tuning aid: 0 = the beam update of every step on the host (one D2H + sync per step, like the reference)


class _BeamState:
    """Device buffers of the device-side beam update (bist_beam_step), one set per (model, beam, max_len): next-step tokens and masks
    (the static inputs of the incremental step graphs), running scores, per-step records, the sticky fallback flag."""

    def __init__(self, dev, beam, max_len):
        z = lambda *shape, dt: torch.zeros(*shape, device=dev, dtype=dt)
        self.beam, self.max_len = beam, max_len
        self.lp = z(beam, dt=torch.float32)
        self.tok = z(beam, 1, dt=torch.long)
        self.mask64 = z(beam, 64, dt=torch.uint8)
        self.mask = {32: z(beam, 32, dt=torch.uint8), 64: z(beam, 64, dt=torch.uint8)}
        kc = beam + 2
        self.cand_val, self.cand_idx, self.eos_val = z(beam, kc, dt=torch.float32), z(beam, kc, dt=torch.int32), z(beam, dt=torch.float32)
        # records in ONE buffer (one copy to the host per turn): parent | token | rows | flag (int32), then score | completed (f32 viewed)
        n = max_len * beam
        self.rec_i = z(2 * n + max_len + 1, dt=torch.int32)
        self.rec_f = z(2 * n, dt=torch.float32)
        self.parent, self.token = self.rec_i[:n], self.rec_i[n:2 * n]
        self.rows, self.flag = self.rec_i[2 * n:2 * n + max_len], self.rec_i[2 * n + max_len:]
        self.score, self.comp = self.rec_f[:n], self.rec_f[n:]
        self.init = torch.zeros(beam, 64, dtype=torch.uint8)
        self.init[0, 0] = 1                      # the turn's first row attends slot 0 (<sos>)
        self.init = self.init.to(dev)

    def reset(self):
        self.lp.zero_()
        self.flag.zero_()
        self.mask64.copy_(self.init)

    def step(self, logp, n, step, min_len, unk, eos, dec_eos, penalty, slot0_next, LkS_next):
        from .. import _lib
        from ..ops import _stream
        V = logp.shape[-1]
        lg = logp.reshape(-1, V)
        mo = self.mask[LkS_next] if LkS_next else None
        _lib.check(_lib.lib.bist_beam_step(lg.data_ptr(), self.lp.data_ptr(), self.tok.data_ptr(), self.mask64.data_ptr(),
                                           mo.data_ptr() if mo is not None else None, self.cand_val.data_ptr(), self.cand_idx.data_ptr(),
                                           self.eos_val.data_ptr(), self.parent.data_ptr(), self.token.data_ptr(), self.score.data_ptr(),
                                           self.comp.data_ptr(), self.rows.data_ptr(), self.flag.data_ptr(), n, V, self.beam, step, min_len, unk, eos,
                                           1 if dec_eos else 0, slot0_next, LkS_next, float(penalty), _stream()), "bist_beam_step")


def _device_beam_turn(model, batch, ft, out0, max_len, unk_symbol, end_symbol, beam, penalty, nbest, min_len, train_args, dec_eos):
    """The beam loop of decode.py:59-99 with the update of every step on the device: twelve graph replays and twenty-four small
    launches queued back to back, ONE copy of the step records to the host at the end.  Returns None when the device flagged a
    case it does not decide like numpy (ties among a row's largest values, NaN, too few candidates): the caller re-runs the turn
    on the host path."""
    dev = batch.query.device
    key = ("beam_state", beam, max_len, dev)
    store = model.__dict__.setdefault("_bist_step_graphs", {})
    bs = store.get(key)
    if bs is None:
        bs = store[key] = _BeamState(dev, beam, max_len)
    bs.reset()
    lks = lambda slot0: 32 if slot0 + beam <= 32 else 64
    bs.step(out0, 1, 0, min_len, unk_symbol, end_symbol, dec_eos, penalty, beam, lks(beam) if max_len > 1 else 0)
    if max_len > 1:
        bn, fn = _turn_for_rows(batch, ft, beam, {}, lazy=True)
    for l in range(1, max_len):
        slot0 = l * beam
        out = _graph_step_incr(model, bn, fn, None, l, slot0, None, train_args, shared=(bs.tok, bs.mask[lks(slot0)]))
        nxt = (l + 1) * beam
        bs.step(out, beam, l, min_len, unk_symbol, end_symbol, dec_eos, penalty, nxt, lks(nxt) if l + 1 < max_len else 0)
    rec_i, rec_f = bs.rec_i.cpu().numpy(), bs.rec_f.cpu().numpy()          # the turn's only synchronisation
    n = max_len * beam
    if int(rec_i[2 * n + max_len]) != 0:
        return None
    parent, token, rows = rec_i[:n].reshape(max_len, beam), rec_i[n:2 * n].reshape(max_len, beam), rec_i[2 * n:2 * n + max_len]
    comp = rec_f[n:].reshape(max_len, beam)
    comp_hyplist, best_state = [], None
    cur = [[]]
    for l in range(max_len):
        if l >= min_len:
            for idx in range(int(rows[l])):
                new_lp = comp[l, idx]
                comp_hyplist.append((cur[idx], new_lp))
                if best_state is None or best_state < new_lp:
                    best_state = new_lp
        cur = [cur[int(parent[l, j])] + [int(token[l, j])] for j in range(beam) if parent[l, j] >= 0]
    if comp_hyplist:
        return sorted(comp_hyplist, key=lambda e: -e[1])[:nbest], best_state
    return [([], 0)], None


# Token tensors of a dialogue (and their masks) padded to multiples of this many positions on their way into the graphs' static buffers
# (_staged_shape, bist_stage_inputs; 0 / 1 = off).  The graphs and cache buffers are per dialogue GEOMETRY; real dialogues come in every
# length, buckets bound the number of geometries.  Padded positions carry the pad id, so every mask excludes them (dataset.py:66-67, 92) and
# the padded rows' own outputs are never read: the result is that of the unpadded dialogue (tested against the reference's golden n-best).
# A test set has a different (query, history, caption) length triple in nearly every turn and every new triple costs ~40 ms of captures.
# An integer n pads every token tensor to a multiple of n; "class" (the default) pads per field to the lengths the decoder kernel's caches
# and the pointer heads come in anyway (_staged_shape: query 32 / 64, history 32 / 64 / 128 / 256, caption multiples of 8), so that a test
# set meets a few dozen geometries in all.  Measured (scripts/decode_eval_sweep.py, 300 turns of growing histories, every turn a new exact
# triple): 46 / 29 / 18 / 11 ms per turn for 0 / 8 / 16 / "class" (21 of 300 turns capture, 16 first-step graphs held); at the bench
# geometry (20, 60, 25) -> (32, 64, 32) the padding costs nothing measurable (scripts/bench_decode.py: 6.6-6.9 ms against 7.0-7.3 ms with 8).
BUCKET = (lambda v: v if v == "class" else int(v))(os.environ.get("BIST_DECODE_BUCKET", "class"))


def beam_search_decode(model, batch, max_len, start_symbol, unk_symbol, end_symbol, pad_symbol, beam=5, penalty=1.0,
                       nbest=5, min_len=1, train_args=None, dec_eos=False):
    dev = batch.query.device
    use_graphs = (STEP_GRAPHS and BATCH_HYPOTHESES and batch.query.is_cuda and not torch.is_grad_enabled() and not model.training
                  and getattr(type(model.mutlimodal_decoder), "REASONING_CACHE", False))
    lp_first = None
    STATS["turns"] += 1
    if use_graphs:
        dec0 = getattr(model, "mutlimodal_decoder", None)
        want_dev = (DEVICE_BEAM and INCREMENTAL and max_len * beam <= 64 and beam + 2 <= 16 and dec0 is not None
                    and getattr(dec0, "FUSED_DECODE", False) and Fn.FUSED_DECODE and min_len >= 0)
        # (from here on `batch` is the dialogue as the turn's graphs hold it: token tensors and masks padded to their length bucket)
        ft, out0, batch = _graph_first_step(model, batch, start_symbol, train_args, host=not want_dev, pad_symbol=pad_symbol)
        if want_dev and ft.get("_bist_pool_ready", False) and out0.shape[-1] <= 4096 and out0.shape[-1] >= beam + 3:
            res = _device_beam_turn(model, batch, ft, out0, max_len, unk_symbol, end_symbol, beam, penalty, nbest, min_len, train_args, dec_eos)
            if res is not None:
                dec0.check_decode_errors()
                STATS["device"] += 1
                return res
            STATS["host_redo"] += 1
            # (ties / NaN / too few candidates: decided on the host exactly like the reference, from the same first step)
        lp_first = out0.cpu().numpy() if want_dev else out0
        STATS["host"] += 0 if want_dev else 1
    else:
        STATS["host"] += 1
        ft = model.encode(batch)
    # the hypotheses' token prefixes live on the HOST (the reference keeps them as device tensors and pays two tiny device
    # launches per candidate: decode.py:88-99); one [n, Lt] copy per step carries them to the device
    # a hypothesis = (tokens after <sos>, log-prob, pool slots of its prefix rows); its token prefix is [<sos>] + tokens (the reference
    # carries it as a device tensor `st`, decode.py:56,92,98: two tiny launches per candidate)
    hyplist = [([], 0.0, ())]
    best_state, comp_hyplist = None, []
    rows_cache = {}
    dec = getattr(model, "mutlimodal_decoder", None)
    # one decode step at a time: position l of hypothesis j is computed ONCE, at step l, into slot l * beam + j of the decoder
    # kernel's self-attention pools; a hypothesis carries the slots of its prefix (``_bist_slots`` on its token tensor)
    incremental = bool(use_graphs and INCREMENTAL and lp_first is not None and max_len * beam <= 64 and dec is not None
                       and getattr(dec, "FUSED_DECODE", False) and Fn.FUSED_DECODE and ft.get("_bist_pool_ready", False))
    for l in range(max_len):
        new_hyplist, argmin = [], 0
        lp_rows = lp_first if l == 0 else None
        if lp_rows is not None:
            pass
        elif BATCH_HYPOTHESES and len(hyplist) > 1 and "_bist_reasoning" in ft:
            n = len(hyplist)
            bn, fn = _turn_for_rows(batch, ft, n, rows_cache)
            if incremental and all(len(slots) == l for _, _, slots in hyplist):
                slot0 = l * beam
                mask_np = np.zeros((n, 32 if slot0 + n <= 32 else 64), dtype=np.uint8)
                for j, (_, _, slots) in enumerate(hyplist):
                    mask_np[j, list(slots) + [slot0 + j]] = 1
                last = torch.tensor([[out[-1]] for out, _, _ in hyplist], dtype=torch.long)
                lp_rows = _graph_step_incr(model, bn, fn, last, l, slot0, mask_np, train_args)             # [n, 1, V]
            elif use_graphs:
                incremental = False              # (the pools no longer hold every hypothesis's prefix)
                trg = torch.tensor([[start_symbol] + [int(t) for t in out] for out, _, _ in hyplist], dtype=torch.long)
                lp_rows = _graph_step(model, bn, fn, trg, train_args)                        # [n, 1, V]
            else:
                trg = torch.tensor([[start_symbol] + [int(t) for t in out] for out, _, _ in hyplist], dtype=torch.long)
                trg = trg.to(dev)
                bn.trg = trg
                bn.trg_mask = subsequent_mask(trg.size(1), dev)
                f2 = model.decode(bn, dict(fn))
                step = dict(f2)
                step["decoded_text"] = f2["decoded_text"][:, -1:].contiguous()
                step["encoded_tgt"] = f2["encoded_tgt"][:, -1:].contiguous()
                lp_rows = model.generator(step, bn, train_args).float().cpu().numpy()    # [n, 1, V]
        for idx, (out, lp, slots) in enumerate(hyplist):
            own_slots = slots + (l * beam + idx,)       # the rows of this hypothesis's prefix, this step's included
            if lp_rows is not None:
                lp_vec = np.squeeze(lp_rows[idx:idx + 1] + lp)
            else:
                st = torch.tensor([[start_symbol] + [int(t) for t in out]], dtype=torch.long)
                batch.trg = st.to(dev)
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
            for o in _descending(lp_vec, beam + 2):
                if o == unk_symbol or (not dec_eos and o == end_symbol):
                    continue
                new_lp = lp_vec[o]
                if len(new_hyplist) == beam:
                    if new_hyplist[argmin][1] < new_lp:
                        new_hyplist[argmin] = (out + [o], new_lp, own_slots)
                        argmin = min(enumerate(new_hyplist), key=lambda e: e[1][1])[0]
                    else:
                        break
                else:
                    new_hyplist.append((out + [o], new_lp, own_slots))
                    if len(new_hyplist) == beam:
                        argmin = min(enumerate(new_hyplist), key=lambda e: e[1][1])[0]
        hyplist = new_hyplist
    dec = getattr(model, "mutlimodal_decoder", None)
    if hasattr(dec, "check_decode_errors"):
        dec.check_decode_errors()            # a timed-out grid barrier of the persistent decoder kernel voids the turn: raise
    if comp_hyplist:
        return sorted(comp_hyplist, key=lambda e: -e[1])[:nbest], best_state
    return [([], 0)], None
