"""Response decoder layer and the layer loop with modality fusion on the HIP kernels
(reference: model/decoder.py)."""
from __future__ import annotations

from typing import Dict

import os
import torch
import torch.nn as nn

import ctypes as C

from .. import functional as Fn
from .. import ops
from .. import stamps as STM
from .._lib import BistDecLayer, BistKvFill, check, lib
from .encoder import _cross_attention, _feed_forward, _self_attention
from .modules import LayerNorm, SublayerConnection, clones

Tensor = torch.Tensor
def ag_fuse_one() -> bool:
    from .. import autograd as ag
    return ag.FUSE_ONE_LAUNCH


FUSE_ON_DEC = os.environ.get("BIST_FUSE_ON_DEC", "1") != "0"      # training: the modality fusion on the decoder layers' stream (0: on the main stream)
CAP_OWN_CHAIN = os.environ.get("BIST_CAP_CHAIN", "1") != "0"      # tuning aid: 0 = the caption layers of a training step on the decoder layers' stream
FAN_JOIN = os.environ.get("BIST_FAN_JOIN", "1") != "0"      # 0 = the gradient sums of the multi-stream fans rely on the engine's ordering alone


class MultimodalDecoderLayer12(nn.Module):
    """Causal self-attention, attention to history, to the query, to the fused/encoded modalities,
    FFN (reference: decoder.py:11-60)."""

    def __init__(self, size, attn, nb_attn, ff, dropout, args):
        super().__init__()
        self.size = size
        self.attn = clones(attn, nb_attn)
        self.ff = ff
        self.sublayer = clones(SublayerConnection(size, dropout), nb_attn + 1)
        self.args = args

    def forward(self, b, ft, x):
        a, s, args = self.attn, self.sublayer, self.args
        his, qry = Fn.fan_take(ft, "encoded_his"), Fn.fan_take(ft, "encoded_query")     # training: per-consumer aliases of the encoded texts (one-pass gradient sum)
        x = _self_attention(s[0], a[0], x, b.trg_mask)                                   # decoder.py:21
        x = _cross_attention(s[1], a[1], x, his, b.his_mask)                             # :22
        x = _cross_attention(s[2], a[2], x, qry, b.query_mask)                           # :23
        cnt = 3
        if args.nb_venc_blocks > 0 and args.nb_cenc_blocks > 0 and getattr(args, "enc_vc_combine", "none") != "none":
            x = _cross_attention(s[cnt], a[cnt], x, ft["encoded_ft"], b.query_mask); cnt += 1            # :27-29
        else:
            if args.include_caption != "none":                                                           # :31-36
                if args.nb_cenc_blocks > 0:
                    x = _cross_attention(s[cnt], a[cnt], x, ft["cap_ft"], b.query_mask)
                else:
                    x = _cross_attention(s[cnt], a[cnt], x, ft["encoded_cap"], b.cap_mask)
                cnt += 1
            if args.nb_venc_blocks > 0:                                                                  # :37-54
                if args.enc_st_combine != "none":
                    raise NotImplementedError("enc_st_combine != 'none' is outside the hot path (SURVEY.md 8a)")
                if args.dec_st_combine == "seq":
                    if args.s2t:
                        x = _cross_attention(s[cnt], a[cnt], x, ft["temporal_ft"], b.query_mask); cnt += 1
                    if args.t2s:
                        x = _cross_attention(s[cnt], a[cnt], x, ft["spatial_ft"], b.query_mask); cnt += 1
                else:
                    tx = _cross_attention(s[cnt], a[cnt], x, ft["temporal_ft"], b.query_mask); cnt += 1
                    sx = _cross_attention(s[cnt], a[cnt], x, ft["spatial_ft"], b.query_mask); cnt += 1
                    x = Fn.add(tx, sx)
        return _feed_forward(s[cnt], self.ff, x)                                                         # :58


class MultimodalDecoder8(nn.Module):
    """Layer loop: visual reasoning layer, caption layer, modality fusion, decoder layer
    (reference: decoder.py:62-186).  Only ``enc_st_combine == 'none'`` is in scope."""

    def __getstate__(self):
        """torch.save / copy.deepcopy: the persistent kernel's scratch, per-turn caches, descriptors and hooks (instance attributes named
        _bist_*: device buffers whose addresses captured graphs hold, ctypes arrays, closures) stay behind; they are rebuilt on demand."""
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_bist_")}

    def __init__(self, v_layer, c_layer, a_layer, layer, venc_N, cenc_N, aenc_N, N, args):
        super().__init__()
        self.layers = clones(layer, N)
        self.N, self.v_N, self.c_N, self.a_N = N, venc_N, cenc_N, aenc_N
        self.norm = LayerNorm(layer.size)
        self.args = args
        if aenc_N > 0:
            raise NotImplementedError("audio reasoning is outside the hot path (SURVEY.md 8a)")
        if self.v_N > 0:
            if args.enc_st_combine != "none":
                raise NotImplementedError("enc_st_combine=%s is outside the hot path: multi-layer models only run "
                                          "with 'none' in the reference (SURVEY.md appendix)" % args.enc_st_combine)
            self.v_layers = clones(v_layer, self.v_N)
            self.spatial_out_norm = LayerNorm(v_layer.size)
            self.temporal_out_norm = LayerNorm(v_layer.size)
        if self.c_N > 0:
            self.c_layers = clones(c_layer, self.c_N)
            self.cap_out_norm = LayerNorm(c_layer.size)
        if self.v_N > 0 and args.enc_vc_combine == "dyn":
            factor = 1 + (args.include_caption != "none") + bool(args.t2s) + bool(args.s2t)    # decoder.py:95-103
            self.vc_combine_W = nn.Linear(v_layer.size * factor, factor - 1)

    def _fuse(self, ft: Dict[str, Tensor]) -> None:
        """ft['encoded_ft'] (decoder.py:137-181).  The concat feeding vc_combine_W is never built:
        each modality multiplies its own column block of the weight and accumulates in place."""
        args = self.args
        mode = getattr(args, "enc_vc_combine", "none")
        if self.v_N == 0:
            return
        if self.c_N > 0 and mode == "sum":
            ft["encoded_ft"] = Fn.add(Fn.add(ft["temporal_ft"], ft["spatial_ft"]), ft["cap_ft"])
            return
        if mode != "dyn":
            return
        parts = [Fn.fan_take(ft, "encoded_query")]
        if self.c_N > 0:
            parts.append(Fn.fan_take(ft, "cap_ft"))
        if args.t2s:
            parts.append(Fn.fan_take(ft, "spatial_ft"))
        if args.s2t:
            parts.append(Fn.fan_take(ft, "temporal_ft"))
        W, bias = self.vc_combine_W.weight, self.vc_combine_W.bias
        d = parts[0].shape[-1]
        if W.shape[1] != d * len(parts):
            raise ValueError("vc_combine_W does not match the enabled modalities")
        if (torch.is_grad_enabled() and self.c_N > 0 and args.t2s and args.s2t and parts[0].is_cuda
                and all(p.is_contiguous() for p in parts)):
            # training: the fusion logits and the weighted sum as ONE autograd node -- every modality tensor has one consumer here instead of
            # two (autograd.FuseDynFn); parts = [query, cap, spatial, temporal], score column j -> temporal, spatial, cap (decoder.py:156-159)
            ft["encoded_ft"] = Fn.fuse_dyn(W, bias, parts, (3, 2, 1))
            return
        if torch.is_grad_enabled() and ag_fuse_one() and Fn.switch_logits_ok(W, bias, parts):
            # training without the one-node form (a direction or the caption stack switched off): the same one-launch logits as autograd.FuseDynFn
            score = Fn.switch_logits(W, bias, parts, out_dtype=parts[0].dtype)
        else:
            score = None
            for j, p in enumerate(parts):                  # concat order: query, cap, spatial, temporal
                score = Fn.linear(p, Fn.column_block(W, j, d), bias if j == 0 else None, out=score, accumulate=j > 0)
        # score column -> modality (decoder.py:156-165): 0 temporal, 1 spatial, 2 cap (both directions);
        # one direction: 0 that direction, 1 cap.  Without a caption layer: 0 temporal, 1 spatial.
        if args.t2s and args.s2t:
            xs = [ft["temporal_ft"], ft["spatial_ft"]] + ([ft["cap_ft"]] if self.c_N > 0 else [])
        elif args.s2t:
            xs = [ft["temporal_ft"], ft["cap_ft"]]
        else:
            xs = [ft["spatial_ft"], ft["cap_ft"]]
        if self.c_N == 0 and not (args.t2s and args.s2t):
            return                                          # the reference defines no encoded_ft here (decoder.py:168-181)
        ft["encoded_ft"] = Fn.fuse_modalities(score.view(*parts[0].shape[:-1], -1), xs)

    # ---- one persistent launch for all decoder layers of a decode step (bist_decoder_stack_fwd) ------------------------------
    FUSED_DECODE = True
    HEAD_LOCAL = os.environ.get("BIST_DECSTACK_HEADLOCAL", "0") == "1"      # the kernel's head-local form for <= 16 rows: measured slower, opt-in

    def _fused_decode_ok(self, b, ft, x) -> bool:
        a = self.args
        if not (self.FUSED_DECODE and Fn.FUSED_DECODE and x.is_cuda and x.dim() == 3):
            return False
        if not (self.v_N > 0 and self.c_N > 0 and getattr(a, "enc_vc_combine", "none") != "none"):
            return False                     # the layer must be the four-attention form of decoder.py:27-29
        n, Lt, d = x.shape
        # the kernel reads ONE dialogue's memories for all rows: a single dialogue, or rows that decode.py replicated from one
        if not (ft["encoded_his"].shape[0] == 1 or ft.get("_bist_shared_rows", False)):
            return False
        Lk = max(ft["encoded_his"].shape[1], ft["encoded_query"].shape[1])
        return ops.decoder_stack_ok(n * Lt, d, self.layers[0].attn[0].h, Lk, x.dtype) and len(self.layers[0].attn) == 4

    def _decode_state(self, dev, dtype):
        """Scratch of the persistent kernel and the per-turn key / value caches: allocated once per decoder (fixed addresses, so
        captured step graphs stay valid), zero-initialised (padding rows must be finite)."""
        st = self.__dict__.get("_bist_dec_state")
        if st is None or st["x0"].device != dev or st["x0"].dtype != dtype:
            z = lambda *shape, dt=dtype: torch.zeros(*shape, device=dev, dtype=dt)
            nl = len(self.layers)
            st = {"x0": z(64, 512), "x1": z(64, 512), "q": z(64, 512), "kc": z(nl, 64, 512), "vc": z(nl, 64, 512), "h": z(64, 2048),
                  "sync": z(8, dt=torch.int32), "masks": {}, "kv": None,
                  "p": z(2, 8, 16, 512, dt=torch.float32)}       # per-head partial output projections of the head-local form (R <= 16 rows)
            self.__dict__["_bist_dec_state"] = st
        return st

    def prepare_decode_cache(self, b, ft, src=None, turn=None) -> None:
        """Keys and (transposed) values of the three memories of every decoder layer for this turn: they depend on the encoded
        history / query and on the layer's fused modalities only, not on the prefix (the unfused path recomputes them in every
        decode step).  All hypothesis rows of a turn hold the same memories: row 0 is used."""
        dev, dtype = ft["encoded_his"].device, ft["encoded_his"].dtype
        st = self._decode_state(dev, dtype)
        cache = ft["_bist_reasoning"]
        if st["kv"] is not None and ((src is not None and st["kv"].get("src") is src) or (turn is not None and st["kv"].get("turn") is turn)):
            st["kv"]["owner"] = cache        # another view of a turn whose memories are already projected: same values
            return
        nl = len(self.layers)
        mems = lambda l: (ft["encoded_his"][0], ft["encoded_query"][0], cache[l]["encoded_ft"][0])
        masks = (b.his_mask[0].reshape(-1), b.query_mask[0].reshape(-1), b.query_mask[0].reshape(-1))
        Lks = [m.shape[0] for m in mems(0)]
        LkPs = [32 if k <= 32 else 64 if k <= 64 else 128 if k <= 128 else 256 if k <= 256 else 512 for k in Lks]     # (histories of 65 .. 512 tokens: the kernel's long core)
        # one set of cache buffers per dialogue geometry, kept for the decoder's lifetime: captured step graphs of that geometry hold
        # their addresses (and the descriptor's), so they are never freed or re-used for another geometry
        kvs = st.setdefault("kv_by_len", {})
        kv = kvs.get((tuple(Lks), nl))
        if kv is None:
            kv = kvs[(tuple(Lks), nl)] = {
                "Lks": Lks, "nl": nl,
                "K": [[torch.zeros(LkPs[c], 512, device=dev, dtype=dtype) for c in range(3)] for _ in range(nl)],
                "VT": [[torch.zeros(512, LkPs[c], device=dev, dtype=dtype) for c in range(3)] for _ in range(nl)],
                "mask": [torch.zeros(LkPs[c], device=dev, dtype=torch.uint8) for c in range(3)]}
        st["kv"] = kv
        for c in range(3):
            kv["mask"][c][:Lks[c]].copy_(masks[c].to(torch.uint8))
        # projections: the history's and the query's keys / values of ALL layers are one product each (they read the encoded text only:
        # on a side stream, beside the per-layer products of the fused modalities, together with the other modules' per-turn constants);
        # then ONE launch scatters the 3 nl packed results into the caches (keys copied, values transposed: bist_decoder_cache_fill)
        wb = self._packed_memory_weights(nl)
        main = torch.cuda.current_stream()
        side = Fn.side_stream(0) if (Fn.CONCURRENT and mems(0)[0].is_cuda) else None
        jobs = (BistKvFill * (3 * nl))()
        keep = []
        if side is not None:
            side.wait_stream(main)
        with torch.cuda.stream(side if side is not None else main):
            for c in (0, 1):
                kvp = Fn.linear(mems(0)[c], wb[c][0], wb[c][1])                               # [Lk, nl * 1024]
                if side is not None:
                    kvp.record_stream(main)                                                   # (read by the fill launch on the main stream)
                keep.append(kvp)
                for l in range(nl):
                    jb = jobs[l * 3 + c]
                    jb.src, jb.K, jb.VT = kvp.data_ptr() + l * 1024 * kvp.element_size(), kv["K"][l][c].data_ptr(), kv["VT"][l][c].data_ptr()
                    jb.Lk, jb.LkP, jb.ld = Lks[c], LkPs[c], kvp.stride(0)
            kv["stamp"] = object()                               # this projection's identity: per-turn constants of other modules hang on it
            for hook in self.__dict__.get("_bist_turn_hooks", {}).values():       # (the pointer generator's folded keys: generator._turn_consts)
                hook(b, ft, kv)
        for l, layer in enumerate(self.layers):
            w, bias = layer.attn[3]._packed((1, 2))                                           # [W_k; W_v] of the fused-modalities memory
            kvp = Fn.linear(mems(l)[2], w, bias)                                              # [Lk, 1024]
            keep.append(kvp)
            jb = jobs[l * 3 + 2]
            jb.src, jb.K, jb.VT, jb.Lk, jb.LkP, jb.ld = kvp.data_ptr(), kv["K"][l][2].data_ptr(), kv["VT"][l][2].data_ptr(), Lks[2], LkPs[2], kvp.stride(0)
        if side is not None:
            main.wait_stream(side)
        check(lib.bist_decoder_cache_fill(jobs, 3 * nl, ops.dtype_code(dtype), ops._stream()), "bist_decoder_cache_fill")
        kv["owner"], kv["src"], kv["turn"] = cache, src, turn  # the reasoning results these keys / values were projected from (one list per turn)

    def _packed_memory_weights(self, nl):
        """[W_k; W_v] of every layer's history (c = 0) and query (c = 1) attention stacked row-wise, [nl * 1024, 512] + bias: one product per
        memory and turn.  Cached by the parameters' values; rewritten in place (captured graphs hold the addresses)."""
        out = []
        cache = self.__dict__.setdefault("_bist_mem_pack", {})
        for c in (0, 1):
            parts = [self.layers[l].attn[1 + c]._packed((1, 2)) for l in range(nl)]
            key = ops.weights_key(*[p for l in range(nl) for p in (self.layers[l].attn[1 + c].linears[1].weight, self.layers[l].attn[1 + c].linears[1].bias,
                                                                   self.layers[l].attn[1 + c].linears[2].weight, self.layers[l].attn[1 + c].linears[2].bias)])
            hit = cache.get(c)
            if hit is None or hit[0] != key:
                w, bias = torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0)
                if hit is not None and hit[1].shape == w.shape and hit[1].dtype == w.dtype and hit[1].device == w.device:
                    hit[1].copy_(w); hit[2].copy_(bias)
                    w, bias = hit[1], hit[2]
                hit = cache[c] = (key, w, bias)
            out.append((hit[1], hit[2]))
        return out

    def decode_cache_is(self, turn) -> bool:
        """The current key / value caches (and the constants that hang on them) hold the memories of turn ``turn``."""
        st = self.__dict__.get("_bist_dec_state")
        return turn is not None and st is not None and st.get("kv") is not None and st["kv"].get("turn") is turn

    def select_decode_cache(self, ft, turn) -> bool:
        """A captured graph has just projected this turn's memories (it wrote the cache buffers of ``ft``'s dialogue geometry): make
        those buffers the current ones and mark them as holding turn ``turn``.  False if that geometry has no buffers yet."""
        st = self.__dict__.get("_bist_dec_state")
        if st is None or not st.get("used"):
            return False
        Lks = (ft["encoded_his"].shape[1], ft["encoded_query"].shape[1], ft["_bist_reasoning"][0]["encoded_ft"].shape[1])
        kv = st.get("kv_by_len", {}).get((Lks, len(self.layers)))
        if kv is None:
            return False
        st["kv"] = kv
        kv["turn"], kv["src"], kv["owner"] = turn, None, None
        return True

    def _decode_desc(self, st):
        """Device array of BistDecLayer descriptors for the current cache buffers (one per dialogue geometry; its CONTENT is rebuilt in
        place when the parameters change, so its address stays valid for captured graphs)."""
        kv = st["kv"]
        ps = ops.module_parameters(self.layers)
        key = ops.weights_key(*ps)
        hit = kv.get("desc")
        if hit is not None and hit[0] == key:
            return hit[1]
        descs = (BistDecLayer * len(self.layers))()
        keep = []
        for l, layer in enumerate(self.layers):
            dsc = descs[l]
            for s_ in range(5):
                dsc.ln_a[s_], dsc.ln_b[s_] = layer.sublayer[s_].norm.a_2.data_ptr(), layer.sublayer[s_].norm.b_2.data_ptr()
            w, bias = layer.attn[0]._packed((0, 1, 2))
            keep += [w, bias]
            dsc.Wqkv, dsc.bqkv = w.data_ptr(), bias.data_ptr()
            for c in range(3):
                at = layer.attn[1 + c]
                dsc.Wq[c], dsc.bq[c] = at.linears[0].weight.data_ptr(), at.linears[0].bias.data_ptr()
                dsc.Kc[c], dsc.VTc[c], dsc.cmask[c] = kv["K"][l][c].data_ptr(), kv["VT"][l][c].data_ptr(), kv["mask"][c].data_ptr()
                dsc.Lk[c], dsc.LkP[c] = kv["Lks"][c], kv["K"][l][c].shape[0]
            for j in range(4):
                dsc.Wo[j], dsc.bo[j] = layer.attn[j].linears[3].weight.data_ptr(), layer.attn[j].linears[3].bias.data_ptr()
            dsc.W1, dsc.b1 = layer.ff.w_1.weight.data_ptr(), layer.ff.w_1.bias.data_ptr()
            dsc.W2, dsc.b2 = layer.ff.w_2.weight.data_ptr(), layer.ff.w_2.bias.data_ptr()
        assert C.sizeof(BistDecLayer) == lib.bist_decoder_layer_desc_bytes()
        host = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8)
        raw = hit[1] if hit is not None else torch.empty(host.numel(), dtype=torch.uint8, device=st["x0"].device)
        raw.copy_(host)
        kv["desc"], kv["desc_keep"] = (key, raw), keep
        return raw

    def _self_mask(self, st, b, n: int, Lt: int, LkS: int):
        """[n*Lt, LkS] uint8: row (j,t) attends key (j',t') iff j' == j and trg_mask[j or 0][t][t'] (dataset.py:101-105).
        Pure causal target masks (data.batch.subsequent_mask tags them; every beam-search step has one) are cached by CONTENT --
        (rows, prefix length, slots) -- and never evicted or replaced: captured step graphs of several dialogue geometries share
        (n, Lt) and bake the buffer's address, so one buffer per key must live as long as the decoder.  Any other target mask is
        built per call (during a capture it then comes from, and lives with, that graph's own memory pool)."""
        tm = b.trg_mask
        causal = getattr(tm, "_bist_causal", None) == Lt and tuple(tm.shape[-2:]) == (Lt, Lt)
        key = (n, Lt, LkS, tm.device)
        if causal:
            hit = st["masks"].get(key)
            if hit is not None:
                return hit
        m = torch.zeros(n * Lt, LkS, device=tm.device, dtype=torch.uint8)
        blk = tm.to(torch.uint8).expand(n, Lt, Lt)
        for j in range(n):
            m[j * Lt:(j + 1) * Lt, j * Lt:(j + 1) * Lt] = blk[j]
        if causal:
            st["masks"][key] = m
        return m

    def check_decode_errors(self) -> None:
        """After a turn: the persistent kernel's sticky error word (one 4-byte read).  Non-zero = one of its grid barriers timed
        out (its 32 workgroups were not resident together), so the rows it returned are not the decoder's output."""
        st = self.__dict__.get("_bist_dec_state")
        if st is not None and st.get("used") and int(st["sync"][4].item()) != 0:
            st["sync"].zero_()
            raise RuntimeError("bist_amd: bist_decoder_stack_fwd timed out at a grid barrier (workgroups not co-resident); "
                               "set BIST_FUSED_DECODE=0 to run the decoder layers as separate launches")

    def _decode_fused(self, b, ft, x):
        n, Lt, d = x.shape
        st = self._decode_state(x.device, x.dtype)
        if st["kv"] is None or st["kv"].get("owner") is not ft["_bist_reasoning"]:
            self.prepare_decode_cache(b, ft)
        st["used"] = True
        incr = ft.get("_bist_incr")
        lkp = max(k.shape[0] for k in st["kv"]["K"][0])       # the padded memory lengths of this turn's caches (the long ones pick the kernel's chunked core)
        if incr is not None:
            # one decode step at a time (decode.py's step graphs): x holds only the NEW position's rows [n, 1, d]; the keys / values of
            # the earlier positions are in the per-layer pools from the earlier steps of this turn; incr = (first free slot, mask
            # [n, LkS] over the pool slots: a hypothesis attends its ancestors' slots and its own)
            slot0, mask = incr
            assert Lt == 1 and mask.shape[0] == n
            out = ops.decoder_stack(self._decode_desc(st), len(self.layers), x.reshape(n, d).contiguous(), st, mask, n, mask.shape[1], slot0,
                                    head_local=self.HEAD_LOCAL, lk_pad_max=lkp)
        else:
            R = n * Lt
            LkS = 32 if R <= 32 else 64
            out = ops.decoder_stack(self._decode_desc(st), len(self.layers), x.reshape(R, d).contiguous(), st, self._self_mask(st, b, n, Lt, LkS), R, LkS,
                                    head_local=self.HEAD_LOCAL, lk_pad_max=lkp)
        ft.update(ft["_bist_reasoning"][-1])
        ft["_bist_turn_consts"] = (self, st["kv"])       # the rows share one dialogue whose per-turn constants live with these caches
        return out.view(n, Lt, d)

    # Keys the reasoning layers write per decoder layer (decoder.py:126-181); everything else in ``ft`` is static.
    _REASONING_KEYS = ("temporal_ft", "spatial_ft", "cap_ft", "encoded_ft")
    REASONING_CACHE = True      # inference only; set False to recompute the reasoning on every decode() like the reference

    def forward(self, b, ft: Dict[str, Tensor], x: Tensor) -> Dict[str, Tensor]:
        """decoder.py:107-186.  The visual / caption reasoning of every layer depends on the encoded inputs only,
        not on the target prefix ``x`` -- yet the reference recomputes it in each of the ~60 ``model.decode`` calls
        of a beam-search turn (decode.py:66; 88 % of the forward).  In inference (eval, no autograd) the
        per-layer results are kept in ``ft`` (the dict ``model.encode`` returned for this turn, so the cache dies
        with the turn) and later calls run only the decoder layers.  Same arithmetic, same results."""
        use_cache = self.REASONING_CACHE and not self.training and not torch.is_grad_enabled()
        cache = ft.get("_bist_reasoning") if use_cache else None
        if cache is not None and len(cache) == len(self.layers):
            if self._fused_decode_ok(b, ft, x):
                x = self._decode_fused(b, ft, x)
            else:
                for l, layer in enumerate(self.layers):
                    ft.update(cache[l])
                    x = layer(b, ft, x)
            ft["decoded_text"] = self.norm(x)
            return ft
        cache = [] if use_cache else None
        if torch.is_grad_enabled():
            # every reader of an encoded text gets an alias of its own (FanOutFn: ONE bist_add_n launch sums their gradients; autograd's own
            # accumulation is a pairwise elementwise launch per extra reader): the query feeds the three reasoning chains, every layer's
            # fusion logits and decoder layer, and two readers per pointer attention; the caption every caption layer and its pointer
            L_ = len(self.layers)
            Fn.fan_set(ft, "encoded_query", 3 + 2 * L_ + 2, join=FAN_JOIN)
            Fn.fan_set(ft, "encoded_his", L_ + 2, join=FAN_JOIN)
            Fn.fan_set(ft, "encoded_cap", L_ + 2, join=FAN_JOIN)
        q = ft["encoded_query"]
        in_ft = {"t2s": Fn.fan_take(ft, "encoded_query"), "s2t": Fn.fan_take(ft, "encoded_query"), "audio": q, "cap": Fn.fan_take(ft, "encoded_query")}
        fused_train = False
        if self.v_N > 0 and "spatiotemporal_ft" in ft:
            # training: the video tensor feeds 4 products per reasoning layer (2 score products, 2 value projections); their [B*T*S, d] gradients are summed in one pass
            L = len(self.layers)
            both = getattr(self.args, "t2s", 1) and getattr(self.args, "s2t", 1)
            vft_ = ft["spatiotemporal_ft"]
            a0 = self.v_layers[0].attn[0]
            fused_train = bool(torch.is_grad_enabled() and both and Fn.FUSED_TRAIN and vft_.is_cuda and vft_.dim() == 4 and vft_.is_contiguous()
                               and q.dtype == vft_.dtype and len(self.v_layers[0].attn) == 6
                               and all(ops.st_stage1_fused_train_ok(vft_.shape[1], vft_.shape[2], q.shape[1], vft_.shape[3], a0.h, dr, vft_.dtype) for dr in (0, 1)))
            if fused_train:
                # stage 1 of both directions as ONE launch each (csrc/st1_fused.hip, training form): no value projections of their own, no
                # score tensors, no region-major copy.  Per layer the video tensor has four readers (per direction: the fused launch's
                # rows and the value projection's weight-gradient / dX products of its backward).  The value / output projection weights
                # of all layers go to fragment order in ONE launch, into buffers that keep their addresses (hipGraph replays re-run it).
                ft["_bist_fused_train"] = True
                ft["_bist_vft_fan"] = Fn.Fan(vft_, 4 * L, FAN_JOIN)
                # deferred optimiser: the pack reads the value / output weights of EVERY reasoning layer, so the pending update of all
                # layer pieces must have landed (the per-layer gates below come too late for this launch)
                Fn.param_gate(len(self.layers))
                ws, outs = [], []
                for vl in self.v_layers[:L]:
                    for ai in (1, 4):
                        ws += [vl.attn[ai].linears[2].weight.detach(), vl.attn[ai].linears[3].weight.detach()]
                        outs += vl.frag_train(ai)
                ops.pack_frag_rows_multi(ws, outs)
            elif (torch.is_grad_enabled() and both and Fn.PERMUTED_T2S and ft["spatiotemporal_ft"].requires_grad
                    and ft["spatiotemporal_ft"].shape[1] >= 64):      # from 64 frames: 21.9 vs 22.3 ms at T = 128; 11.8 vs 11.7 ms at T = 32
                # t2s works on a region-major copy of the video tensor (made once, shared by all layers): contiguous score
                # runs and value tiles in its stage-1 core instead of 16-byte pieces (Fn.permute_ts)
                ft["_bist_vft_fan"] = Fn.Fan(ft["spatiotemporal_ft"], 2 * L + 1, FAN_JOIN)
                ft["_bist_vftp_fan"] = Fn.Fan(Fn.permute_ts(ft["_bist_vft_fan"].take()), 2 * L, FAN_JOIN)
            else:
                ft["_bist_vft_fan"] = Fn.Fan(ft["spatiotemporal_ft"], 4 * L, FAN_JOIN)
        dec_pending = None
        # inference with the reasoning cache: the decoder layers run AFTER the reasoning layers, as the persistent launch of later decode
        # steps (bist_decoder_stack_fwd) -- which also leaves this call's self-attention keys / values in the per-layer pools
        fused_after = cache is not None and self._fused_decode_ok(b, ft, x)
        # Training: the value projections of layer l+1 (two big GEMMs that depend on the video tensor only) are issued on the
        # caption stream ahead of decoder layer l, and awaited through an event just before the stage-1 cores; their
        # backward products then run on that stream under the small-kernel chains of the two directions.
        values_ahead = (torch.is_grad_enabled() and not (fused_train and Fn.fused_train_own_v(ft["spatiotemporal_ft"].shape[1]) == 1) and self.c_N > 0 and self.v_N > 0 and Fn.CONCURRENT and x.is_cuda and Fn.VALUES_AHEAD
                        and getattr(self.args, "t2s", 1) and getattr(self.args, "s2t", 1) and "_bist_vft_fan" in ft)

        def issue_values(l):
            main_, side_ = torch.cuda.current_stream(), Fn.side_stream(1)
            fan = ft.get("_bist_vftp_fan") or ft["_bist_vft_fan"]
            va = fan.take()
            side_.wait_stream(main_)
            with torch.cuda.stream(side_):
                STM.mark("L%d values ahead" % l) if STM.ENABLED else None
                v1 = self.v_layers[l].train_value(va, 1)          # t2s only -- its consumer (and its gradient) live on the main stream; an
                v4 = None                                         # edge between two side streams crashes hipGraph capture
                ev = torch.cuda.Event()
                ev.record(side_)
            Fn._keep_taken(v1); Fn._keep_taken(v4)               # allocated on this stream, consumed (and saved) on the main one
            ft["_bist_v_pre"] = (v1, v4, ev)
        Fn.param_gate(1)                     # deferred optimiser: layer 0's parameters (and everything outside the layer stacks) are final
        x = Fn.bucket_mark(x, 0)             # everything recorded from here on belongs to the layer stacks (see the mark inside the loop)
        if values_ahead:
            issue_values(0)
        conc_keep = Fn.CONCURRENT and x.is_cuda
        for l, layer in enumerate(self.layers):
            STM.LAYER = l
            Fn.param_gate(1 + min(l + 1, len(self.layers) - 1))      # layer l + 1: its value projection is issued during layer l
            fork_cap = self.c_N > 0 and self.v_N > 0 and Fn.CONCURRENT and x.is_cuda
            if fork_cap:                     # the caption reasoning layer is independent of the visual one
                main, side = torch.cuda.current_stream(), Fn.side_stream(1)
                # Training under the split-graph executor: the caption layers on the FOURTH chain (the inference / leaf stream, idle in a
                # training step) instead of ahead of the decoder layer on its stream -- that stream carried 455 us per layer forward (value
                # projection 30, decoder layer 285, caption layer 100) against 270 on each direction's chain and bounded the forward pass.
                cside = Fn.fourth_stream() if (CAP_OWN_CHAIN and torch.is_grad_enabled() and Fn.fourth_stream() is not None) else side
                cside.wait_stream(main)
                with torch.cuda.stream(cside):
                    in_ft["cap"] = STM.through(in_ft["cap"], "cap in", True)
                    in_ft = self.c_layers[l](in_ft, ft, b)
                    in_ft["cap"] = STM.through(in_ft["cap"], "cap out", True)
                    ft["cap_ft"], in_ft["cap"] = Fn.layernorm_res(in_ft["cap"], self.cap_out_norm.a_2, self.cap_out_norm.b_2, self.cap_out_norm.eps)                           # decoder.py:132
            if self.v_N > 0:
                if torch.is_grad_enabled():
                    ft["_bist_out_norms"] = (self.spatial_out_norm, self.temporal_out_norm)
                if fork_cap:
                    ft["_bist_cap_fork"] = True
                in_ft = self.v_layers[l](in_ft, ft, b)
                ft.pop("_bist_out_norms", None)
                ft.pop("_bist_cap_fork", None)
                if in_ft.pop("_norms_done", False):
                    pass                              # the layer applied the two output norms at the end of its direction chains (encoder.py)
                else:
                    if self.args.s2t:
                        ft["temporal_ft"] = self.temporal_out_norm(in_ft["s2t"])             # decoder.py:127
                    if self.args.t2s:
                        # the t2s stream goes on to the next layer THROUGH the norm's node (its gradient is added inside the norm's backward
                        # kernel instead of by an accumulation launch); the s2t stream is not chained: its norm runs on the main stream, its
                        # next layer on a side stream, and the chain would put two cross-stream hops per layer on that direction's backward
                        n_ = self.spatial_out_norm
                        ft["spatial_ft"], in_ft["t2s"] = Fn.layernorm_res(in_ft["t2s"], n_.a_2, n_.b_2, n_.eps)      # :129
            if fork_cap:
                main.wait_stream(cside)
            elif self.c_N > 0:
                in_ft = self.c_layers[l](in_ft, ft, b)
                ft["cap_ft"], in_ft["cap"] = Fn.layernorm_res(in_ft["cap"], self.cap_out_norm.a_2, self.cap_out_norm.b_2, self.cap_out_norm.eps)                               # :132
            if torch.is_grad_enabled() and conc_keep:
                # every tensor handed from one stream's chain to another's at this layer boundary (see Fn._keep_taken)
                for t_ in (*[v for v in in_ft.values() if torch.is_tensor(v)], ft.get("cap_ft"), ft.get("spatial_ft"), ft.get("temporal_ft")):
                    Fn._keep_taken(t_)
            if torch.is_grad_enabled() and l + 1 == len(self.layers) and getattr(self.args, "auto_encoder", 0):
                # the last layer's outputs also feed the auto-encoder heads (optimize.py:66-82): an alias per consumer, one-pass gradient sum
                for k_ in ("cap_ft", "spatial_ft", "temporal_ft"):
                    Fn.fan_set(ft, k_, 2)
            STM.lmark("main joined")
            # Training with the decoder layer pipelined onto the caption / decoder stream: the fusion of the modalities (decoder.py:140-165) goes
            # there too, ahead of its only consumer -- its forward launches (3) and, above all, its backward ones (~10 per layer, 80-150 us:
            # the fusion logits' weight gradient) then leave the main stream's chain, the one that bounds the step
            # (only where the caption layers have a chain of their own: otherwise that stream already carries them and bounds the forward pass)
            fuse_on_dec = (FUSE_ON_DEC and torch.is_grad_enabled() and fork_cap and Fn.PIPELINE_DECODER and not fused_after and cache is None
                           and cside is not side)
            if not fuse_on_dec:
                self._fuse(ft)
                if STM.ENABLED:
                    ft["encoded_ft"] = STM.through(ft["encoded_ft"], "fused", True)
            if cache is not None:
                cache.append({k: ft[k] for k in self._REASONING_KEYS if k in ft})
            # (training, several ranks) everything recorded from here on -- the value projections of layer l + 1, decoder layer l, the
            # iterations l + 1 .. -- is behind this mark: when its backward has run, the gradients of reasoning / caption layers >= l + 1 are final
            x = Fn.bucket_mark(x, l + 1)
            if values_ahead and l + 1 < len(self.layers):
                issue_values(l + 1)
            if fused_after:
                pass                             # inference: all decoder layers as one launch after the loop (they need only the cached results)
            elif fork_cap and Fn.PIPELINE_DECODER:
                # The decoder layer needs this layer's fused memory, the NEXT reasoning layer does not need the decoder
                # layer: it goes to the caption stream (ahead of the next caption layer) and runs under the next
                # layer's visual reasoning; the join at the end of that layer (or below) waits for it.
                side.wait_stream(main)
                Fn._keep_taken(x); Fn._keep_taken(ft.get("encoded_ft"))
                with torch.cuda.stream(side):
                    if fuse_on_dec:
                        self._fuse(ft)
                        if STM.ENABLED:
                            ft["encoded_ft"] = STM.through(ft["encoded_ft"], "fused", True)
                    x = STM.through(layer(b, ft, STM.through(x, "dec in", True)), "dec out", True)                                                      # :182
                dec_pending = side
            else:
                x = layer(b, ft, x)                                                          # :182
        if dec_pending is not None:
            torch.cuda.current_stream().wait_stream(dec_pending)
        if fused_after:
            ft["_bist_reasoning"] = cache
            x = self._decode_fused(b, ft, x)
            ft["_bist_fused_first"] = True       # (decode.py: position 0 of the turn is in the kernel's pools)
        ft.pop("_bist_vft_fan", None)
        ft.pop("_bist_fused_train", None)
        ft.pop("_bist_qmask2", None)
        ft.pop("_bist_vftp_fan", None)
        ft.pop("_bist_v_pre", None)
        ft.pop("_bist_pre_vid", None)
        if cache is not None:
            ft["_bist_reasoning"] = cache
        ft["decoded_text"] = STM.through(self.norm(x), "decoded_text")                        # :185
        if torch.is_grad_enabled():
            Fn.fan_set(ft, "decoded_text", 5)          # vocabulary logits, switch logits, one query projection per pointer attention
        return ft
