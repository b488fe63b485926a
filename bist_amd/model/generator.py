"""Log-probability heads on the HIP kernels (reference: model/generator.py).

The vocabulary projection is the MFMA GEMM against the shared embedding matrix (f32 output for
both compute dtypes), the softmax / copy-distribution scatter / mixture / log are one kernel
(``pointer_mix``).  No hard-coded ``.cuda()`` (the reference has two, generator.py:66,113).
"""
from __future__ import annotations

import os
from typing import Dict

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops

Tensor = torch.Tensor
UNK = 0


class Generator(nn.Module):
    """log_softmax(x . W^T) with the shared embedding matrix (reference: generator.py:11-27)."""

    def __init__(self, d_model, vocab, W=None):
        super().__init__()
        if W is not None:
            self.proj = W
            self.shared_W = True
        else:
            self.proj = nn.Linear(d_model, vocab)
            self.shared_W = False

    def forward(self, ft, batch, args, ft_key="decoded_text"):
        x = Fn.fan_take(ft, ft_key)                              # an alias of ft[ft_key] set aside for this consumer (one-pass gradient sum)
        if self.shared_W:
            logits = Fn.linear(x, self.proj, None, out_dtype=torch.float32)
        else:
            logits = Fn.linear(x, self.proj.weight, self.proj.bias, out_dtype=torch.float32)
        return Fn.log_softmax(logits).view(*x.shape[:-1], -1)


def _pointer_source(name: str, ft: Dict[str, Tensor], batch):
    """(token ids, (encoded text for the key projection, the same for the text vector), mask): the two readers of the encoded text get
    an alias each when the layer loop set a fan up (one-pass gradient sum)."""
    if name == "query":
        return batch.query, (Fn.fan_take(ft, "encoded_query"), Fn.fan_take(ft, "encoded_query")), batch.query_mask
    if name == "his":
        return batch.his, (Fn.fan_take(ft, "encoded_his"), Fn.fan_take(ft, "encoded_his")), batch.his_mask
    if name == "cap":
        return batch.cap, (Fn.fan_take(ft, "encoded_cap"), Fn.fan_take(ft, "encoded_cap")), batch.cap_mask
    raise ValueError("unknown pointer source %r" % name)


def _pointer_probs(attn, x: Tensor, enc: Tensor, mask: Tensor) -> Tensor:
    """The single-head pointer attention's probabilities [B,Lt,L] f32 (generator.py:109-110): only
    Q and K are projected -- the reference also computes V, P.V and the output projection and
    throws them away."""
    B, Lt, d = x.shape
    q = Fn.linear(x, attn.linears[0].weight, attn.linears[0].bias).view(B, Lt, d)
    k = Fn.linear(enc, attn.linears[1].weight, attn.linears[1].bias).view(B, enc.shape[1], d)
    _, p = Fn.mha_packed(q, k, k, "q_k_v", mask, 1, True)
    attn.attn = p
    return p.view(B, Lt, -1)


def _text_vector(p: Tensor, enc: Tensor) -> Tensor:
    """sum_t p[b,i,t] * enc[b,t,:]  (generator.py:117-118) as a batched GEMM; p f32 [B,Lt,L]."""
    B, Lt, L = p.shape
    d = enc.shape[-1]
    if not torch.is_grad_enabled() and p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and enc.is_contiguous() \
            and d % (8 if enc.dtype == torch.bfloat16 else 4) == 0 and d <= 1024:
        return ops.text_vector(p, enc)           # inference: one launch instead of a cast and a fallback-path GEMM (K = L is unaligned)
    return Fn.bmm_nn(Fn.cast(p, enc.dtype), enc)


def _pointer_head(attn, x: Tensor, enc_k: Tensor, enc_v: Tensor, mask: Tensor, text: Tensor, mask_unk: bool):
    """(p [B,Lt,L] f32, text vector [B,Lt,d]) of one pointer source (generator.py:104-118).  Inside its envelope: the two projections as one
    launch and the attention + text vector as one (Fn.pointer_attn: no value product, no separate mask operations); else the generic
    attention core and a product."""
    B, Lt, d = x.shape
    if Fn.pointer_attn_ok(x, enc_k, enc_v) and attn.h == 1:
        q, k = Fn.linear_pair(x, attn.linears[0].weight, attn.linears[0].bias, enc_k, attn.linears[1].weight, attn.linears[1].bias)
        p, tv = Fn.pointer_attn(q.view(B, Lt, d), k.view(B, -1, d), enc_v, mask, text if mask_unk else None, UNK)
        attn.attn = p.view(B, 1, Lt, -1)
        return p, tv
    if mask_unk:
        mask = mask & (text != UNK).unsqueeze(-2)                                             # generator.py:106-107
    p = _pointer_probs(attn, x, enc_k, mask)
    return p, _text_vector(p, enc_v)


def _switch_logits(lin: nn.Linear, parts) -> Tensor:
    """lin(cat(parts)) without the concat: each part multiplies its own column block."""
    d = parts[0].shape[-1]
    if Fn.switch_logits_ok(lin.weight, lin.bias, parts):
        return Fn.switch_logits(lin.weight, lin.bias, parts)          # one launch (three with its backward) instead of one product per part
    out = None
    for j, p in enumerate(parts):
        out = Fn.linear(p, Fn.column_block(lin.weight, j, d), lin.bias if j == 0 else None, out=out, accumulate=j > 0,
                        out_dtype=torch.float32)
    return out


class PointerGenerator(nn.Module):
    """Single copy source with a sigmoid switch (reference: generator.py:29-75)."""

    def __init__(self, d_model, vocab_gen, pointer_attn):
        super().__init__()
        self.vocab_gen = vocab_gen
        self.pointer_gen_W = nn.Linear(d_model * 3, 1)
        self.pointer_attn = pointer_attn
        pointer_attn.keep_attn = True

    def forward(self, ft, batch, args):
        x = ft["decoded_text"]
        B, Lt, d = x.shape
        if args.ptr_ft == "query+cap":
            raise NotImplementedError("ptr_ft='query+cap' is outside the hot path")
        text, (enc_k, enc_v), mask = _pointer_source(args.ptr_ft, ft, batch)
        logits = Fn.linear(Fn.fan_take(ft, "decoded_text"), self.vocab_gen, None, out_dtype=torch.float32)
        p, tv = _pointer_head(self.pointer_attn, Fn.fan_take(ft, "decoded_text"), enc_k, enc_v, mask, text, bool(args.mask_unk))
        sw = _switch_logits(self.pointer_gen_W, [Fn.fan_take(ft, "decoded_text"), tv, Fn.fan_take(ft, "encoded_tgt")])    # generator.py:71
        return Fn.pointer_mix(logits, sw, [p], [text], Lt, sigmoid_switch=True).view(B, Lt, -1)


class MultiPointerGenerator(nn.Module):
    """Several copy sources mixed by a softmax switch (reference: generator.py:77-127)."""

    def __init__(self, d_model, vocab_gen, pointer_attn, nb_pointer_ft):
        super().__init__()
        self.vocab_gen = vocab_gen
        self.pointer_gen_W = nn.Linear(d_model * (nb_pointer_ft + 2), nb_pointer_ft + 1)
        self.pointer_attn = pointer_attn
        for a in pointer_attn:
            a.keep_attn = True

    DECODE_FAST = os.environ.get("BIST_POINTER_DECODE", "1") != "0"      # tuning aid: 0 = the pointer heads of a decode step as separate launches

    def _turn_consts(self, b, ft, kv, names, mask_unk):
        """The per-TURN constants of the decode-step launch (bist_pointer_decode_mix_fwd) for the dialogue whose memories the decoder has
        just projected into ``kv`` (decoder.prepare_decode_cache calls this; row 0 of every tensor: all rows hold the same dialogue).
        Buffers are allocated once per dialogue geometry and rewritten in place: captured step graphs hold their addresses."""
        d = self.pointer_gen_W.weight.shape[1] // (len(names) + 2)
        ns = len(names) + 1
        key = (tuple(names), tuple(ft["encoded_" + n].shape[1] for n in names))
        g = kv.setdefault("gen_by_len", {}).get(key)
        if g is None:
            g = kv["gen_by_len"][key] = {"key": key, "src": []}
            for name in names:
                enc = ft["encoded_" + name]
                L, dev = enc.shape[1], enc.device
                f32 = lambda *shape: torch.zeros(*shape, device=dev, dtype=torch.float32)
                g["src"].append({"M": f32(L, d), "c": f32(L), "E": f32(L, ns), "mask": torch.zeros(L, device=dev, dtype=torch.uint8),
                                 "text": torch.zeros(L, device=dev, dtype=torch.long)})
        kv["gen"] = g
        for idx, name in enumerate(names):
            text, (enc, _), mask = _pointer_source(name, {"encoded_" + name: ft["encoded_" + name]}, b)
            sj, at, enc0 = g["src"][idx], self.pointer_attn[idx], enc[0]
            L = enc0.shape[0]
            m = mask[0].reshape(-1)
            if mask_unk:
                m = m & (text[0] != UNK)                                                   # generator.py:106-107
            sj["mask"].copy_(m)
            sj["text"].copy_(text[0])
            kp = Fn.linear(enc0, at.linears[1].weight, at.linears[1].bias)                 # the keys of generator.py:109, [L, d]
            wq = at.linears[0].weight
            ops.gemm(kp, wq, sj["M"], M=L, N=d, K=d, a_rs=d, a_ks=1, b_rs=1, b_ks=wq.stride(0), ldc=d)     # M = K W_q
            Fn.linear(kp, at.linears[0].bias.view(1, d), None, out=sj["c"].view(L, 1), out_dtype=torch.float32)    # c = K b_q
            Fn.linear(enc0, Fn.column_block(self.pointer_gen_W.weight, 2 + idx, d), None, out=sj["E"], out_dtype=torch.float32)
        g["stamp"] = kv["stamp"]
        return g

    def _forward_decode(self, ft, batch, args):
        """One decode step's heads for rows that share one dialogue, or None when the case is outside the launch's envelope."""
        tc = ft.get("_bist_turn_consts")
        x = ft["decoded_text"]
        tgt = ft.get("encoded_tgt")
        names = args.ptr_ft.split(",")
        if tc is None or not self.DECODE_FAST or torch.is_grad_enabled() or not x.is_cuda or tgt is None or tgt.shape != x.shape or x.shape[-1] > 1024 \
                or x.shape[-1] % 8 or len(names) > 3 or any(ft.get("encoded_" + n) is None or ft["encoded_" + n].shape[1] > 512 for n in names):
            return None
        dec, kv = tc
        B, Lt, d = x.shape
        hooks = dec.__dict__.setdefault("_bist_turn_hooks", {})
        key = (id(self), tuple(names), bool(args.mask_unk))
        if key not in hooks:                 # from now on the decoder refreshes the constants whenever it projects a turn's memories
            hooks[key] = lambda b_, ft_, kv_, names=names, mu=bool(args.mask_unk): self._turn_consts(b_, ft_, kv_, names, mu)
        g = kv.get("gen")
        if g is None or g.get("stamp") is not kv.get("stamp") or g["key"] != (tuple(names), tuple(ft["encoded_" + n].shape[1] for n in names)):
            g = self._turn_consts(batch, ft, kv, names, bool(args.mask_unk))
        logits = Fn.linear(x, self.vocab_gen, None, out_dtype=torch.float32)
        srcs = []
        for idx, sj in enumerate(g["src"]):
            s2 = dict(sj)
            if getattr(self.pointer_attn[idx], "keep_attn", False):
                s2["p"] = torch.empty((B * Lt, sj["text"].shape[0]), device=x.device, dtype=torch.float32)
                self.pointer_attn[idx].attn = s2["p"].view(B, 1, Lt, -1)
            srcs.append(s2)
        out = ops.pointer_decode_mix(x.reshape(B * Lt, d).contiguous(), tgt.reshape(B * Lt, d).contiguous(), logits.view(B * Lt, -1), srcs,
                                     self.pointer_gen_W.weight, self.pointer_gen_W.bias, 1.0 / (d ** 0.5))
        ft["_bist_ptr_fast"] = True
        return out.view(B, Lt, -1)

    def forward(self, ft, batch, args):
        x = ft["decoded_text"]
        B, Lt, d = x.shape
        fast = self._forward_decode(ft, batch, args) if "_bist_turn_consts" in ft else None
        if fast is not None:
            return fast
        logits = Fn.linear(Fn.fan_take(ft, "decoded_text"), self.vocab_gen, None, out_dtype=torch.float32)
        ps, texts, vec = [], [], [Fn.fan_take(ft, "decoded_text"), Fn.fan_take(ft, "encoded_tgt")]     # generator.py:92
        for idx, name in enumerate(args.ptr_ft.split(",")):
            text, (enc_k, enc_v), mask = _pointer_source(name, ft, batch)
            p, tv = _pointer_head(self.pointer_attn[idx], Fn.fan_take(ft, "decoded_text"), enc_k, enc_v, mask, text, bool(args.mask_unk))
            ps.append(p); texts.append(text)
            vec.append(tv)
        sw = _switch_logits(self.pointer_gen_W, vec)
        return Fn.pointer_mix(logits, sw, ps, texts, Lt).view(B, Lt, -1)
