"""Transformer primitives of the hot path on the HIP kernels.

Same class names, constructor signatures, parameter names (``a_2``/``b_2``, ``linears.{0..3}``,
``w_1``/``w_2``, ``lut``) and ``forward`` signatures as the reference's model/modules.py, so
``state_dict``s and whole-module pickles are interchangeable; the arithmetic is dispatched to
libbist_hip.so through ``bist_amd.functional`` (no nn.Linear/F.softmax/torch.matmul call).
"""
from __future__ import annotations

import copy
import math
from typing import Optional

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops

Tensor = torch.Tensor


def clones(module: nn.Module, n: int) -> nn.ModuleList:
    """n independent deep copies (reference: modules.py:10-12)."""
    return nn.ModuleList(copy.deepcopy(module) for _ in range(n))


class LayerNorm(nn.Module):
    """a_2 * (x - mean) / (std_unbiased + eps) + b_2   (reference: modules.py:20-31)."""

    def __init__(self, features: int, eps: float = 1e-6):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(features))
        self.b_2 = nn.Parameter(torch.zeros(features))
        self.eps = eps

    def forward(self, x: Tensor) -> Tensor:
        return Fn.layernorm(x, self.a_2, self.b_2, self.eps)

    def with_residual(self, x: Tensor, lazy: bool = False):
        """(LN(x), x'): use x' as the residual operand of the sublayer's last GEMM (see Fn.layernorm_res).
        lazy: LN(x) goes to exactly one ``Fn.linear`` and nowhere else -- it then runs as that projection's prologue."""
        return Fn.layernorm_res(x, self.a_2, self.b_2, self.eps, lazy=lazy)


class SublayerConnection(nn.Module):
    """x + dropout(sublayer(norm(x)))   (reference: modules.py:33-44).

    The layer classes of encoder.py/decoder.py fuse the residual add (and dropout) into the
    last GEMM of the sublayer and only borrow ``self.norm``; this generic form stays for callers
    that pass an arbitrary ``sublayer`` callable.
    """

    def __init__(self, size: int, dropout: float):
        super().__init__()
        self.norm = LayerNorm(size)
        self.dropout = nn.Dropout(dropout)   # holds p; the mask itself is generated inside the kernels

    @property
    def p(self) -> float:
        """drop probability (a property, not an attribute: a whole-module pickle written by the reference has only ``dropout``)"""
        return float(self.dropout.p)

    def forward(self, x: Tensor, sublayer) -> Tensor:
        y = sublayer(self.norm(x))
        if self.training and self.p > 0:
            return Fn.add_dropout(x, y, (self.p, Fn.next_seed("sub", self)), 0)
        return Fn.add(x, y)


class MultiHeadedAttention(nn.Module):
    """Multi-head attention with the reference's masking rule (modules.py:54-100).

    ``self.attn`` (the probabilities, [N,h,Lq,Lk] f32) is only materialised when
    ``self.keep_attn`` is set: the pointer generators are its only readers (generator.py:62-63,
    109-110) and writing it for every attention on the path would be pure HBM traffic.
    """

    keep_attn = False      # class default (instances unpickled from a reference checkpoint have no such attribute)

    def __init__(self, h: int, d_model: int, d_in: int = -1, dropout: float = 0.1):
        super().__init__()
        assert d_model % h == 0
        self.d_k = d_model // h
        self.h = h
        if d_in < 0:
            d_in = d_model
        self.linears = clones(nn.Linear(d_in, d_model), 3)
        self.linears.append(nn.Linear(d_model, d_in))
        self.attn: Optional[Tensor] = None
        self.dropout = nn.Dropout(p=dropout)

    # -- packed projection weights (device-side concatenation, cached while parameters are unchanged)
    def _packed(self, idx):
        pk = self.__dict__.get("_pk")
        if pk is not None:               # bist_amd.train.Trainer laid the projections out contiguously:
            return pk[tuple(idx)]        # the packed operands (and their gradients) are plain views
        ws = [self.linears[i].weight for i in idx]
        bs = [self.linears[i].bias for i in idx]
        key = (tuple(idx),) + ops.weights_key(*ws, *bs)
        cache = self.__dict__.setdefault("_pack_cache", {})
        hit = cache.get(tuple(idx))
        if hit is not None and hit[0] == key and not torch.is_grad_enabled():
            return hit[1], hit[2]
        w, b = Fn.pack_rows(*ws), Fn.pack_rows(*bs)
        if not torch.is_grad_enabled():
            if hit is not None and hit[1].shape == w.shape and hit[1].device == w.device and hit[1].dtype == w.dtype:
                hit[1].copy_(w); hit[2].copy_(b)            # same buffers: captured hipGraphs keep reading them
                w, b = hit[1], hit[2]
            cache[tuple(idx)] = (key, w, b)
        return w, b

    def context(self, query: Tensor, key: Tensor, value: Tensor, mask: Optional[Tensor], want_p: Optional[bool] = None) -> Tensor:
        """Head-concatenated attention output BEFORE the output projection, [N,Lq,d].

        The Q/K/V projections are as few GEMMs as the aliasing of the arguments allows (one packed
        [N,L,3d] output for self-attention, q + packed [N,Lk,2d] for cross-attention); the attention
        core reads them in place as column views.  In training the probabilities are dropped inside the core
        (modules.py:62-63; self.dropout holds p) -- ``self.attn`` keeps the probabilities BEFORE dropout, its only
        readers being the pointer attentions, which are built with dropout=0 (mtn.py:89-92)."""
        d = self.h * self.d_k
        n, lq = query.shape[0], query.shape[1]
        keep = self.keep_attn if want_p is None else want_p
        if query is key and key is value:
            w, b = self._packed((0, 1, 2))
            qkv = Fn.linear(query, w, b).view(n, lq, 3 * d)
            ctx, p = Fn.mha_packed(qkv, None, None, "qkv", mask, self.h, keep, Fn.attn_drop(self))
        else:
            lk = key.shape[1]
            if key is value:
                w, b = self._packed((1, 2))
                q, kv = Fn.linear_pair(query, self.linears[0].weight, self.linears[0].bias, key, w, b)      # one launch
                q, kv = q.view(n, lq, d), kv.view(n, lk, 2 * d)
                ctx, p = Fn.mha_packed(q, kv, None, "q_kv", mask, self.h, keep, Fn.attn_drop(self))
            else:
                q = Fn.linear(query, self.linears[0].weight, self.linears[0].bias).view(n, lq, d)
                k = Fn.linear(key, self.linears[1].weight, self.linears[1].bias).view(n, lk, d)
                v = Fn.linear(value, self.linears[2].weight, self.linears[2].bias).view(n, lk, d)
                ctx, p = Fn.mha_packed(q, k, v, "q_k_v", mask, self.h, keep, Fn.attn_drop(self))
        self.attn = p
        return ctx

    def forward(self, query: Tensor, key: Tensor, value: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        """The reference's call form: like modules.py:94 it always leaves the probabilities in ``self.attn`` (the fused layer
        classes call ``context`` instead and skip that HBM write unless ``keep_attn`` is set)."""
        ctx = self.context(query, key, value, mask, want_p=True)
        out = Fn.linear(ctx, self.linears[3].weight, self.linears[3].bias)
        return out.view(query.shape[0], query.shape[1], -1)


class PositionwiseFeedForward(nn.Module):
    """w_2(relu(w_1 x))   (reference: modules.py:102-113); ReLU is the first GEMM's epilogue."""

    def __init__(self, d_model: int, d_ff: int, dropout: float = 0.1, d_out: int = -1):
        super().__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        if d_out < 0:
            d_out = d_model
        self.w_2 = nn.Linear(d_ff, d_out)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: Tensor, residual: Optional[Tensor] = None, out_drop: Optional[dict] = None) -> Tensor:
        inner = {}
        if self.training and self.dropout.p > 0:          # dropout(relu(w_1 x)), modules.py:113
            inner = {"drop_p": float(self.dropout.p), "drop_seed": Fn.next_seed("ffn", self)}
        hdn = Fn.linear(x, self.w_1.weight, self.w_1.bias, act=Fn.ACT_RELU, **inner)
        return Fn.linear(hdn, self.w_2.weight, self.w_2.bias, residual=residual, out_shape=(*x.shape[:-1], self.w_2.weight.shape[0]),
                         **(out_drop or {}))


class Embeddings(nn.Module):
    """lut(x) * sqrt(d_model)   (reference: modules.py:115-123)."""

    def __init__(self, d_model: int, vocab: int):
        super().__init__()
        self.lut = nn.Embedding(vocab, d_model)
        self.d_model = d_model

    def forward(self, x):
        if x is None:
            return x
        zero = torch.zeros((x.shape[1], self.d_model), device=self.lut.weight.device, dtype=torch.float32)
        return Fn.embed_pe(x, self.lut.weight, zero)


class PositionalEncoding(nn.Module):
    """x + pe[:, :L]   (reference: modules.py:125-144); ``pe`` is the same registered buffer."""

    def __init__(self, d_model: int, dropout: float, max_len: int = 5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0.0, max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0.0, d_model, 2) * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))

    def table(self) -> Tensor:
        """f32 [max_len, d] table.  If the module was cast to bf16 the buffer lost precision, so the
        f32 table is rebuilt from the closed form (same expression as __init__) and cached."""
        pe = self.pe[0]
        if pe.dtype == torch.float32:
            return pe
        cached = self.__dict__.get("_pe_f32")
        if cached is None or cached.device != pe.device:
            max_len, d_model = pe.shape
            t = torch.zeros(max_len, d_model)
            position = torch.arange(0.0, max_len).unsqueeze(1)
            div_term = torch.exp(torch.arange(0.0, d_model, 2) * -(math.log(10000.0) / d_model))
            t[:, 0::2] = torch.sin(position * div_term)
            t[:, 1::2] = torch.cos(position * div_term)
            cached = t.to(pe.device)
            self.__dict__["_pe_f32"] = cached
        return cached

    def forward(self, x):
        if x is None:
            return x
        L = x.shape[1]
        pe = self.table()[:L].to(x.dtype).contiguous()
        if self.training and self.dropout.p > 0:
            return Fn.add_dropout(x, pe, (float(self.dropout.p), Fn.next_seed("pe", self)), 1)
        return Fn.add(x, pe)


def embed_with_position(seq: nn.Sequential, ids: Tensor, pos0: int = 0) -> Tensor:
    """Fused ``nn.Sequential(Embeddings, PositionalEncoding)`` (mtn.py:79-82) in one kernel; the ids are positions pos0 .. of their sequences."""
    emb, pos = seq[0], seq[1]
    drop = None
    if pos.training and pos.dropout.p > 0:                 # dropout after the position is added (modules.py:144)
        drop = (float(pos.dropout.p), Fn.next_seed("pe", pos))
    return Fn.embed_pe(ids, emb.lut.weight, pos.table()[pos0:] if pos0 else pos.table(), drop)
