"""Composite steps of the hot path, expressed on the kernels of libbist_hip.so.

The algebra used here (checked against the oracle in tests/):
  * the key projection is folded into the query:  (Q_h W_k,h) X^T == Q_h (X W_k,h^T)^T and the key
    bias is constant along the softmax axis, so it cancels -- K is never materialised;
  * in stage 2 the value projection is applied AFTER the weighted sum:
    P (Y W_v^T + b_v) == (P Y) W_v^T + b_v because each row of P sums to one.
Every function takes and returns device tensors; nothing here touches the CPU.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import ops
from .ops import ACT_NONE, ACT_RELU

Tensor = torch.Tensor

linear = ops.linear
layernorm = ops.layernorm
mha_core = ops.mha_core
st_stage1_pv = ops.st_stage1_pv
st_stage2 = ops.st_stage2
embed_pe = ops.embed_pe
fuse_modalities = ops.fuse_modalities
add = ops.add


def head_fold(q: Tensor, wk: Tensor, h: int, alpha: float) -> Tensor:
    """Qf[m, hh*d + n] = alpha * sum_c q[m, hh*dk + c] * wk[hh*dk + c, n]     q [M,d], wk [d,d] -> [M, h*d].

    One batched GEMM over the heads (B operand read "NN": row stride 1, k stride d)."""
    M, d = q.shape
    dk = d // h
    out = torch.empty((M, h * d), device=q.device, dtype=q.dtype)
    ops.gemm(q, wk, out, M=M, N=d, K=dk, a_rs=q.stride(0), a_ks=1, b_rs=1, b_ks=wk.stride(0), ldc=h * d,
             batch=(1, h), a_bs=(0, dk), b_bs=(0, dk * wk.stride(0)), c_bs=(0, d), alpha=alpha)
    return out


def head_unfold(py: Tensor, wv: Tensor, bv: Tensor, h: int) -> Tensor:
    """O[m, hh*dk + c] = sum_n py[m, hh*d + n] * wv[hh*dk + c, n] + bv[hh*dk + c]    py [M,h*d] -> [M,d]."""
    M = py.shape[0]
    d = wv.shape[1]
    dk = d // h
    out = torch.empty((M, d), device=py.device, dtype=py.dtype)
    ops.gemm(py, wv, out, M=M, N=dk, K=d, a_rs=py.stride(0), b_rs=wv.stride(0), ldc=d, bias=bv,
             batch=(1, h), a_bs=(0, d), b_bs=(0, dk * wv.stride(0)), c_bs=(0, dk), bias_bs2=dk)
    return out


def st_scores(qf: Tensor, vft: Tensor) -> Tensor:
    """scores[b, r, ts] = qf[b, r, :] . vft[b, ts, :]   qf [B,R,d], vft [B,TS,d] -> f32 [B,R,TS]."""
    B, R, d = qf.shape
    TS = vft.shape[1]
    out = torch.empty((B, R, TS), device=qf.device, dtype=torch.float32)
    ops.gemm(qf, vft, out, M=R, N=TS, K=d, a_rs=qf.stride(1), b_rs=vft.stride(1), ldc=TS, batch=(B, 1),
             a_bs=(qf.stride(0), 0), b_bs=(vft.stride(0), 0), c_bs=(R * TS, 0))
    return out


def pack_rows(*ws: Tensor) -> Tensor:
    """Concatenate weight matrices / biases row-wise (device-side data movement only)."""
    return torch.cat(ws, dim=0)
