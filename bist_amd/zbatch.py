"""The two directions of a reasoning layer as ONE sequence of launches (reference: model/encoder.py:172-188).

``VidEncoderLayer4.forward`` runs A0 -> A1 -> A2 -> F0 on the t2s stream of the query and A3 -> A4 -> A5 -> F1 on the s2t stream:
the same sixteen small operations on same-shaped [B, Lq, d] tensors with different weights.  Issued as two chains they are two
queues of latency-bound launches (each ~5 us whatever it computes) that overlap poorly inside a replayed hipGraph; here every
query-side operation is issued ONCE for both directions over a stacked [2, B, Lq, d] tensor ("z" = direction index):

  * LayerNorm            bist_layernorm_fwd_multi / _bwd_multi: one launch, set z = (rows of z, gain_z, offset_z);
  * projections          bist_gemm with batch1 = 2: operand / bias / output / residual strides of the outer batch index are the
                         DISTANCES between the two directions' tensors (two parameter tensors are always "at a constant stride");
                         the two weight gradients accumulate into the trainer's flat gradient the same way;
  * fold / un-fold       the per-head batched products with batch = (2, h);
  * attention cores      small self-attention: the stacked batch as it is (2B sequences); stage 1 and stage 2 stay per direction
                         (different group / key counts) inside ONE autograd node each (ZStage1Fn, ZStage2Fn), which read and write
                         the halves of the stacked tensors in place -- nothing is copied to stack or unstack.

Every function here has a hand-written backward on the same kernels; the hand-offs of bist_amd/autograd.py that save launches are
kept (residual gradient folded into the LayerNorm backward, dropout-masked gradient from the LayerNorm backward to the producing
product, the gate of drop(relu(.)) in the consuming product's dX epilogue).  Every intermediate has exactly ONE consumer, so
autograd never launches an accumulation of its own.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence, Tuple

import torch
from torch.autograd import Function

from . import autograd as ag
from . import ops
from ._lib import ACT_GATE, ACT_NONE, ACT_RELU, check, lib
from .ops import _ptr, _stream, dtype_code

Tensor = torch.Tensor

# Measured (round 3, BASELINE configs[1], same box, `python bench.py`): the lock-step form is 1 143 launches and 14.1 ms of summed kernel
# time per training step against 1 397 launches / 16.2-16.7 ms for the two-chain form -- and 12.1-12.4 ms per step against 11.7-11.8:
# with both directions on ONE stream the two directions' chip-filling stage-1 launches run back to back, while the two-chain form
# overlaps one direction's big launches with the other's small ones (1.38 kernels in flight on average against 1.16).  Under rocprofv3
# (which serialises part of that overlap) the lock-step form is the faster one (13.7 vs 14.6 ms).  So it is OPT-IN: BIST_ZBATCH=1.
ENABLED = os.environ.get("BIST_ZBATCH", "0") != "0"
# tuning aid (lock-step form): the s2t direction's stage-1 / stage-2 launches on a side stream beside the t2s direction's, and the query side
# of stage 2 under the stage-1 launches.  Measured 13.4 vs 12.35 ms per step: every fork / join is a cross-queue dependency in the replayed
# hipGraph (~10 us on the critical path each), three per layer and pass cost more than the overlap returns.  Off by default.
DIR_STREAMS = os.environ.get("BIST_Z_DIR_STREAMS", "0") != "0"


class _DirStreams:
    """fork: the side stream waits for the main stream; `on(z)`: context of direction z's launches; join: main waits for side.
    Every edge is main <-> side (an edge between two side streams crashes hipGraph capture in the HIP runtime)."""

    def __init__(self, enabled: bool):
        from . import functional as Fn
        self.main = self.side = None
        if enabled and DIR_STREAMS and Fn.CONCURRENT:
            self.main, self.side = torch.cuda.current_stream(), Fn.side_stream(0)
            self.side.wait_stream(self.main)

    def on(self, z: int):
        import contextlib
        return torch.cuda.stream(self.side) if (z == 1 and self.side is not None) else contextlib.nullcontext()

    def join(self):
        if self.side is not None:
            self.main.wait_stream(self.side)


def zstride(p0: Tensor, p1: Tensor) -> int:
    """Distance p1 - p0 in ELEMENTS between two same-shaped, same-strided tensors (the stride of the outer batch index)."""
    if p0.shape != p1.shape or p0.dtype != p1.dtype or p0.stride() != p1.stride() or p0.device != p1.device:
        raise ValueError("bist_amd.zbatch: the two directions' operands must share shape, strides, dtype and device")
    diff, es = p1.data_ptr() - p0.data_ptr(), p0.element_size()
    if diff % es:
        raise ValueError("bist_amd.zbatch: operands are not a whole number of elements apart")
    return diff // es


def _flat2(x: Tensor) -> Tensor:
    """[2, ..., c] stacked tensor -> contiguous [2*M, c] view."""
    c = x.shape[-1]
    x2 = x.reshape(-1, c)
    return x2 if x2.is_contiguous() else x2.contiguous()


# ----------------------------------------------------------------------------------------------
# LayerNorm of both directions (+ the pass-through of x for the sublayer's residual add)
# ----------------------------------------------------------------------------------------------
class ZLayerNormResFn(Function):
    """(LN_z(x[z]), x) for the stacked x [2, ..., d] with parameters (a0, b0), (a1, b1).  The second output is x itself (the
    residual operand of the sublayer's last product); its gradient comes back here and is added inside the backward kernel."""

    @staticmethod
    def forward(ctx, x, a0, b0, a1, b1, eps, up_drop=None):
        d = x.shape[-1]
        x2 = _flat2(x)
        M = x2.shape[0] // 2
        y = torch.empty(x2.shape, device=x.device, dtype=x.dtype)
        ops.layernorm_multi([x2[:M], x2[M:]], [a0, a1], [b0, b1], [y[:M], y[M:]], eps)
        ctx.save_for_backward(x2, a0, a1)
        ctx.cfg = (eps, b0.dtype, tuple(x.shape), M, d)
        ctx.up_drop = up_drop
        ctx.dst = [(getattr(a, "_acc32", None), getattr(b, "_acc32", None)) for a, b in ((a0, b0), (a1, b1))]
        ctx.set_materialize_grads(False)
        return y.view(x.shape), x

    @staticmethod
    def backward(ctx, dy, dres):
        x2, a0, a1 = ctx.saved_tensors
        eps, bdt, x_shape, M, d = ctx.cfg
        if dy is None:
            dy = torch.zeros(x2.shape, device=x2.device, dtype=x2.dtype)
        dy2 = _flat2(dy)
        add2 = None
        if dres is not None:
            add2 = _flat2(dres)
            if add2.dtype != x2.dtype:
                add2 = add2.to(x2.dtype)
        dx = torch.empty(x2.shape, device=x2.device, dtype=x2.dtype)
        direct = all(da is not None and db is not None for da, db in ctx.dst)
        if direct:
            dab = [ctx.dst[0], ctx.dst[1]]
        else:
            acc = torch.zeros((2, 2, d), device=x2.device, dtype=torch.float32)
            dab = [(acc[0, 0], acc[0, 1]), (acc[1, 0], acc[1, 1])]
        gains = (a0, a1)
        defer = (direct and ops.LNGRAD_QUEUE is not None and d * x2.element_size() == 1024
                 and all(t.data_ptr() % 16 == 0 for t in (dy2, x2, a0, a1) + ((add2,) if add2 is not None else ()))
                 and (M * d * x2.element_size()) % 16 == 0)
        up = ctx.up_drop
        dz = torch.empty(x2.shape, device=x2.device, dtype=x2.dtype) if (up is not None and up[2] == d) else None
        zdrop = C.byref(ops.BistDrop(up[0], up[1] & 0xFFFFFFFFFFFFFFFF, _ptr(ops.DROP_CTR))) if dz is not None else None
        sets = []
        for z in range(2):
            lo, hi = z * M, (z + 1) * M
            sets.append((dy2[lo:hi], x2[lo:hi], gains[z], dx[lo:hi], None if defer else dab[z][0], None if defer else dab[z][1],
                         add2[lo:hi] if add2 is not None else None, dz[lo:hi] if dz is not None else None, lo))
        ops.layernorm_bwd_multi(sets, M, d, dy2.stride(0), x2.stride(0), d, eps, add2.stride(0) if add2 is not None else 0, zdrop, x2.dtype)
        if defer:
            for z in range(2):
                ops.LNGRAD_QUEUE.append((dy2[z * M:(z + 1) * M], x2[z * M:(z + 1) * M], gains[z], dab[z][0], dab[z][1], eps))
        gx = dx.view(x_shape)
        if dz is not None:
            gx._bist_dz = (dz, up[0], up[1])          # survives only if autograd hands THIS tensor to the producer's backward
        if direct:
            return gx, None, None, None, None, None, None
        cast = ag._to_dtype_from_f32
        return gx, cast(dab[0][0], a0.dtype), cast(dab[0][1], bdt), cast(dab[1][0], a1.dtype), cast(dab[1][1], bdt), None, None


def layernorm_res(x: Tensor, n0, n1):
    """n0 / n1: the two directions' LayerNorm modules (a_2, b_2, eps).  -> (LN_z(x), x')"""
    return ZLayerNormResFn.apply(x, n0.a_2, n0.b_2, n1.a_2, n1.b_2, n0.eps, getattr(x, "_bist_drop", None))


# ----------------------------------------------------------------------------------------------
# nn.Linear of both directions
# ----------------------------------------------------------------------------------------------
class ZLinearFn(Function):
    """y[z] = drop(act(x[z] . W_z^T + b_z)) + residual[z]   for the stacked x [2M, K] -> [2M, N] in one bist_gemm (batch1 = 2)."""

    @staticmethod
    def forward(ctx, x, w0, w1, b0, b1, residual, act, drop_p, drop_seed, out_shape):
        K = x.shape[-1]
        x2 = _flat2(x)
        M, N = x2.shape[0] // 2, w0.shape[0]
        wzs = zstride(w0, w1)
        y = torch.empty((2 * M, N), device=x.device, dtype=x.dtype)
        r2 = None
        if residual is not None:
            r2 = _flat2(residual)
            if r2.shape != (2 * M, N):
                raise ValueError("bist_amd.zbatch.linear: the residual must be the stacked [2, M, N] tensor")
        g = ops.gemm_desc(x2, w0, y, M=M, N=N, K=K, a_rs=x2.stride(0), b_rs=w0.stride(0), ldc=N, bias=b0, residual=r2,
                          ldr=r2.stride(0) if r2 is not None else 0, act=act, batch=(2, 1), a_bs=(M * x2.stride(0), 0), b_bs=(wzs, 0),
                          c_bs=(M * N, 0), r_bs=((M * r2.stride(0)) if r2 is not None else 0, 0),
                          bias_bs1=zstride(b0, b1) if b0 is not None else 0, drop_p=drop_p, drop_seed=drop_seed)
        check(lib.bist_gemm(C.byref(g), _stream()), "bist_gemm")
        ctx.save_for_backward(x2, w0, w1, y if act == ACT_RELU else None)
        ctx.cfg = (act, drop_p, drop_seed, b0 is not None, b0.dtype if b0 is not None else None, residual is not None,
                   tuple(residual.shape) if residual is not None else None, tuple(x.shape), M, N, K)
        ctx.w_dst = (getattr(w0, "_grad_view", None), getattr(w1, "_grad_view", None))
        ctx.b_dst = (getattr(b0, "_acc32", None), getattr(b1, "_acc32", None)) if b0 is not None else (None, None)
        gate = getattr(x, "_bist_gate", None)             # x = drop(relu(.)) of the producing z-linear, saved here as x2
        ctx.gate = gate if (gate is not None and x.dim() == 2 and x.is_contiguous()) else None
        return y if out_shape is None else y.view(out_shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w0, w1, y = ctx.saved_tensors
        act, drop_p, drop_seed, has_bias, bias_dtype, has_res, res_shape, x_shape, M, N, K = ctx.cfg
        pre = getattr(dy, "_bist_dz", None)
        dy = dy.reshape(2 * M, N)
        if not dy.is_contiguous():
            dy = dy.contiguous()
        dres = dy.view(res_shape) if (has_res and ctx.needs_input_grad[5]) else None
        dz = dy if dy.dtype == x2.dtype else ops.cast(dy, x2.dtype)
        tag = getattr(ctx, "gate_tag", None)
        handed = (pre is not None and len(pre) == 4 and act == ACT_RELU and pre[1:3] == (float(drop_p), int(drop_seed))
                  and pre[0].numel() == 2 * M * N and pre[0].dtype == x2.dtype)
        if tag is not None and tag.gated and not handed:
            raise RuntimeError("bist_amd.zbatch: the gated gradient of a drop(relu(.)) output did not reach its producer")
        if handed:
            dz = pre[0].view(2 * M, N)
        elif pre is not None and len(pre) == 3 and act == ACT_NONE and pre[1:] == (float(drop_p), int(drop_seed)) and pre[0].numel() == 2 * M * N \
                and pre[0].dtype == x2.dtype:
            dz = pre[0].view(2 * M, N)                 # already masked by the LayerNorm backward that produced dy
        elif act == ACT_RELU or drop_p > 0:
            dz2 = torch.empty_like(dz)
            yy = y if y is not None else dz
            check(lib.bist_epilogue_bwd(dz.data_ptr(), yy.data_ptr(), dz2.data_ptr(), 2 * M, N, N, N, N, act, drop_p, drop_seed,
                                        _ptr(ops.DROP_CTR) if drop_p > 0 else None, dtype_code(dz.dtype), _stream()), "bist_epilogue_bwd")
            dz = dz2
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        wzs = zstride(w0, w1)
        dx = dws = None
        g_dx = g_dw = None
        if need_dx:
            dx = torch.empty((2 * M, K), device=dz.device, dtype=x2.dtype)
            kw = dict(M=M, N=K, K=N, a_rs=N, a_ks=1, b_rs=1, b_ks=w0.stride(0), ldc=K, batch=(2, 1), a_bs=(M * N, 0), b_bs=(wzs, 0), c_bs=(M * K, 0))
            if ctx.gate is not None:
                g_dx = ops.gemm_desc(dz, w0, dx, alpha=1.0 / (1.0 - ctx.gate[0]), act=ACT_GATE, residual=x2, ldr=x2.stride(0),
                                     r_bs=(M * x2.stride(0), 0), **kw)
            else:
                g_dx = ops.gemm_desc(dz, w0, dx, **kw)
        direct = ctx.w_dst[0] is not None and ctx.w_dst[1] is not None
        if need_dw:
            kw = dict(M=N, N=K, K=M, a_rs=1, a_ks=N, b_rs=1, b_ks=x2.stride(0), batch=(2, 1), a_bs=(M * N, 0), b_bs=(M * x2.stride(0), 0))
            if direct:
                gv0 = ctx.w_dst[0]
                gzs = zstride(gv0, ctx.w_dst[1])
                g_dw = ops.gemm_desc(dz, x2, gv0, ldc=gv0.stride(0), c_bs=(gzs, 0), residual=gv0, ldr=gv0.stride(0), r_bs=(gzs, 0), **kw)
            else:
                dws = torch.empty((2, N, K), device=dz.device, dtype=w0.dtype)
                g_dw = ops.gemm_desc(dz, x2, dws, ldc=K, c_bs=(N * K, 0), **kw)
        if g_dx is not None and g_dw is not None:
            ops.gemm_pair(g_dx, g_dw)
        elif g_dx is not None:
            check(lib.bist_gemm(C.byref(g_dx), _stream()), "bist_gemm")
        elif g_dw is not None:
            check(lib.bist_gemm(C.byref(g_dw), _stream()), "bist_gemm")
        db0 = db1 = None
        if has_bias and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]):
            outs = []
            for z in range(2):
                acc = ctx.b_dst[z] if ctx.b_dst[z] is not None else ag._f32_zeros((N,), dz)
                part = dz[z * M:(z + 1) * M]
                if ctx.b_dst[z] is not None and ops.COLSUM_QUEUE is not None:
                    ops.COLSUM_QUEUE.append((part, acc, M, N))
                else:
                    check(lib.bist_col_sum_acc(part.data_ptr(), acc.data_ptr(), M, N, N, dtype_code(dz.dtype), _stream()), "bist_col_sum_acc")
                outs.append(None if ctx.b_dst[z] is not None else ag._to_dtype_from_f32(acc, bias_dtype))
            db0, db1 = outs
        if dx is not None:
            dx = dx.view(x_shape)
            if ctx.gate is not None:
                dx._bist_dz = (dx, ctx.gate[0], ctx.gate[1], "gate")
                ctx.gate.gated += 1
        dw0, dw1 = (dws[0], dws[1]) if dws is not None else (None, None)
        return dx, dw0, dw1, db0, db1, dres, None, None, None, None


def linear(x: Tensor, lin0, lin1, *, act: int = ACT_NONE, residual: Optional[Tensor] = None, drop_p: float = 0.0, drop_seed: int = 0,
           out_shape=None) -> Tensor:
    """lin0 / lin1: (weight, bias) of the two directions' nn.Linear (any tensors: parameter views of a packed projection too)."""
    (w0, b0), (w1, b1) = lin0, lin1
    y = ZLinearFn.apply(x, w0, w1, b0, b1, residual, act, drop_p, drop_seed, tuple(out_shape) if out_shape is not None else None)
    if drop_p > 0 and act == ACT_NONE:
        y._bist_drop = (float(drop_p), int(drop_seed), w0.shape[0])       # a LayerNorm that consumes y hands the masked gradient back
    if act == ACT_RELU and residual is None and out_shape is None and ag.GATE_HANDOFF and torch.is_grad_enabled():
        y._bist_gate = ag.GateTag(drop_p, drop_seed)
        if y.grad_fn is not None:
            y.grad_fn.gate_tag = y._bist_gate
    return y


# ----------------------------------------------------------------------------------------------
# per-head fold / un-fold of both directions
# ----------------------------------------------------------------------------------------------
class ZHeadFoldFn(Function):
    """Qf[z, m, hh*d+n] = alpha * sum_c q[z, m, hh*dk+c] wk_z[hh*dk+c, n]:   q [2M, d] -> [2M, h*d]."""

    @staticmethod
    def forward(ctx, q, wk0, wk1, h, alpha):
        q = _flat2(q)
        M, d = q.shape[0] // 2, q.shape[1]
        dk = d // h
        wzs = zstride(wk0, wk1)
        ld = wk0.stride(0)
        out = torch.empty((2 * M, h * d), device=q.device, dtype=q.dtype)
        ops.gemm(q, wk0, out, M=M, N=d, K=dk, a_rs=q.stride(0), a_ks=1, b_rs=1, b_ks=ld, ldc=h * d, batch=(2, h),
                 a_bs=(M * q.stride(0), dk), b_bs=(wzs, dk * ld), c_bs=(M * h * d, d), alpha=alpha)
        ctx.save_for_backward(q, wk0, wk1)
        ctx.cfg = (h, alpha, M, d)
        ctx.w_dst = (getattr(wk0, "_grad_view", None), getattr(wk1, "_grad_view", None))
        return out

    @staticmethod
    def backward(ctx, dqf):
        q, wk0, wk1 = ctx.saved_tensors
        h, alpha, M, d = ctx.cfg
        dk = d // h
        dqf = dqf.reshape(2 * M, h * d)
        if not dqf.is_contiguous():
            dqf = dqf.contiguous()
        wzs, ld = zstride(wk0, wk1), wk0.stride(0)
        dq = torch.empty((2 * M, d), device=q.device, dtype=q.dtype)
        g_dq = ops.gemm_desc(dqf, wk0, dq, M=M, N=dk, K=d, a_rs=h * d, b_rs=ld, ldc=d, batch=(2, h), a_bs=(M * h * d, d),
                             b_bs=(wzs, dk * ld), c_bs=(M * d, dk), alpha=alpha)
        kw = dict(M=dk, N=d, K=M, a_rs=1, a_ks=q.stride(0), b_rs=1, b_ks=h * d, batch=(2, h), a_bs=(M * q.stride(0), dk), b_bs=(M * h * d, d), alpha=alpha)
        if ctx.w_dst[0] is not None and ctx.w_dst[1] is not None:
            gv = ctx.w_dst[0]
            gzs = zstride(gv, ctx.w_dst[1])
            g_dw = ops.gemm_desc(q, dqf, gv, ldc=gv.stride(0), c_bs=(gzs, dk * gv.stride(0)), residual=gv, ldr=gv.stride(0),
                                 r_bs=(gzs, dk * gv.stride(0)), **kw)
            ops.gemm_pair(g_dq, g_dw)
            return dq, None, None, None, None
        dws = torch.empty((2, d, d), device=q.device, dtype=wk0.dtype)
        g_dw = ops.gemm_desc(q, dqf, dws, ldc=d, c_bs=(d * d, dk * d), **kw)
        ops.gemm_pair(g_dq, g_dw)
        return dq, dws[0], dws[1], None, None


def head_fold(q: Tensor, wk0: Tensor, wk1: Tensor, h: int, alpha: float) -> Tensor:
    return ZHeadFoldFn.apply(q, wk0, wk1, h, alpha)


class ZHeadUnfoldFn(Function):
    """O[z, m, hh*dk+c] = sum_n py[z, m, hh*d+n] wv_z[hh*dk+c, n] (+ bv_z[hh*dk+c]):   py [2M, h*d] -> [2M, d]."""

    @staticmethod
    def forward(ctx, py, wv0, wv1, bv0, bv1, h):
        py = _flat2(py)
        d = wv0.shape[1]
        M, dk = py.shape[0] // 2, d // h
        wzs, ld = zstride(wv0, wv1), wv0.stride(0)
        out = torch.empty((2 * M, d), device=py.device, dtype=py.dtype)
        ops.gemm(py, wv0, out, M=M, N=dk, K=d, a_rs=py.stride(0), b_rs=ld, ldc=d, bias=bv0, batch=(2, h), a_bs=(M * py.stride(0), d),
                 b_bs=(wzs, dk * ld), c_bs=(M * d, dk), bias_bs2=dk if bv0 is not None else 0,
                 bias_bs1=zstride(bv0, bv1) if bv0 is not None else 0)
        ctx.save_for_backward(py, wv0, wv1)
        ctx.cfg = (h, bv0.dtype if bv0 is not None else None, M, d)
        ctx.w_dst = (getattr(wv0, "_grad_view", None), getattr(wv1, "_grad_view", None))
        ctx.b_dst = (getattr(bv0, "_acc32", None), getattr(bv1, "_acc32", None)) if bv0 is not None else (None, None)
        return out

    @staticmethod
    def backward(ctx, do):
        py, wv0, wv1 = ctx.saved_tensors
        h, bdt, M, d = ctx.cfg
        dk = d // h
        do = do.reshape(2 * M, d)
        if not do.is_contiguous():
            do = do.contiguous()
        wzs, ld = zstride(wv0, wv1), wv0.stride(0)
        dpy = torch.empty((2 * M, h * d), device=py.device, dtype=py.dtype)
        g_dpy = ops.gemm_desc(do, wv0, dpy, M=M, N=d, K=dk, a_rs=d, a_ks=1, b_rs=1, b_ks=ld, ldc=h * d, batch=(2, h), a_bs=(M * d, dk),
                              b_bs=(wzs, dk * ld), c_bs=(M * h * d, d))
        kw = dict(M=dk, N=d, K=M, a_rs=1, a_ks=d, b_rs=1, b_ks=h * d, batch=(2, h), a_bs=(M * d, dk), b_bs=(M * h * d, d))
        dws = None
        if ctx.w_dst[0] is not None and ctx.w_dst[1] is not None:
            gv = ctx.w_dst[0]
            gzs = zstride(gv, ctx.w_dst[1])
            g_dw = ops.gemm_desc(do, py, gv, ldc=gv.stride(0), c_bs=(gzs, dk * gv.stride(0)), residual=gv, ldr=gv.stride(0),
                                 r_bs=(gzs, dk * gv.stride(0)), **kw)
        else:
            dws = torch.empty((2, d, d), device=py.device, dtype=wv0.dtype)
            g_dw = ops.gemm_desc(do, py, dws, ldc=d, c_bs=(d * d, dk * d), **kw)
        ops.gemm_pair(g_dpy, g_dw)
        db = [None, None]
        if bdt is not None:
            for z in range(2):
                acc = ctx.b_dst[z] if ctx.b_dst[z] is not None else ag._f32_zeros((d,), do)
                check(lib.bist_col_sum_acc(do[z * M:(z + 1) * M].data_ptr(), acc.data_ptr(), M, d, d, dtype_code(do.dtype), _stream()), "bist_col_sum_acc")
                db[z] = None if ctx.b_dst[z] is not None else ag._to_dtype_from_f32(acc, bdt)
        return dpy, (dws[0] if dws is not None else None), (dws[1] if dws is not None else None), db[0], db[1], None


def head_unfold(py: Tensor, wv0: Tensor, wv1: Tensor, bv0: Optional[Tensor], bv1: Optional[Tensor], h: int) -> Tensor:
    return ZHeadUnfoldFn.apply(py, wv0, wv1, bv0, bv1, h)


class ZScaledBiasFn(Function):
    """x[z] + s[z, m, head] * bias_z   (the value bias of stage 2 when the probabilities were dropped), both directions."""

    @staticmethod
    def forward(ctx, x, s, b0, b1, h):
        ctx.save_for_backward(s, b0, b1)
        ctx.h = h
        ctx.b_dst = (getattr(b0, "_acc32", None), getattr(b1, "_acc32", None))
        return ops.scaled_bias_z(x, s, b0, zstride(b0, b1), h, 2)

    @staticmethod
    def backward(ctx, dy):
        s, b0, b1 = ctx.saved_tensors
        h = ctx.h
        d = dy.shape[-1]
        dy2 = dy.reshape(-1, d)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        ds = torch.empty(s.shape, device=s.device, dtype=torch.float32)
        direct = ctx.b_dst[0] is not None and ctx.b_dst[1] is not None
        if direct:
            acc0, azs = ctx.b_dst[0], zstride(ctx.b_dst[0], ctx.b_dst[1])
        else:
            acc = ag._f32_zeros((2, d), dy2)
            acc0, azs = acc, d
        check(lib.bist_scaled_bias_bwd_z(dy2.data_ptr(), s.data_ptr(), b0.data_ptr(), ds.data_ptr(), acc0.data_ptr(), dy2.shape[0], h, d // h,
                                         2, zstride(b0, b1), azs, dtype_code(dy2.dtype), _stream()), "bist_scaled_bias_bwd_z")
        if direct:
            return dy, ds, None, None, None
        return dy, ds, ag._to_dtype_from_f32(acc[0], b0.dtype), ag._to_dtype_from_f32(acc[1], b1.dtype), None


def scaled_bias(x: Tensor, s: Tensor, b0: Tensor, b1: Tensor, h: int) -> Tensor:
    return ZScaledBiasFn.apply(x, s, b0, b1, h)


# ----------------------------------------------------------------------------------------------
# stacking / un-stacking at the boundary of the lock-step region
# ----------------------------------------------------------------------------------------------
class ZStackFn(Function):
    """[a; b] -> [2, ...] (two device-side copies; the gradient's halves go back as views)."""

    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty((2,) + tuple(a.shape), device=a.device, dtype=a.dtype)
        ops.copy_into(out[0], a.contiguous())
        ops.copy_into(out[1], b.contiguous())
        return out

    @staticmethod
    def backward(ctx, g):
        return g[0], g[1]


def stack2(a: Tensor, b: Tensor) -> Tensor:
    return ZStackFn.apply(a, b)


class ZUnstackFn(Function):
    """n0 aliases of x[0] and n1 aliases of x[1] for n0 + n1 consumers; backward writes the sums of the aliases' gradients into
    the halves of ONE stacked gradient (bist_add_n per half) -- no pairwise accumulation, no concatenation by autograd."""

    @staticmethod
    def forward(ctx, x, n0, n1):
        ctx.n = (n0, n1)
        ctx.meta = (tuple(x.shape), x.dtype, x.device)
        return tuple(x[0].view_as(x[0]) for _ in range(n0)) + tuple(x[1].view_as(x[1]) for _ in range(n1))

    @staticmethod
    def backward(ctx, *grads):
        n0, n1 = ctx.n
        shape, dtype, dev = ctx.meta
        out = torch.empty(shape, device=dev, dtype=dtype)
        for z, gs in ((0, grads[:n0]), (1, grads[n0:])):
            gs = [g if g.dtype == dtype else g.to(dtype) for g in gs if g is not None]
            if not gs:
                out[z].zero_()
            else:
                ops.add_n(gs, out=out[z])
        return out, None, None


def unstack(x: Tensor, n0: int = 1, n1: int = 1):
    """-> (list of n0 aliases of x[0], list of n1 aliases of x[1])"""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return [x[0]] * n0, [x[1]] * n1
    outs = ZUnstackFn.apply(x, n0, n1)
    return list(outs[:n0]), list(outs[n0:])


# ----------------------------------------------------------------------------------------------
# stage 1 of both directions (A1 / A4)
# ----------------------------------------------------------------------------------------------
class ZStage1Fn(Function):
    """(y_t2s [B,S,Lq,d], y_s2t [B,T,Lq,d], x') from the folded queries qf [2, B, Lq*h, d], the sublayer input x [2, B, Lq, d] (the
    residual of the EXPANDED query, encoder.py:121,148; x' is x again for the next consumer), the video rows (one alias per
    direction; vft0 region-major [B,S,T,d] when `permuted`), the projected values v0 / v1 and the output projections.
    Training form: per direction score product -> softmax + P.V core -> output projection with the row-mapped residual; the
    backward walks the same kernels and writes the two directions' query-side gradients into the halves of stacked tensors."""

    @staticmethod
    def forward(ctx, qf, x, vft0, vft1, v0, v1, tmask, wo0, bo0, wo1, bo1, cfg):
        B, T, S, Lq, h, dk, permuted, adrop, sdrop = cfg
        d = h * dk
        qf3 = qf.reshape(2, B, Lq * h, d)
        x2 = x.reshape(2, B * Lq, d)
        ys, saved = [], []
        # Tensors that outlive this call (saved for backward / returned) are allocated HERE, on the calling stream's pool, before the
        # fork: a block handed out under the side stream would return to the side stream's pool when freed and could be re-used by a
        # later side-stream launch while a consumer on another stream still reads it.
        geo = []
        for z in range(2):
            direction = 1 if (z == 1 or permuted) else 0
            Tc, Sc = (S, T) if (z == 0 and permuted) else (T, S)
            G = Sc if direction == 0 else Tc
            geo.append((direction, Tc, Sc, G,
                        torch.empty((B, Lq * h, Tc * Sc), device=qf.device, dtype=torch.float32),
                        torch.empty((B, G, Lq, d), device=qf.device, dtype=qf.dtype),
                        torch.empty((B * G * Lq, d), device=qf.device, dtype=qf.dtype)))
        kmask8 = ag._mask_u8(tmask.reshape(B, -1)) if tmask is not None else None
        ds = _DirStreams(qf.is_cuda)
        for z, (vft, v, wo, bo) in enumerate(((vft0, v0, wo0, bo0), (vft1, v1, wo1, bo1))):
          with ds.on(z):
            # geometry of this direction as the core sees it: permuted t2s = "s2t" over the region-major tensors + the frame mask
            direction, Tc, Sc, G, scores, o, y = geo[z]
            m8 = kmask8 if z == 0 else None
            q3 = qf3[z]
            TS = Tc * Sc
            vf = vft.reshape(B, TS, d)
            ops.gemm(q3, vf, scores, M=Lq * h, N=TS, K=d, a_rs=q3.stride(1), b_rs=vf.stride(1), ldc=TS, batch=(B, 1),
                     a_bs=(q3.stride(0), 0), b_bs=(vf.stride(0), 0), c_bs=(Lq * h * TS, 0))
            ops.st_stage1_pv(scores, v, m8, B=B, T=Tc, S=Sc, Lq=Lq, h=h, dk=dk, direction=direction, drop=adrop[z], out=o)
            sp, ss = sdrop[z] if sdrop[z] is not None else (0.0, 0)
            ops.linear(o.view(B * G * Lq, d), wo, bo, residual=x2[z], res_map=(G * Lq, Lq), drop_p=sp, drop_seed=ss, out=y)
            ys.append(y.view(B, G, Lq, d))
            saved += [scores, o, m8]
        ds.join()
        ctx.save_for_backward(qf3, vft0, vft1, v0, v1, wo0, wo1, *saved)
        ctx.cfg = cfg
        ctx.bias = (bo0.dtype, tuple(x.shape))
        ctx.w_dst = (getattr(wo0, "_grad_view", None), getattr(wo1, "_grad_view", None))
        ctx.b_dst = (getattr(bo0, "_acc32", None), getattr(bo1, "_acc32", None))
        ctx.set_materialize_grads(False)
        return ys[0], ys[1], x

    @staticmethod
    def backward(ctx, dy0, dy1, dxp):
        qf3, vft0, vft1, v0, v1, wo0, wo1, sc0, o0, m80, sc1, o1, m81 = ctx.saved_tensors
        B, T, S, Lq, h, dk, permuted, adrop, sdrop = ctx.cfg
        bdt, x_shape = ctx.bias
        d = h * dk
        dev, dt = qf3.device, qf3.dtype
        dqf = torch.empty(qf3.shape, device=dev, dtype=dt)
        dres = torch.empty((2, B * Lq, d), device=dev, dtype=dt)
        outs = {}
        dxp2 = None
        if dxp is not None:          # the gradient that came back through x' (the sublayer input's next consumer)
            dxp2 = dxp.reshape(2, B * Lq, d)
            if not dxp2.is_contiguous():
                dxp2 = dxp2.contiguous()
        pre = []                     # escaping tensors: allocated before the fork (see forward)
        for z, (vft, v) in enumerate(((vft0, v0), (vft1, v1))):
            Tc, Sc = (S, T) if (z == 0 and permuted) else (T, S)
            pre.append((torch.empty((B, Tc, Sc, d), device=dev, dtype=v.dtype), torch.empty((B, Tc * Sc, d), device=dev, dtype=vft.dtype)))
        direct = all(w is not None for w in ctx.w_dst) and all(b_ is not None for b_ in ctx.b_dst)
        ds = _DirStreams(dev.type == "cuda" and direct)     # (fresh weight-gradient tensors would be allocated under the side stream)
        for z, (dy, vft, v, wo, scores, o, m8) in enumerate(((dy0, vft0, v0, wo0, sc0, o0, m80), (dy1, vft1, v1, wo1, sc1, o1, m81))):
          with ds.on(z):
              direction = 1 if (z == 1 or permuted) else 0
              Tc, Sc = (S, T) if (z == 0 and permuted) else (T, S)
              G = Sc if direction == 0 else Tc
              TS = Tc * Sc
              M = B * G * Lq
              dv, dvft = pre[z]
              if dy is None:
                  dy = torch.zeros((M, d), device=dev, dtype=dt)
              dy = dy.reshape(M, d)
              if not dy.is_contiguous():
                  dy = dy.contiguous()
              # gradient of the un-expanded query: the sum over the groups (+ the gradient that came back through x')
              check(lib.bist_group_sum_add(dy.data_ptr(), dxp2[z].data_ptr() if dxp2 is not None else None, dres[z].data_ptr(), B, G, Lq * d,
                                           dtype_code(dt), _stream()), "bist_group_sum_add")
              dzz = dy
              if sdrop[z] is not None and sdrop[z][0] > 0:
                  dzz = torch.empty_like(dy)
                  check(lib.bist_epilogue_bwd(dy.data_ptr(), dy.data_ptr(), dzz.data_ptr(), M, d, d, d, d, ACT_NONE, sdrop[z][0], sdrop[z][1],
                                              _ptr(ops.DROP_CTR), dtype_code(dt), _stream()), "bist_epilogue_bwd")
              o2 = o.view(M, d)
              do, dwo, dbo = ag._linear_grads(o2, wo, dzz, 1.0, ctx.w_dst[z], ctx.b_dst[z], bdt, True, ctx.needs_input_grad[7 + 2 * z],
                                              ctx.needs_input_grad[8 + 2 * z])
              direct16 = v.dtype == torch.bfloat16
              dsc = torch.empty(scores.shape, device=dev, dtype=torch.bfloat16 if direct16 else scores.dtype)
              check(lib.bist_st_stage1_pv_bwd(scores.data_ptr(), v.data_ptr(), _ptr(m8), do.data_ptr(), dsc.data_ptr(), dtype_code(dsc.dtype),
                                              dv.data_ptr(), B, Tc, Sc, Lq, h, dk, v.stride(-2), d, direction, ops.drop_ref(adrop[z]),
                                              dtype_code(v.dtype), _stream()), "bist_st_stage1_pv_bwd")
              g = dsc if dsc.dtype == dt else ops.cast(dsc, dt)
              vf = vft.reshape(B, TS, d)
              q3 = qf3[z]
              R = Lq * h
              ops.gemm(g, vf, dqf[z], M=R, N=d, K=TS, a_rs=TS, a_ks=1, b_rs=1, b_ks=vf.stride(1), ldc=d, batch=(B, 1),
                       a_bs=(R * TS, 0), b_bs=(vf.stride(0), 0), c_bs=(R * d, 0))
              ops.gemm(g, q3, dvft, M=TS, N=d, K=R, a_rs=1, a_ks=TS, b_rs=1, b_ks=q3.stride(1), ldc=d, batch=(B, 1),
                       a_bs=(R * TS, 0), b_bs=(q3.stride(0), 0), c_bs=(TS * d, 0))
              outs[z] = (dvft.view(vft.shape), dv, dwo, dbo)
        ds.join()
        return (dqf.view(2, B, Lq * h, d), dres.view(x_shape), outs[0][0], outs[1][0], outs[0][1], outs[1][1], None,
                outs[0][2], outs[0][3], outs[1][2], outs[1][3], None)


# ----------------------------------------------------------------------------------------------
# stage 2 of both directions (A2 / A5)
# ----------------------------------------------------------------------------------------------
class ZStage2Fn(Function):
    """(PY [2,B,Lq,h,d], rowsum [2,B,Lq,h] f32 or None) from q2f [2,B,Lq,h,d] and the two directions' stage-1 outputs y0 [B,S,Lq,d],
    y1 [B,T,Lq,d]: one st_stage2 launch per direction writing its half of the stacked outputs."""

    @staticmethod
    def forward(ctx, q2f, y0, y1, gmask1, h, drops):
        B, Lq, d = y0.shape[0], y0.shape[2], y0.shape[3]
        q5 = q2f.reshape(2, B, Lq, h, d)
        if not q5.is_contiguous():
            q5 = q5.contiguous()
        m8 = ag._mask_u8(gmask1.reshape(B, y1.shape[1])) if gmask1 is not None else None
        py = torch.empty((2, B, Lq, h, d), device=y0.device, dtype=y0.dtype)
        with_rs = any(dr is not None and dr[0] > 0 for dr in drops)
        rs = torch.empty((2, B, Lq, h), device=y0.device, dtype=torch.float32) if with_rs else None
        ds = _DirStreams(y0.is_cuda)
        for z, (y, mk) in enumerate(((y0, None), (y1, m8))):
          with ds.on(z):
            yc = y if y.is_contiguous() else y.contiguous()
            check(lib.bist_st_stage2_fwd(q5[z].data_ptr(), yc.data_ptr(), _ptr(mk), py[z].data_ptr(), rs[z].data_ptr() if rs is not None else None,
                                         B, y.shape[1], Lq, h, d, ops.drop_ref(drops[z]), dtype_code(y.dtype), _stream()), "bist_st_stage2_fwd")
        ds.join()
        ctx.save_for_backward(q5, y0, y1, m8)
        ctx.cfg = (h, drops, tuple(q2f.shape))
        ctx.set_materialize_grads(False)
        if rs is None:
            ctx.mark_non_differentiable()
            return py, None
        return py, rs

    @staticmethod
    def backward(ctx, dpy, drs):
        q5, y0, y1, m8 = ctx.saved_tensors
        h, drops, q_shape = ctx.cfg
        B, Lq, d = y0.shape[0], y0.shape[2], y0.shape[3]
        if dpy is None:
            dpy = torch.zeros(q5.shape, device=q5.device, dtype=q5.dtype)
        dpy = dpy.reshape(q5.shape)
        if not dpy.is_contiguous():
            dpy = dpy.contiguous()
        if drs is not None:
            drs = drs.reshape(2, B, Lq, h).contiguous().float()
        dq = torch.empty_like(q5)
        dys = []
        ycs = [y if y.is_contiguous() else y.contiguous() for y in (y0, y1)]
        pre = [torch.empty_like(yc) for yc in ycs]           # escaping tensors: allocated before the fork (see ZStage1Fn.forward)
        ds = _DirStreams(y0.is_cuda)
        for z, (y, mk) in enumerate(((y0, None), (y1, m8))):
          with ds.on(z):
            yc, dy = ycs[z], pre[z]
            check(lib.bist_st_stage2_bwd(q5[z].data_ptr(), yc.data_ptr(), _ptr(mk), dpy[z].data_ptr(), drs[z].data_ptr() if drs is not None else None,
                                         dq[z].data_ptr(), dy.data_ptr(), B, y.shape[1], Lq, h, d, ops.drop_ref(drops[z]), dtype_code(y.dtype),
                                         _stream()), "bist_st_stage2_bwd")
            dys.append(dy)
        ds.join()
        return dq.view(q_shape), dys[0], dys[1], None, None, None
