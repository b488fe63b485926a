"""torch.autograd.Function wrappers: forward AND backward of every step run on libbist_hip.so.

autograd itself is plumbing here (it records the graph and routes gradient tensors); no gradient is
computed by an aten kernel except trivial view/concat bookkeeping.  Each backward recomputes what it
needs from the saved inputs (probabilities are never stored) and calls the kernels declared in
include/bist_hip.h under "Backward kernels".
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence, Tuple

import torch
from torch.autograd import Function

from . import ops
from ._lib import ACT_GATE, ACT_NONE, ACT_RELU, check, lib
from .ops import _ptr, _stream, dtype_code

Tensor = torch.Tensor


def _f32_zeros(shape, like: Tensor) -> Tensor:
    return torch.zeros(shape, device=like.device, dtype=torch.float32)


def _to_dtype_from_f32(acc: Tensor, dtype: torch.dtype) -> Tensor:
    if dtype == torch.float32:
        return acc
    out = torch.empty(acc.shape, device=acc.device, dtype=dtype)
    check(lib.bist_cast_from_f32(acc.data_ptr(), out.data_ptr(), acc.numel(), dtype_code(dtype), _stream()), "bist_cast_from_f32")
    return out


class _WeightGradStream:
    """Weight-gradient GEMMs accumulate into the trainer's flat gradient and nothing downstream of the backward pass
    reads them, so they leave the critical path: on ``ops.WGRAD_STREAM`` (one stream for ALL of them, which also
    serialises accumulations into a shared weight) after a fork from the current stream.  The operands are kept alive
    in ``ops.WGRAD_KEEP`` until the trainer joins the stream."""

    def __init__(self, *keep):
        self.keep = keep
        self.cm = None

    def __enter__(self):
        st = ops.WGRAD_STREAM
        if st is not None:
            st.wait_stream(torch.cuda.current_stream())
            ops.WGRAD_KEEP.append(self.keep)
            self.cm = torch.cuda.stream(st)
            self.cm.__enter__()
        return self

    def __exit__(self, *exc):
        if self.cm is not None:
            self.cm.__exit__(*exc)
        return False


class _Leaf:
    """Products of a backward node that nothing on its chain of dependent launches reads -- weight gradients (they accumulate into the
    trainer's flat gradient) and the video tensor's gradients of the frame-grid products (summed once, at the end, by FanOutFn) -- leave
    the chain: on ``ops.LEAF_STREAM`` (set by the trainer around a backward pass whose step is replayed by the split executor; a chain
    of its own there) after a fork from the current stream.  Operands are kept alive in ``ops.WGRAD_KEEP`` until the trainer has joined
    the streams; a tensor that escapes to another autograd node takes ``ready()`` -- the event its consumer waits on -- as its
    ``_bist_ready`` attribute (FanOutFn honours it).  Without a leaf stream the body runs in line."""

    def __init__(self, *keep, on: bool = True):
        self.keep = keep
        self.cm = None
        self.st = None
        self.on = on

    def __enter__(self):
        st = ops.LEAF_STREAM if self.on else None
        if st is not None and st != torch.cuda.current_stream():
            st.wait_stream(torch.cuda.current_stream())
            ops.WGRAD_KEEP.append(self.keep)
            self.st = st
            self.cm = torch.cuda.stream(st)
            self.cm.__enter__()
        return self

    def ready(self):
        if self.st is None:
            return None
        ev = torch.cuda.Event()
        ev.record(self.st)
        return ev

    def __exit__(self, *exc):
        if self.cm is not None:
            self.cm.__exit__(*exc)
        return False


def _mask_u8(mask: Optional[Tensor]):
    if mask is None:
        return None
    m = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
    return m if m.stride(-1) == 1 else m.contiguous()


# ----------------------------------------------------------------------------------------------
# linear
# ----------------------------------------------------------------------------------------------
class GateTag:
    """Travels with y = drop(relu(z)) of a producing linear layer (attribute ``_bist_gate`` of y, ``gate_tag`` of its autograd node).
    A consuming linear layer that sees it computes its dX already gated by y > 0 and scaled by 1/(1-p) -- the gradient of z -- and
    records that here, so that the producer's backward can tell "the hand-off arrived" from "it was lost on the way" (a second
    consumer, a hook or a view re-wrapping the gradient): the latter would silently mask and scale twice, and raises instead."""
    __slots__ = ("p", "seed", "gated")

    def __init__(self, p, seed):
        self.p, self.seed, self.gated = float(p), int(seed), 0

    def __getitem__(self, i):                      # (p, seed) view for the consumers
        return (self.p, self.seed)[i]


class LinearFn(Function):
    """y = drop(act(alpha * x.W^T + bias)) + residual[row map]   (GEMM epilogue, include/bist_hip.h)."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, act, res_map, alpha, drop_p, drop_seed, out_dtype, out_shape=None, leaf=False):
        ctx.leaf = bool(leaf)
        K = x.shape[-1]
        ln = getattr(x, "_bist_ln", None)              # x is a pending LayerNorm output: this product normalises the rows itself and fills x
        if ln is not None:
            x._bist_ln = None
        x2 = x.reshape(-1, K)
        y = ops.linear(x2, w, bias, act=act, residual=residual, res_map=res_map, alpha=alpha, out_dtype=out_dtype,
                       drop_p=drop_p, drop_seed=drop_seed, ln=ln)
        ctx.save_for_backward(x2, w, y if act == ACT_RELU else None)
        ctx.cfg = (act, res_map, alpha, drop_p, drop_seed, bias is not None, bias.dtype if bias is not None else None,
                   residual is not None, tuple(residual.shape) if residual is not None else None, tuple(x.shape))
        # trainer-owned gradient destinations (bist_amd/train.py): weight gradients are accumulated by the
        # GEMM straight into the flat gradient buffer, bias gradients into the fp32 accumulator
        ctx.w_dst = getattr(w, "_grad_view", None)
        ctx.b_dst = getattr(bias, "_acc32", None) if bias is not None else None
        gate = getattr(x, "_bist_gate", None)             # x = drop(relu(.)) of the producing linear layer, 2-D and saved here as x2
        ctx.gate = gate if (gate is not None and x.dim() == 2 and x.is_contiguous() and x.dtype in (torch.bfloat16, torch.float32)) else None
        # the caller's shape is produced HERE (not by a view afterwards): a view node between this Function and its consumer
        # would re-wrap the gradient tensor and drop the consumer's _bist_dz hand-off (see backward)
        return y if out_shape is None else y.view(out_shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        act, res_map, alpha, drop_p, drop_seed, has_bias, bias_dtype, has_res, res_shape, x_shape = ctx.cfg
        M, K = x2.shape
        N = w.shape[0]
        pre = getattr(dy, "_bist_dz", None)            # (masked gradient, p, seed) left by _ln_backward on this very tensor
        if pre is None and getattr(ctx, "_bist_expect_dz", False):
            raise RuntimeError("bist_amd: the fused loss hands this projection its gradient as an attribute of an UNINITIALISED placeholder "
                               "tensor, and the attribute did not arrive (a hook, retain_grad or a view on the logits re-wrapped it): using "
                               "the placeholder would feed garbage into the shared embedding's gradient -- set BIST_AE_GROUPED=0 for such graphs")
        dy = dy.reshape(M, N)
        if not dy.is_contiguous():
            dy = dy.contiguous()
        dres = None
        if has_res and ctx.needs_input_grad[3]:
            if res_map != (0, 0):
                outer, inner = res_map
                G = outer // inner
                dres = torch.empty((M // G, N), device=dy.device, dtype=dy.dtype)
                check(lib.bist_group_sum(dy.data_ptr(), dres.data_ptr(), M // outer, G, inner * N, dtype_code(dy.dtype), _stream()),
                      "bist_group_sum")
                dres = dres.view(res_shape) if dres.numel() == math.prod(res_shape) else dres
            else:
                dres = dy.view(res_shape)
        tag = getattr(ctx, "gate_tag", None)
        handed = (pre is not None and len(pre) == 4 and act == ACT_RELU and pre[1:3] == (float(drop_p), int(drop_seed))
                  and pre[0].numel() == M * N and pre[0].dtype == x2.dtype)
        masked = (pre is not None and len(pre) == 3 and act == ACT_NONE and pre[1:] == (float(drop_p), int(drop_seed)) and pre[0].numel() == M * N
                  and pre[0].dtype == x2.dtype)
        # (a handed-over gradient replaces dy altogether: dy may then be an uninitialised placeholder of another dtype -- no cast of it)
        dz = dy if (dy.dtype == x2.dtype or handed or masked) else ops.cast(dy, x2.dtype)
        if tag is not None and tag.gated and not handed:
            raise RuntimeError("bist_amd: a consuming linear layer gated its input gradient with this layer's drop(relu(.)) output, but "
                               "the gated tensor did not reach this backward (second consumer, hook or view on the hidden activation): "
                               "masking it again would be wrong -- set BIST_GATE_HANDOFF=0 for such graphs")
        if handed:
            dz = pre[0].view(M, N)                     # already gated (y > 0, 1/(1-p)) by the consumer's dX product
        elif masked:
            dz = pre[0].view(M, N)                     # already masked by the LayerNorm backward that produced dy (or handed over in the operand dtype)
        elif act == ACT_RELU or drop_p > 0:
            dz2 = torch.empty_like(dz)
            yy = y if y is not None else dz
            if y is not None and y.dtype != dz.dtype:
                raise RuntimeError("relu epilogue backward needs the output in the operand dtype")
            check(lib.bist_epilogue_bwd(dz.data_ptr(), yy.data_ptr(), dz2.data_ptr(), M, N, N, N, N, act, drop_p, drop_seed,
                                        _ptr(ops.DROP_CTR) if drop_p > 0 else None, dtype_code(dz.dtype), _stream()),
                  "bist_epilogue_bwd")
            dz = dz2
        if ctx.leaf and ctx.gate is None and ops.LEAF_STREAM is not None and (ops.LEAF_MASK & 1):
            # a projection of the video tensor (the stage-1 values): both backward products are leaves -- dX is one of the video tensor's
            # gradients (FanOutFn sums them at the end), dW accumulates into the flat gradient
            with _Leaf(x2, w, dz) as lf:
                dx, dw, db = _linear_grads(x2, w, dz, alpha, ctx.w_dst, ctx.b_dst, bias_dtype if has_bias else None,
                                           ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2])
                ev = lf.ready()
            if dx is not None:
                dx = dx.view(x_shape)
                if ev is not None:
                    dx._bist_ready = ev
                    ops.WGRAD_KEEP.append((dx,))
            return dx, dw, db, dres, None, None, None, None, None, None, None, None
        dx, dw, db = _linear_grads(x2, w, dz, alpha, ctx.w_dst, ctx.b_dst, bias_dtype if has_bias else None,
                                   ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2], gate=ctx.gate)
        if dx is not None:
            dx = dx.view(x_shape)
            if ctx.gate is not None:
                dx._bist_dz = (dx, ctx.gate[0], ctx.gate[1], "gate")      # survives only if autograd hands THIS tensor to the producer
                if isinstance(ctx.gate, GateTag):
                    ctx.gate.gated += 1
        return dx, dw, db, dres, None, None, None, None, None, None, None, None


def _linear_grads(x2, w, dz, alpha, w_dst, b_dst, bias_dtype, need_dx, need_dw, need_db, gate=None):
    """dX = alpha dZ.W, dW = alpha dZ^T.X (accumulated into the trainer's gradient view when there is one), db = colsum(dZ)
    of y = alpha x.W^T + b -- the two products as ONE launch when both are small (bist_gemm_pair)."""
    M, K = x2.shape
    N = w.shape[0]
    dx = dw = db = None
    g_dx = g_dw = None
    if need_dx:
        dx = torch.empty((M, K), device=dz.device, dtype=x2.dtype)
        if gate is not None:      # x2 = drop(relu(z)) of the producing layer: dX gated by x2 > 0 and scaled by 1/(1-p) = the gradient of z
            g_dx = ops.gemm_desc(dz, w, dx, M=M, N=K, K=N, a_rs=N, a_ks=1, b_rs=1, b_ks=w.stride(0), ldc=K,
                                 alpha=alpha / (1.0 - gate[0]), act=ACT_GATE, residual=x2, ldr=x2.stride(0))
        else:
            g_dx = ops.gemm_desc(dz, w, dx, M=M, N=K, K=N, a_rs=N, a_ks=1, b_rs=1, b_ks=w.stride(0), ldc=K, alpha=alpha)
    if need_dw:
        if w_dst is not None:                # dW accumulates in place: C = alpha * dz^T x + C
            g_dw = ops.gemm_desc(dz, x2, w_dst, M=N, N=K, K=M, a_rs=1, a_ks=N, b_rs=1, b_ks=x2.stride(0), ldc=w_dst.stride(0), alpha=alpha,
                                 residual=w_dst, ldr=w_dst.stride(0))
        else:
            dw = torch.empty((N, K), device=dz.device, dtype=w.dtype)
            g_dw = ops.gemm_desc(dz, x2, dw, M=N, N=K, K=M, a_rs=1, a_ks=N, b_rs=1, b_ks=x2.stride(0), ldc=K, alpha=alpha)
    if g_dx is not None and g_dw is not None and ops.WGRAD_STREAM is None:
        ops.gemm_pair(g_dx, g_dw)
    else:
        if g_dx is not None:
            check(lib.bist_gemm(C.byref(g_dx), _stream()), "bist_gemm")
        if g_dw is not None:
            with _WeightGradStream(dz, x2):
                check(lib.bist_gemm(C.byref(g_dw), _stream()), "bist_gemm")
    if need_db:
        acc = b_dst if b_dst is not None else _f32_zeros((N,), dz)
        if b_dst is not None and ops.COLSUM_QUEUE is not None:
            ops.COLSUM_QUEUE.append((dz, acc, M, N))     # the trainer sums all bias gradients in a few batched launches
        else:
            check(lib.bist_col_sum_acc(dz.data_ptr(), acc.data_ptr(), M, N, N, dtype_code(dz.dtype), _stream()), "bist_col_sum_acc")
        db = None if b_dst is not None else _to_dtype_from_f32(acc, bias_dtype)
    return dx, dw, db


class LinearPairFn(Function):
    """(x1.W1^T + b1, x2.W2^T + b2): two independent plain projections (the query and the packed key/value projection of a
    cross-attention, modules.py:89-91) in one launch forward, and one launch per projection backward."""

    @staticmethod
    def forward(ctx, x1, w1, b1, x2, w2, b2):
        ops.ensure_ln(x1); ops.ensure_ln(x2)          # (the paired launch has no LayerNorm prologue)
        xs = (x1.reshape(-1, x1.shape[-1]), x2.reshape(-1, x2.shape[-1]))
        ys, descs = [], []
        for x, w, b in ((xs[0], w1, b1), (xs[1], w2, b2)):
            y = torch.empty((x.shape[0], w.shape[0]), device=x.device, dtype=x.dtype)
            descs.append(ops.gemm_desc(x, w, y, M=x.shape[0], N=w.shape[0], K=x.shape[1], a_rs=x.stride(0), b_rs=w.stride(0), ldc=y.stride(0), bias=b))
            ys.append(y)
        ops.gemm_pair(descs[0], descs[1])
        ctx.save_for_backward(xs[0], w1, xs[1], w2)
        ctx.cfg = [(getattr(w, "_grad_view", None), getattr(b, "_acc32", None) if b is not None else None,
                    b.dtype if b is not None else None, tuple(x.shape)) for w, b, x in ((w1, b1, x1), (w2, b2, x2))]
        return ys[0], ys[1]

    @staticmethod
    def backward(ctx, dy1, dy2):
        x1, w1, x2, w2 = ctx.saved_tensors
        out = []
        for i, (x, w, dy) in enumerate(((x1, w1, dy1), (x2, w2, dy2))):
            w_dst, b_dst, bdt, x_shape = ctx.cfg[i]
            if dy is None:
                out += [None, None, None]
                continue
            dz = dy.reshape(x.shape[0], w.shape[0])
            if not dz.is_contiguous():
                dz = dz.contiguous()
            dx, dw, db = _linear_grads(x, w, dz, 1.0, w_dst, b_dst, bdt, ctx.needs_input_grad[3 * i], ctx.needs_input_grad[3 * i + 1],
                                       bdt is not None and ctx.needs_input_grad[3 * i + 2])
            out += [dx.view(x_shape) if dx is not None else None, dw, db]
        return tuple(out)


GATE_HANDOFF = os.environ.get("BIST_GATE_HANDOFF", "1") != "0"      # tuning aid


def linear(x, w, bias=None, *, act=ACT_NONE, residual=None, res_map=(0, 0), alpha=1.0, out=None, out_dtype=None,
           accumulate=False, drop_p=0.0, drop_seed=0, out_shape=None, leaf=False):
    """leaf: both backward products of this projection are leaves of the backward pass (see _Leaf): x is the video tensor."""
    if not torch.is_grad_enabled():
        y = ops.linear(x, w, bias, act=act, residual=residual, res_map=res_map, alpha=alpha, out=out, out_dtype=out_dtype,
                       accumulate=accumulate, drop_p=drop_p, drop_seed=drop_seed)
        return y if out_shape is None else y.view(out_shape)
    if accumulate:                       # out-of-place under autograd: the running sum is the residual
        residual = out
    if residual is not None and residual.dim() != 2:
        residual = residual.reshape(-1, residual.shape[-1])
    y = LinearFn.apply(x, w, bias, residual, act, tuple(res_map), alpha, drop_p, drop_seed, out_dtype,
                       tuple(out_shape) if out_shape is not None else None, bool(leaf))
    if drop_p > 0 and act == ACT_NONE and res_map == (0, 0):
        # y = drop(z) + res: a LayerNorm that consumes y can hand the masked gradient of z back (see _ln_backward)
        y._bist_drop = (float(drop_p), int(drop_seed), w.shape[0])
    if act == ACT_RELU and residual is None and out_shape is None and GATE_HANDOFF:
        # y = drop(relu(z)): a linear layer that consumes y gates its dX product with y > 0 and hands the result back as the
        # gradient of z (LinearFn.backward), so no separate masking pass runs between the two backward products
        y._bist_gate = GateTag(drop_p, drop_seed)
        if y.grad_fn is not None:
            y.grad_fn.gate_tag = y._bist_gate          # the node object is the ctx of LinearFn.backward
    return y


# ----------------------------------------------------------------------------------------------
# batched-GEMM steps of the folded attention
# ----------------------------------------------------------------------------------------------
class HeadFoldFn(Function):
    """Qf[m, hh*d+n] = alpha * sum_c q[m, hh*dk+c] wk[hh*dk+c, n]."""

    @staticmethod
    def forward(ctx, q, wk, h, alpha):
        ctx.save_for_backward(q, wk)
        ctx.cfg = (h, alpha)
        ctx.w_dst = getattr(wk, "_grad_view", None)
        M, d = q.shape
        dk = d // h
        out = torch.empty((M, h * d), device=q.device, dtype=q.dtype)
        ops.gemm(q, wk, out, M=M, N=d, K=dk, a_rs=q.stride(0), a_ks=1, b_rs=1, b_ks=wk.stride(0), ldc=h * d,
                 batch=(1, h), a_bs=(0, dk), b_bs=(0, dk * wk.stride(0)), c_bs=(0, d), alpha=alpha)
        return out

    @staticmethod
    def backward(ctx, dqf):
        q, wk = ctx.saved_tensors
        h, alpha = ctx.cfg
        M, d = q.shape
        dk = d // h
        dqf = dqf.contiguous()
        dq = torch.empty((M, d), device=q.device, dtype=q.dtype)
        g_dq = ops.gemm_desc(dqf, wk, dq, M=M, N=dk, K=d, a_rs=h * d, b_rs=wk.stride(0), ldc=d, batch=(1, h), a_bs=(0, d),
                             b_bs=(0, dk * wk.stride(0)), c_bs=(0, dk), alpha=alpha)
        gv = ctx.w_dst
        if gv is not None:
            g_dw = ops.gemm_desc(q, dqf, gv, M=dk, N=d, K=M, a_rs=1, a_ks=q.stride(0), b_rs=1, b_ks=h * d, ldc=gv.stride(0), batch=(1, h),
                                 a_bs=(0, dk), b_bs=(0, d), c_bs=(0, dk * gv.stride(0)), alpha=alpha, residual=gv, ldr=gv.stride(0),
                                 r_bs=(0, dk * gv.stride(0)))
            ops.gemm_pair(g_dq, g_dw)            # (N,N) + (T,T): one launch
            return dq, None, None, None
        check(lib.bist_gemm(C.byref(g_dq), _stream()), "bist_gemm")
        dwk = torch.empty((d, d), device=q.device, dtype=wk.dtype)
        ops.gemm(q, dqf, dwk, M=dk, N=d, K=M, a_rs=1, a_ks=q.stride(0), b_rs=1, b_ks=h * d, ldc=d, batch=(1, h),
                 a_bs=(0, dk), b_bs=(0, d), c_bs=(0, dk * d), alpha=alpha)
        return dq, dwk, None, None


class HeadUnfoldFn(Function):
    """O[m, hh*dk+c] = sum_n py[m, hh*d+n] wv[hh*dk+c, n] + bv[hh*dk+c]."""

    @staticmethod
    def forward(ctx, py, wv, bv, h):
        ctx.save_for_backward(py, wv)
        ctx.cfg = (h, bv.dtype if bv is not None else None)
        ctx.w_dst, ctx.b_dst = getattr(wv, "_grad_view", None), (getattr(bv, "_acc32", None) if bv is not None else None)
        M = py.shape[0]
        d = wv.shape[1]
        dk = d // h
        out = torch.empty((M, d), device=py.device, dtype=py.dtype)
        ops.gemm(py, wv, out, M=M, N=dk, K=d, a_rs=py.stride(0), b_rs=wv.stride(0), ldc=d, bias=bv, batch=(1, h),
                 a_bs=(0, d), b_bs=(0, dk * wv.stride(0)), c_bs=(0, dk), bias_bs2=dk)
        return out

    @staticmethod
    def backward(ctx, do):
        py, wv = ctx.saved_tensors
        h, bdt = ctx.cfg
        M = py.shape[0]
        d = wv.shape[1]
        dk = d // h
        do = do.contiguous()
        dpy = torch.empty((M, h * d), device=py.device, dtype=py.dtype)
        g_dpy = ops.gemm_desc(do, wv, dpy, M=M, N=d, K=dk, a_rs=d, a_ks=1, b_rs=1, b_ks=wv.stride(0), ldc=h * d, batch=(1, h),
                              a_bs=(0, dk), b_bs=(0, dk * wv.stride(0)), c_bs=(0, d))
        gv = ctx.w_dst
        if gv is not None:
            g_dw = ops.gemm_desc(do, py, gv, M=dk, N=d, K=M, a_rs=1, a_ks=d, b_rs=1, b_ks=h * d, ldc=gv.stride(0), batch=(1, h),
                                 a_bs=(0, dk), b_bs=(0, d), c_bs=(0, dk * gv.stride(0)), residual=gv, ldr=gv.stride(0),
                                 r_bs=(0, dk * gv.stride(0)))
            ops.gemm_pair(g_dpy, g_dw)           # (N,T) + (T,T): one launch
            dwv = None
        else:
            check(lib.bist_gemm(C.byref(g_dpy), _stream()), "bist_gemm")
            dwv = torch.empty((d, d), device=py.device, dtype=wv.dtype)
            ops.gemm(do, py, dwv, M=dk, N=d, K=M, a_rs=1, a_ks=d, b_rs=1, b_ks=h * d, ldc=d, batch=(1, h),
                     a_bs=(0, dk), b_bs=(0, d), c_bs=(0, dk * d))
        if bdt is None:                      # no bias here: it is applied by ScaledBiasFn (attention dropout)
            return dpy, dwv, None, None
        acc = ctx.b_dst if ctx.b_dst is not None else _f32_zeros((d,), do)
        check(lib.bist_col_sum_acc(do.data_ptr(), acc.data_ptr(), M, d, d, dtype_code(do.dtype), _stream()), "bist_col_sum_acc")
        return dpy, dwv, (None if ctx.b_dst is not None else _to_dtype_from_f32(acc, bdt)), None


class StScoresFn(Function):
    """scores[b,r,ts] = qf[b,r,:] . vft[b,ts,:]  (f32 out)."""

    @staticmethod
    def forward(ctx, qf, vft):
        ctx.save_for_backward(qf, vft)
        B, R, d = qf.shape
        TS = vft.shape[1]
        out = torch.empty((B, R, TS), device=qf.device, dtype=torch.float32)
        ops.gemm(qf, vft, out, M=R, N=TS, K=d, a_rs=qf.stride(1), b_rs=vft.stride(1), ldc=TS, batch=(B, 1),
                 a_bs=(qf.stride(0), 0), b_bs=(vft.stride(0), 0), c_bs=(R * TS, 0))
        return out

    @staticmethod
    def backward(ctx, dsc):
        qf, vft = ctx.saved_tensors
        B, R, d = qf.shape
        TS = vft.shape[1]
        g = getattr(dsc, "_bist_dsc", None)        # bf16 gradient handed over by StStage1PvFn.backward (dsc is then a placeholder)
        if g is None or g.dtype != qf.dtype:
            g = ops.cast(dsc.contiguous(), qf.dtype)
        dqf = torch.empty((B, R, d), device=qf.device, dtype=qf.dtype)
        ops.gemm(g, vft, dqf, M=R, N=d, K=TS, a_rs=TS, a_ks=1, b_rs=1, b_ks=vft.stride(1), ldc=d, batch=(B, 1),
                 a_bs=(R * TS, 0), b_bs=(vft.stride(0), 0), c_bs=(R * d, 0))
        dv = torch.empty((B, TS, d), device=qf.device, dtype=vft.dtype)
        ops.gemm(g, qf, dv, M=TS, N=d, K=R, a_rs=1, a_ks=TS, b_rs=1, b_ks=qf.stride(1), ldc=d, batch=(B, 1),
                 a_bs=(R * TS, 0), b_bs=(qf.stride(0), 0), c_bs=(TS * d, 0))
        return dqf, dv


class BmmNNFn(Function):
    """C[b] = A[b] . B[b]   A [Z,M,K], B [Z,K,N] (both row-major) -> [Z,M,N]."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        Z, M, K = a.shape
        N = b.shape[2]
        out = torch.empty((Z, M, N), device=a.device, dtype=a.dtype)
        ops.gemm(a, b, out, M=M, N=N, K=K, a_rs=a.stride(1), a_ks=1, b_rs=1, b_ks=b.stride(1), ldc=N, batch=(Z, 1),
                 a_bs=(a.stride(0), 0), b_bs=(b.stride(0), 0), c_bs=(M * N, 0))
        return out

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        Z, M, K = a.shape
        N = b.shape[2]
        dc = dc.contiguous()
        da = torch.empty((Z, M, K), device=a.device, dtype=a.dtype)       # dA = dC . B^T
        ops.gemm(dc, b, da, M=M, N=K, K=N, a_rs=N, b_rs=b.stride(1), ldc=K, batch=(Z, 1), a_bs=(M * N, 0),
                 b_bs=(b.stride(0), 0), c_bs=(M * K, 0))
        db = torch.empty((Z, K, N), device=a.device, dtype=b.dtype)       # dB = A^T . dC
        ops.gemm(a, dc, db, M=K, N=N, K=M, a_rs=1, a_ks=a.stride(1), b_rs=1, b_ks=N, ldc=N, batch=(Z, 1),
                 a_bs=(a.stride(0), 0), b_bs=(M * N, 0), c_bs=(K * N, 0))
        return da, db


# ----------------------------------------------------------------------------------------------
# row-wise steps
# ----------------------------------------------------------------------------------------------
def _ln_backward(ctx, dy, dres):
    x, a = ctx.saved_tensors
    eps, bdt = ctx.cfg
    d = x.shape[-1]
    x2 = x.reshape(-1, d)
    if dy is None:                          # only the residual branch carried a gradient
        dy = torch.zeros_like(x)
    dy2 = dy.reshape(-1, d)
    if dy2.stride(1) != 1:
        dy2 = dy2.contiguous()
    add2 = None
    if dres is not None:
        add2 = dres.reshape(-1, d)
        if add2.stride(1) != 1 or add2.dtype != x.dtype:
            add2 = add2.to(x.dtype).contiguous()
    dx = torch.empty(x2.shape, device=x.device, dtype=x.dtype)
    direct = ctx.a_dst is not None and ctx.b_dst is not None
    da, db = (ctx.a_dst, ctx.b_dst) if direct else (_f32_zeros((d,), x), _f32_zeros((d,), x))
    # trainer: dx only on the critical path; the gain/offset gradients of every LayerNorm are summed in one batched launch
    defer = (direct and ops.LNGRAD_QUEUE is not None and d * x.element_size() == 1024 and a.data_ptr() % 16 == 0
             and all(t.data_ptr() % 16 == 0 and (t.stride(0) * t.element_size()) % 16 == 0 for t in (dy2, x2) + ((add2,) if add2 is not None else ())))
    up = getattr(ctx, "up_drop", None)          # x = drop(z) + res came out of a GEMM with this dropout epilogue
    dz = torch.empty(x2.shape, device=x.device, dtype=x.dtype) if (up is not None and up[2] == d) else None
    zdrop = C.byref(ops.BistDrop(up[0], up[1] & 0xFFFFFFFFFFFFFFFF, _ptr(ops.DROP_CTR))) if dz is not None else None
    check(lib.bist_layernorm_bwd(dy2.data_ptr(), x2.data_ptr(), a.data_ptr(), dx.data_ptr(),
                                 None if defer else da.data_ptr(), None if defer else db.data_ptr(),
                                 x2.shape[0], d, dy2.stride(0), x2.stride(0), d, eps,
                                 add2.data_ptr() if add2 is not None else None, add2.stride(0) if add2 is not None else 0,
                                 _ptr(dz), zdrop, dtype_code(x.dtype), _stream()),
          "bist_layernorm_bwd")
    if defer:
        ops.LNGRAD_QUEUE.append((dy2, x2, a, da, db, eps))
    gx = dx.view(x.shape)
    if dz is not None:
        gx._bist_dz = (dz, up[0], up[1])        # survives only if autograd hands THIS tensor to the producer's backward
    if direct:
        return gx, None, None, None
    return gx, _to_dtype_from_f32(da, a.dtype), _to_dtype_from_f32(db, bdt), None


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, a, b, eps, up_drop=None, lazy=False):
        ctx.save_for_backward(x, a)
        ctx.cfg = (eps, b.dtype)
        ctx.up_drop = up_drop
        ctx.a_dst, ctx.b_dst = getattr(a, "_acc32", None), getattr(b, "_acc32", None)
        return ops.layernorm(x, a, b, eps, lazy=lazy)

    @staticmethod
    def backward(ctx, dy):
        return _ln_backward(ctx, dy, None) + (None, None)


class LayerNormResFn(Function):
    """(LN(x), x): the second output is x itself, handed to the sublayer's residual add.  Its gradient comes
    back HERE, so the LayerNorm backward kernel adds it to dx (x + sublayer(LN(x)), modules.py:44) and
    autograd never launches a separate add for the two uses of x."""

    @staticmethod
    def forward(ctx, x, a, b, eps, up_drop=None, lazy=False):
        ctx.save_for_backward(x, a)
        ctx.cfg = (eps, b.dtype)
        ctx.up_drop = up_drop
        ctx.a_dst, ctx.b_dst = getattr(a, "_acc32", None), getattr(b, "_acc32", None)
        ctx.set_materialize_grads(False)
        return ops.layernorm(x, a, b, eps, lazy=lazy), x

    @staticmethod
    def backward(ctx, dy, dres):
        return _ln_backward(ctx, dy, dres) + (None, None)


class EmbedFn(Function):
    @staticmethod
    def forward(ctx, ids, lut, pe, drop=None):
        ctx.save_for_backward(ids)
        ctx.cfg = (tuple(lut.shape), lut.dtype)
        ctx.drop = drop
        ctx.dst = getattr(lut, "_acc32", None)
        return ops.embed_pe(ids, lut, pe, drop=drop)

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        shape, dt = ctx.cfg
        dy = dy.contiguous()
        acc = ctx.dst if ctx.dst is not None else _f32_zeros(shape, dy)
        check(lib.bist_embed_bwd(ids.contiguous().data_ptr(), dy.data_ptr(), acc.data_ptr(), ids.numel(), shape[1],
                                 ops.drop_ref(ctx.drop), dtype_code(dy.dtype), _stream()), "bist_embed_bwd")
        return None, (None if ctx.dst is not None else _to_dtype_from_f32(acc, dt)), None, None


class FuseFn(Function):
    @staticmethod
    def forward(ctx, score, *xs):
        ctx.save_for_backward(score, *xs)
        return ops.fuse_modalities(score, xs)

    @staticmethod
    def backward(ctx, dout):
        score, *xs = ctx.saved_tensors
        n, d = score.shape[-1], xs[0].shape[-1]
        dout = dout.contiguous()
        score = score.contiguous()
        xs = [x.contiguous() for x in xs]
        rows = dout.numel() // d
        dscore = torch.empty_like(score)
        dxs = [torch.empty_like(x) for x in xs]
        xa = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        da = (C.c_void_p * n)(*[x.data_ptr() for x in dxs])
        check(lib.bist_fuse_modalities_bwd(score.data_ptr(), xa, dout.data_ptr(), dscore.data_ptr(), da, rows, n, d,
                                           dtype_code(score.dtype), _stream()), "bist_fuse_modalities_bwd")
        return (dscore, *dxs)


FUSE_ONE_LAUNCH = os.environ.get("BIST_FUSE_ONE_LAUNCH", "1") != "0"      # tuning aid: 0 = the fusion logits and their backward as one product per part


class FuseDynFn(Function):
    """The dynamic modality fusion of decoder.py:142-159 as ONE autograd node: score = cat(parts) W^T + b through per-part column blocks
    (never the concat), out = sum_j softmax(score)[..., j] xs[j] with xs = parts[xs_idx[j]].  Every modality tensor feeds both the score
    product and the weighted sum; as separate nodes that is two gradients per tensor and an accumulation launch by autograd for each
    (three tensors x six layers).  Here the gradient of the weighted sum rides as the residual operand of the score product's dX launch."""

    @staticmethod
    def forward(ctx, W, bias, xs_idx, *parts):
        d = parts[0].shape[-1]
        p2 = [p.reshape(-1, d) for p in parts]
        p2 = [p if p.is_contiguous() else p.contiguous() for p in p2]
        n_, ns_ = len(p2), W.shape[0]
        one = (FUSE_ONE_LAUNCH and 1 <= n_ <= 4 and ns_ <= 4 and d % 8 == 0 and W.stride(1) == 1 and W.stride(0) % 8 == 0 and W.data_ptr() % 16 == 0
               and W.dtype in (torch.bfloat16, torch.float32) and all(p.dtype == W.dtype and p.data_ptr() % 16 == 0 for p in p2)
               and (bias is None or bias.dtype == W.dtype))
        if one:                                          # the fusion logits as ONE launch over the un-concatenated parts (bist_switch_logits_fwd)
            score = torch.empty((p2[0].shape[0], ns_), device=W.device, dtype=W.dtype)
            arr = (C.c_void_p * n_)(*[p.data_ptr() for p in p2])
            check(lib.bist_switch_logits_fwd(arr, n_, W.data_ptr(), W.stride(0), _ptr(bias), score.data_ptr(), dtype_code(score.dtype), p2[0].shape[0], d, ns_,
                                             dtype_code(W.dtype), _stream()), "bist_switch_logits_fwd")
        else:
            score = None
            for j, pj in enumerate(p2):                  # concat order of the reference: query, cap, spatial, temporal
                score = ops.linear(pj, W[:, j * d:(j + 1) * d], bias if j == 0 else None, out=score, accumulate=j > 0)
        ctx.one = one
        xs = [p2[k] for k in xs_idx]
        out = ops.fuse_modalities(score, xs)
        ctx.save_for_backward(W, score, *p2)
        ctx.cfg = (tuple(xs_idx), [tuple(p.shape) for p in parts], bias.dtype if bias is not None else None)
        ctx.w_dst, ctx.b_dst = getattr(W, "_grad_view", None), (getattr(bias, "_acc32", None) if bias is not None else None)
        return out.view(parts[0].shape)

    @staticmethod
    def backward(ctx, dout):
        W, score, *p2 = ctx.saved_tensors
        xs_idx, shapes, bdt = ctx.cfg
        n, d = score.shape[-1], p2[0].shape[-1]
        M = p2[0].shape[0]
        dout = dout.reshape(M, d)
        if not dout.is_contiguous():
            dout = dout.contiguous()
        xs = [p2[k] for k in xs_idx]
        dscore = torch.empty_like(score)
        dxs = [torch.empty_like(x) for x in xs]
        xa = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        da = (C.c_void_p * n)(*[x.data_ptr() for x in dxs])
        check(lib.bist_fuse_modalities_bwd(score.data_ptr(), xa, dout.data_ptr(), dscore.data_ptr(), da, M, n, d, dtype_code(score.dtype), _stream()),
              "bist_fuse_modalities_bwd")
        from_sum = {k: dxs[j] for j, k in enumerate(xs_idx)}
        gw = ctx.w_dst
        if getattr(ctx, "one", False):
            # every part's gradient (dscore . W_j + the weighted sum's addend) in one launch, the weight and bias gradients in another
            # (bist_switch_logits_bwd) where a product per part took eight
            n_ = len(p2)
            dps = [torch.empty((M, d), device=pj.device, dtype=pj.dtype) for pj in p2]
            arr = (C.c_void_p * n_)(*[p.data_ptr() for p in p2])
            darr = (C.c_void_p * n_)(*[g.data_ptr() for g in dps])
            rarr = (C.c_void_p * n_)(*[_ptr(from_sum.get(j)) for j in range(n_)])
            wd = gw if gw is not None else torch.empty(W.shape, device=W.device, dtype=W.dtype)
            want_b = bdt is not None
            bd = (ctx.b_dst if ctx.b_dst is not None else torch.empty((n,), device=W.device, dtype=torch.float32)) if want_b else None
            if ops.LEAF_STREAM is not None and (ops.LEAF_MASK & 2) and gw is not None and (not want_b or ctx.b_dst is not None):
                # the parts' gradients on this chain; the weight / bias gradient (one slow launch on few workgroups: a leaf) off it
                check(lib.bist_switch_logits_bwd(arr, n_, W.data_ptr(), W.stride(0), dscore.data_ptr(), dtype_code(dscore.dtype), darr, rarr,
                                                None, 0, dtype_code(wd.dtype), 0, None, 0, M, d, n, dtype_code(W.dtype), _stream()),
                      "bist_switch_logits_bwd")
                with _Leaf(dscore, *p2):
                    check(lib.bist_switch_logits_bwd(arr, n_, W.data_ptr(), W.stride(0), dscore.data_ptr(), dtype_code(dscore.dtype), None, None,
                                                    wd.data_ptr(), wd.stride(0), dtype_code(wd.dtype), 1,
                                                    _ptr(bd), 1 if want_b else 0, M, d, n, dtype_code(W.dtype), _stream()),
                          "bist_switch_logits_bwd")
            else:
                check(lib.bist_switch_logits_bwd(arr, n_, W.data_ptr(), W.stride(0), dscore.data_ptr(), dtype_code(dscore.dtype), darr, rarr,
                                                wd.data_ptr(), wd.stride(0), dtype_code(wd.dtype), 1 if gw is not None else 0,
                                                _ptr(bd), 1 if (want_b and ctx.b_dst is not None) else 0, M, d, n, dtype_code(W.dtype), _stream()),
                      "bist_switch_logits_bwd")
            db = None
            if want_b and ctx.b_dst is None:
                db = _to_dtype_from_f32(bd, bdt)
            return (None if gw is not None else wd, db, None, *[g.view(shapes[j]) for j, g in enumerate(dps)])
        dW = None if gw is not None else torch.empty(W.shape, device=W.device, dtype=W.dtype)
        grads = []
        for j, pj in enumerate(p2):
            Wj = W[:, j * d:(j + 1) * d]
            dpj = torch.empty((M, d), device=pj.device, dtype=pj.dtype)
            res = from_sum.get(j)
            g_dx = ops.gemm_desc(dscore, Wj, dpj, M=M, N=d, K=n, a_rs=n, a_ks=1, b_rs=1, b_ks=Wj.stride(0), ldc=d, residual=res, ldr=d if res is not None else 0)
            check(lib.bist_gemm(C.byref(g_dx), _stream()), "bist_gemm")
            dst = gw[:, j * d:(j + 1) * d] if gw is not None else dW[:, j * d:(j + 1) * d]
            g_dw = ops.gemm_desc(dscore, pj, dst, M=n, N=d, K=M, a_rs=1, a_ks=n, b_rs=1, b_ks=pj.stride(0), ldc=dst.stride(0),
                                 residual=dst if gw is not None else None, ldr=dst.stride(0) if gw is not None else 0)
            check(lib.bist_gemm(C.byref(g_dw), _stream()), "bist_gemm")
            grads.append(dpj.view(shapes[j]))
        db = None
        if bdt is not None:
            acc = ctx.b_dst if ctx.b_dst is not None else _f32_zeros((n,), dscore)
            if ctx.b_dst is not None and ops.COLSUM_QUEUE is not None:
                ops.COLSUM_QUEUE.append((dscore, acc, M, n))
            else:
                check(lib.bist_col_sum_acc(dscore.data_ptr(), acc.data_ptr(), M, n, n, dtype_code(dscore.dtype), _stream()), "bist_col_sum_acc")
            db = None if ctx.b_dst is not None else _to_dtype_from_f32(acc, bdt)
        return (dW, db, None, *grads)


class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        if a.shape != b.shape:
            raise ValueError("AddFn: equal shapes only (broadcast adds carry no gradient on this path)")
        return ops.add(a, b)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class BucketMarkFn(Function):
    """Identity whose backward tells the trainer that the backward pass of everything recorded AFTER this node has been issued (the
    engine runs nodes in decreasing sequence number, i.e. in reverse program order): Fn.bucket_mark, the overlapped gradient exchange."""
    @staticmethod
    def forward(ctx, x, k, notify):
        ctx.k, ctx.notify = k, notify
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        ctx.notify(ctx.k)
        return dy, None, None


class AddDropoutFn(Function):
    """mode 0: a + dropout(b) (SublayerConnection.forward, modules.py:44); mode 1: dropout(a + b) with b a constant table
    (PositionalEncoding.forward, modules.py:144).  The backward regenerates the mask (bist_epilogue_bwd)."""

    @staticmethod
    def forward(ctx, a, b, drop, mode):
        ctx.drop, ctx.mode, ctx.same = drop, mode, a.shape == b.shape
        return ops.add_dropout(a.contiguous(), b.contiguous(), drop, mode)

    @staticmethod
    def backward(ctx, dy):
        dm = ops.dropout_mask_grad(dy, ctx.drop)
        if ctx.mode == 0:
            return dy, (dm if ctx.same else None), None, None
        return dm, (dm if ctx.same else None), None, None


class PermuteTSFn(Function):
    """[B,T,S,d] -> [B,S,T,d]; the gradient takes the same kernel back."""

    @staticmethod
    def forward(ctx, x):
        return ops.permute_ts(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.permute_ts(dy)


class FanOutFn(Function):
    """n aliases of one tensor for n consumers; backward sums their gradients in ONE pass (bist_add_n) where autograd's
    own accumulation makes n-1 pairwise passes.  The video tensor of the reasoning layers has 3 L consumers, each with a
    [B*T*S, d] gradient; the encoded texts feed every decoder layer."""

    @staticmethod
    def forward(ctx, x, n, join=False):
        ctx.n = n
        ctx.join = bool(join)
        ctx.set_materialize_grads(False)        # aliases nobody read carry no gradient: None, not a zero tensor autograd would fill and we would sum
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if ctx.join and len(gs) > 1 and gs[0].is_cuda:
            # join=True (a fan made on the MAIN stream whose consumers run on the side streams too): the sum waits for the present position of
            # every side stream.  The engine orders a gradient behind its producing NODE's stream; measured with a fan for encoded_tgt, that
            # did not cover a gradient arriving from the caption stream in the replayed step (it ran 0.4 ms faster and updated unrelated
            # weights differently from the eager step; with this wait both agree bit for bit)
            from . import functional as Fn_
            if Fn_.CONCURRENT:
                cur = torch.cuda.current_stream()
                for st in Fn_.live_side_streams():
                    if st != cur:
                        cur.wait_stream(st)
        for g in gs:                          # a gradient finished on another stream than its node's carries the event to wait on
            ev = getattr(g, "_bist_ready", None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
        if not gs:
            return None, None, None
        if len(gs) == 1:
            return gs[0], None, None
        if len({g.dtype for g in gs}) > 1:
            gs = [g.to(gs[0].dtype) for g in gs]
        return ops.add_n(gs), None, None


class CastFn(Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return ops.cast(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        return ops.cast(dy.contiguous(), ctx.src), None


# ----------------------------------------------------------------------------------------------
# attention cores
# ----------------------------------------------------------------------------------------------
class MhaCoreFn(Function):
    """mha_core over (possibly packed) projections.  mode: 'qkv' (a = [N,L,3d]), 'q_kv' (a = q, b = [N,Lk,2d]),
    'q_k_v' (a, b, c).  Returns (O, P or None); a gradient arriving for P is honoured (pointer generator)."""

    @staticmethod
    def forward(ctx, a, b, c, mode, mask, h, want_p, drop=None):
        q, k, v = MhaCoreFn._views(a, b, c, mode)
        m8 = _mask_u8(mask)
        o, p = ops.mha_core(q, k, v, m8, h, want_p=want_p, drop=drop)
        ctx.save_for_backward(a, b, c, m8)
        ctx.cfg = (mode, h)
        ctx.drop = drop
        if p is None:
            ctx.mark_non_differentiable()
            return o, None
        return o, p

    @staticmethod
    def _views(a, b, c, mode):
        if mode == "qkv":
            d = a.shape[-1] // 3
            return a[..., :d], a[..., d:2 * d], a[..., 2 * d:]
        if mode == "q_kv":
            d = a.shape[-1]
            return a, b[..., :d], b[..., d:]
        return a, b, c

    @staticmethod
    def backward(ctx, do, dp):
        a, b, c, m8 = ctx.saved_tensors
        mode, h = ctx.cfg
        q, k, v = MhaCoreFn._views(a, b, c, mode)
        N, Lq, d = q.shape
        Lk = k.shape[1]
        dk_ = d // h
        ga = torch.empty(a.shape, device=a.device, dtype=a.dtype)
        gb = torch.empty(b.shape, device=a.device, dtype=b.dtype) if b is not None else None
        gc = torch.empty(c.shape, device=a.device, dtype=c.dtype) if c is not None else None
        dq, dkk, dv = MhaCoreFn._views(ga, gb, gc, mode)
        if do is not None:
            do = do.contiguous()
            if do.dtype != q.dtype:
                do = ops.cast(do, q.dtype)
        if dp is not None:
            dp = dp.contiguous().float() if dp.dtype != torch.float32 else dp.contiguous()
        mbs = mqs = 0
        if m8 is not None:
            mbs = m8.stride(0) if m8.shape[0] > 1 else 0
            mqs = m8.stride(1) if m8.shape[1] > 1 else 0
        check(lib.bist_mha_core_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), _ptr(m8), _ptr(do), _ptr(dp),
                                    dq.data_ptr(), dkk.data_ptr(), dv.data_ptr(), N, Lq, Lk, h, dk_,
                                    q.stride(1), k.stride(1), v.stride(1), d, q.stride(0), k.stride(0), v.stride(0), Lq * d,
                                    dq.stride(1), dkk.stride(1), dv.stride(1), dq.stride(0), dkk.stride(0), dv.stride(0),
                                    mbs, mqs, 1.0 / math.sqrt(dk_), ops.drop_ref(ctx.drop), dtype_code(q.dtype), _stream()),
              "bist_mha_core_bwd")
        return ga, gb, gc, None, None, None, None, None


class StStage1PvFn(Function):
    @staticmethod
    def forward(ctx, scores, v, tmask, dims, direction, drop=None):
        B, T, S, Lq, h, dk = dims
        m8 = _mask_u8(tmask.reshape(B, T if direction == 0 else S)) if tmask is not None else None      # one entry per key
        out = ops.st_stage1_pv(scores, v, m8, B=B, T=T, S=S, Lq=Lq, h=h, dk=dk, direction=direction, drop=drop)
        ctx.save_for_backward(scores, v, m8)
        ctx.cfg = (dims, direction)
        ctx.drop = drop
        ctx.to_scores_fn = type(getattr(scores, "grad_fn", None)).__name__ == "StScoresFnBackward"
        return out

    @staticmethod
    def backward(ctx, do):
        scores, v, m8 = ctx.saved_tensors
        (B, T, S, Lq, h, dk), direction = ctx.cfg
        d = h * dk
        do = do.contiguous()
        # When the scores come from StScoresFn, its backward takes the gradient as a bf16 GEMM operand: the kernel writes
        # dscores in bf16 and the tensor travels as an attribute of an (uninitialised, never read) f32 placeholder --
        # autograd would cast a bf16 gradient of an f32 tensor back to f32.  Saves the f32 -> bf16 pass per direction.
        direct16 = ctx.to_scores_fn and v.dtype == torch.bfloat16
        dsc = torch.empty(scores.shape, device=scores.device, dtype=torch.bfloat16 if direct16 else scores.dtype)
        dv = torch.empty((B, T, S, d), device=v.device, dtype=v.dtype)
        check(lib.bist_st_stage1_pv_bwd(scores.data_ptr(), v.data_ptr(), _ptr(m8), do.data_ptr(), dsc.data_ptr(), dtype_code(dsc.dtype), dv.data_ptr(),
                                        B, T, S, Lq, h, dk, v.stride(-2), d, direction, ops.drop_ref(ctx.drop), dtype_code(v.dtype),
                                        _stream()),
              "bist_st_stage1_pv_bwd")
        if direct16:
            holder = torch.empty_like(scores)
            holder._bist_dsc = dsc
            dsc = holder
        return dsc, dv, None, None, None, None


class St1FusedTrainFn(Function):
    """Stage 1 of one direction (A1 / A4), training: y = x + drop(W_o . MHA(LN(x), X_g, X_g) + b_o) on the expanded query, forward as ONE
    launch (bist_st_stage1_fused_train_fwd: value projection, scores, masked softmax + dropout, P.V, output projection + dropout +
    residual) that leaves V, the probabilities and the context rows behind for this backward -- the same kernels the unfused path runs:
    group sum + dropout mask of dy, output projection dX / dW, bist_st_stage1_pv_bwd_p (probabilities instead of a score gather and a
    softmax recompute), the two score-product gradients and the value projection's dX / dW.  No fp32 score tensor exists at any point.
    vft_a / vft_b: two aliases of the video tensor (its gradient through the scores and through the value projection)."""

    @staticmethod
    def forward(ctx, qf, x, vft_a, vft_b, v_in, tmask, wv, bv, wo, bo, wv_frag, wo_frag, cfg):
        h, direction, adrop, sdrop = cfg[:4]
        B, T, S, d = vft_a.shape
        K = T if direction == 0 else S
        m8 = _mask_u8(tmask.reshape(B, K)) if tmask is not None else None
        y, v, p, o = ops.st_stage1_fused_train(qf.reshape(B, -1, d), vft_a, m8, wv_frag, bv, wo_frag, bo, x, h=h, direction=direction,
                                               attn_drop=adrop, sub_drop=sdrop, want_v=v_in is None)
        ctx.own_v = v_in is None
        ctx.save_for_backward(qf, vft_a, v if v_in is None else v_in, p, o, m8, wv, wo)
        ctx.cfg = (cfg, tuple(x.shape), bv.dtype, tuple(qf.shape))
        ctx.w_dst = (getattr(wv, "_grad_view", None), getattr(wo, "_grad_view", None))
        ctx.b_dst = (getattr(bv, "_acc32", None), getattr(bo, "_acc32", None))
        ctx.set_materialize_grads(False)
        return y, x          # x again for its next consumer (stage 2's sublayer): that gradient comes back here and is folded into the group sum

    @staticmethod
    def backward(ctx, dy, dxp):
        qf, vft, v, p, o, m8, wv, wo = ctx.saved_tensors
        cfg_, x_shape, bdt, qf_shape = ctx.cfg
        h, direction, adrop, sdrop = cfg_[:4]
        offload = bool(cfg_[4]) if len(cfg_) > 4 else False
        B, T, S, d = vft.shape
        dk = d // h
        G = S if direction == 0 else T
        Lq = x_shape[1]
        M, TS, R = B * G * Lq, T * S, Lq * h
        dev, dt = vft.device, vft.dtype
        if dy is None:
            dy = torch.zeros((M, d), device=dev, dtype=dt)
        dy = dy.reshape(M, d)
        if not dy.is_contiguous():
            dy = dy.contiguous()
        if dxp is not None:
            dxp = dxp.reshape(B * Lq, d)
            if not dxp.is_contiguous() or dxp.dtype != dt:
                dxp = dxp.to(dt).contiguous()
        dres = torch.empty((B * Lq, d), device=dev, dtype=dt)          # gradient of the un-expanded query: the sum over the groups (+ its other gradient)
        dz = dy
        if sdrop is not None and sdrop[0] > 0 and (Lq * d * dy.element_size()) % 16 == 0 and dy.data_ptr() % 16 == 0 and (dxp is None or dxp.data_ptr() % 16 == 0):
            dz = torch.empty_like(dy)                                   # ... and the dropout-masked gradient, in the same pass over dy
            check(lib.bist_group_sum_mask(dy.data_ptr(), _ptr(dxp), dres.data_ptr(), dz.data_ptr(), B, G, Lq * d, ops.drop_ref(sdrop), dtype_code(dt), _stream()),
                  "bist_group_sum_mask")
        else:
            check(lib.bist_group_sum_add(dy.data_ptr(), _ptr(dxp), dres.data_ptr(), B, G, Lq * d, dtype_code(dt), _stream()), "bist_group_sum_add")
            if sdrop is not None and sdrop[0] > 0:
                dz = torch.empty_like(dy)
                check(lib.bist_epilogue_bwd(dy.data_ptr(), dy.data_ptr(), dz.data_ptr(), M, d, d, d, d, ACT_NONE, sdrop[0], sdrop[1] & 0xFFFFFFFFFFFFFFFF,
                                            _ptr(ops.DROP_CTR), dtype_code(dt), _stream()), "bist_epilogue_bwd")
        # Off the chain (offload: this direction runs on the MAIN stream of a several-stream step and the trainer owns the weight gradients):
        # the output projection's weight gradient and the video tensor's gradient through the scores are not inputs of anything on this
        # direction's chain -- they go to the caption / decoder stream; the video gradient carries the event its consumer (the one-pass sum of
        # the video tensor's gradients, FanOutFn) waits on, the weight gradient is joined with every side stream at the end of the backward pass.
        side = None
        if ops.LEAF_STREAM is not None and (ops.LEAF_MASK & 1) and ctx.w_dst[1] is not None and ctx.b_dst[1] is not None and dev.type == "cuda":
            side = ops.LEAF_STREAM              # split executor: a chain of its own for the leaves of BOTH directions
        elif offload and ctx.w_dst[1] is not None and ctx.b_dst[1] is not None and dev.type == "cuda":
            from . import functional as Fn_
            if Fn_.CONCURRENT:
                side = Fn_.side_stream(1)
        o2 = o.view(M, d)
        dvft_a = torch.empty((B, TS, d), device=dev, dtype=dt)          # (escapes: allocated on this stream's pool)
        if side is None:
            do, dwo, dbo = _linear_grads(o2, wo, dz, 1.0, ctx.w_dst[1], ctx.b_dst[1], bdt, True, ctx.needs_input_grad[8], ctx.needs_input_grad[9])
        else:
            do, _, _ = _linear_grads(o2, wo, dz, 1.0, None, None, None, True, False, False)
            dwo = dbo = None
        dsc = torch.empty((B, R, TS), device=dev, dtype=dt)
        dv = torch.empty((B, T, S, d), device=dev, dtype=dt)
        check(lib.bist_st_stage1_pv_bwd_p(p.data_ptr(), p.shape[-1], v.data_ptr(), _ptr(m8), do.data_ptr(), dsc.data_ptr(), dtype_code(dt), dv.data_ptr(),
                                          B, T, S, Lq, h, dk, v.stride(-2), d, direction, ops.drop_ref(adrop), dtype_code(dt), _stream()), "bist_st_stage1_pv_bwd_p")
        q3, vf = qf.reshape(B, R, d), vft.reshape(B, TS, d)
        dqf = torch.empty((B, R, d), device=dev, dtype=dt)
        ops.gemm(dsc, vf, dqf, M=R, N=d, K=TS, a_rs=TS, a_ks=1, b_rs=1, b_ks=vf.stride(1), ldc=d, batch=(B, 1),
                 a_bs=(R * TS, 0), b_bs=(vf.stride(0), 0), c_bs=(R * d, 0))

        def video_grad():
            ops.gemm(dsc, q3, dvft_a, M=TS, N=d, K=R, a_rs=1, a_ks=TS, b_rs=1, b_ks=q3.stride(1), ldc=d, batch=(B, 1),
                     a_bs=(R * TS, 0), b_bs=(q3.stride(0), 0), c_bs=(TS * d, 0))
        if side is None:
            video_grad()
        else:
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            ops.WGRAD_KEEP.append((dz, o2, dsc, q3, dvft_a))
            for t_ in (dz, o2, dsc, q3, dvft_a):
                t_.record_stream(side)              # read / written by the side stream after this call has released them
            with torch.cuda.stream(side):
                _linear_grads(o2, wo, dz, 1.0, ctx.w_dst[1], ctx.b_dst[1], bdt, False, True, ctx.needs_input_grad[9])
                video_grad()
                ev = torch.cuda.Event()
                ev.record(side)
            dvft_a._bist_ready = ev
        gva = dvft_a.view(vft.shape)
        if getattr(dvft_a, "_bist_ready", None) is not None:
            gva._bist_ready = dvft_a._bist_ready
        if not ctx.own_v:            # the value projection is a product of its own (another stream): its backward takes dV from here
            return (dqf.view(qf_shape), dres.view(x_shape), gva, None, dv, None, None, None, dwo, dbo, None, None, None)
        with _Leaf(vft, wv, dv, on=bool(ops.LEAF_MASK & 1)) as lf:        # the launch projected the values itself: the projection's two backward products are leaves too
            dvft_b, dwv, dbv = _linear_grads(vft.view(B * TS, d), wv, dv.view(B * TS, d), 1.0, ctx.w_dst[0], ctx.b_dst[0], bdt, True,
                                             ctx.needs_input_grad[6], ctx.needs_input_grad[7])
            ev_b = lf.ready()
        gvb = dvft_b.view(vft.shape)
        if ev_b is not None:
            gvb._bist_ready = ev_b
            ops.WGRAD_KEEP.append((dvft_b,))
        return (dqf.view(qf_shape), dres.view(x_shape), gva, gvb, None, None, dwv, dbv, dwo, dbo, None, None, None)


class StStage2Fn(Function):
    """(PY, rowsum): rowsum [B,Lq,h] f32 = sum_g P'[g] is only produced under dropout (None otherwise)."""

    @staticmethod
    def forward(ctx, q2f, y, gmask, h, drop=None):
        B, G = y.shape[0], y.shape[1]
        m8 = _mask_u8(gmask.reshape(B, G)) if gmask is not None else None
        res = ops.st_stage2(q2f, y, m8, h=h, drop=drop)
        ctx.save_for_backward(q2f, y, m8)
        ctx.h, ctx.drop = h, drop
        ctx.set_materialize_grads(False)
        if isinstance(res, tuple):
            return res
        ctx.mark_non_differentiable()
        return res, None

    @staticmethod
    def backward(ctx, dpy, drs):
        q2f, y, m8 = ctx.saved_tensors
        B, G, Lq, d = y.shape
        dpy = torch.zeros((B, Lq, ctx.h, d), device=y.device, dtype=y.dtype) if dpy is None else dpy.contiguous()
        if drs is not None:
            drs = drs.contiguous().float()
        dq = torch.empty_like(q2f)
        dy = torch.empty_like(y)
        check(lib.bist_st_stage2_bwd(q2f.data_ptr(), y.data_ptr(), _ptr(m8), dpy.data_ptr(), _ptr(drs), dq.data_ptr(), dy.data_ptr(),
                                     B, G, Lq, ctx.h, d, ops.drop_ref(ctx.drop), dtype_code(y.dtype), _stream()), "bist_st_stage2_bwd")
        return dq, dy, None, None, None


class ScaledBiasFn(Function):
    """x + s[m, head] * bias   (the value bias of stage 2 when the probabilities were dropped)."""

    @staticmethod
    def forward(ctx, x, s, bias, h):
        ctx.save_for_backward(s, bias)
        ctx.h = h
        ctx.b_dst = getattr(bias, "_acc32", None)
        return ops.scaled_bias(x, s, bias, h)

    @staticmethod
    def backward(ctx, dy):
        s, bias = ctx.saved_tensors
        h = ctx.h
        d = dy.shape[-1]
        dy2 = dy.reshape(-1, d).contiguous()
        ds = torch.empty(s.shape, device=s.device, dtype=torch.float32)
        acc = ctx.b_dst if ctx.b_dst is not None else _f32_zeros((d,), dy2)
        check(lib.bist_scaled_bias_bwd(dy2.data_ptr(), s.data_ptr(), bias.data_ptr(), ds.data_ptr(), acc.data_ptr(), dy2.shape[0], h,
                                       d // h, dtype_code(dy2.dtype), _stream()), "bist_scaled_bias_bwd")
        return dy, ds, (None if ctx.b_dst is not None else _to_dtype_from_f32(acc, bias.dtype)), None


# ----------------------------------------------------------------------------------------------
# output heads and loss
# ----------------------------------------------------------------------------------------------
class PointerMixFn(Function):
    @staticmethod
    def forward(ctx, logits, sw, Lt, sigmoid_switch, n, *rest):
        ps, texts = rest[:n], rest[n:]
        out = ops.pointer_mix(logits, sw, ps, texts, Lt, sigmoid_switch)
        ctx.save_for_backward(logits, sw, out, *ps, *texts)
        ctx.cfg = (Lt, sigmoid_switch, n)
        return out

    @staticmethod
    def backward(ctx, dout):
        Lt, sig, n = ctx.cfg
        logits, sw, out, *rest = ctx.saved_tensors
        ps, texts = rest[:n], rest[n:]
        rows, V = logits.shape
        dout = dout.contiguous()
        logits, sw = logits.contiguous(), sw.contiguous()
        p2 = [p.reshape(rows, -1).contiguous() for p in ps]
        t2 = [t.contiguous() for t in texts]
        dlogits, dsw = torch.empty_like(logits), torch.empty_like(sw)
        dps = [torch.empty_like(p) for p in p2]
        pp = (C.c_void_p * n)(*[p.data_ptr() for p in p2])
        tt = (C.c_void_p * n)(*[t.data_ptr() for t in t2])
        ll = (C.c_int32 * n)(*[p.shape[1] for p in p2])
        dd = (C.c_void_p * n)(*[p.data_ptr() for p in dps])
        check(lib.bist_pointer_mix_bwd(logits.data_ptr(), sw.data_ptr(), n, pp, tt, ll, out.data_ptr(), dout.data_ptr(),
                                       dlogits.data_ptr(), dsw.data_ptr(), dd, rows, Lt, V, 1 if sig else 0, _stream()),
              "bist_pointer_mix_bwd")
        return (dlogits, dsw, None, None, None, *[g.view(p.shape) for g, p in zip(dps, ps)], *([None] * n))


class SumTermsFn(Function):
    """Sum of same-shaped f32 terms in one launch (bist_add_n); every term receives the upstream gradient itself."""

    @staticmethod
    def forward(ctx, *terms):
        ctx.n = len(terms)
        return ops.add_n([t.contiguous() for t in terms])

    @staticmethod
    def backward(ctx, g):
        return (g,) * ctx.n


class StackRowsFn(Function):
    """n <= 4 same-shaped tensors [..., d] behind one another as [n * rows, d] (one launch, bist_stack_rows); the gradient's slices go
    back as views."""

    @staticmethod
    def forward(ctx, *xs):
        ctx.shapes = [tuple(x.shape) for x in xs]
        xs = [x.reshape(-1, x.shape[-1]) for x in xs]
        xs = [x if x.is_contiguous() else x.contiguous() for x in xs]
        rows, d = xs[0].shape
        out = torch.empty((len(xs) * rows, d), device=xs[0].device, dtype=xs[0].dtype)
        arr = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
        check(lib.bist_stack_rows(arr, len(xs), out.data_ptr(), rows * d * out.element_size(), _stream()), "bist_stack_rows")
        ctx.n, ctx.rows = len(xs), rows
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return tuple(g[j * ctx.rows:(j + 1) * ctx.rows].view(ctx.shapes[j]) for j in range(ctx.n))


class XentSmoothLossFn(Function):
    """G losses (one [1] tensor each) of G groups of M rows that share their targets: sum_rows KL(smoothed target || softmax(logits)) /
    denom straight from the f32 logits [G * M, V] (bist_xent_smooth_fwd / _bwd: no log-probabilities, no f32 gradient of them).  The
    logits' gradient leaves in the operand dtype of the producing projection and travels as `_bist_dz` of an uninitialised f32
    placeholder (LinearFn.backward takes it from there: autograd would cast a bf16 gradient of an f32 tensor back to f32)."""

    @staticmethod
    def forward(ctx, logits, target, denom, smoothing, pad, G, grad_dtype):
        R, V = logits.shape
        M = R // G
        logits, target = logits.contiguous(), target.contiguous()
        rows = torch.empty((R,), device=logits.device, dtype=torch.float32)
        lse = torch.empty((R,), device=logits.device, dtype=torch.float32)
        check(lib.bist_xent_smooth_fwd(logits.data_ptr(), target.data_ptr(), M, R, V, smoothing, pad, rows.data_ptr(), lse.data_ptr(), _stream()),
              "bist_xent_smooth_fwd")
        out = torch.empty((G, 4), device=logits.device, dtype=torch.float32)           # (one 16-byte slot per term: the terms feed bist_add_n)
        check(lib.bist_sum_div_groups(rows.data_ptr(), M, G, _ptr(denom), out.data_ptr(), 4, _stream()), "bist_sum_div_groups")
        ctx.save_for_backward(logits, lse, target, denom)
        ctx.cfg = (M, G, smoothing, pad, grad_dtype)
        return tuple(out[g, :1] for g in range(G))

    @staticmethod
    def backward(ctx, *gouts):
        logits, lse, target, denom = ctx.saved_tensors
        M, G, smoothing, pad, grad_dtype = ctx.cfg
        R, V = logits.shape
        gs = [g if g is not None else torch.zeros(1, device=logits.device) for g in gouts]
        if all(g.data_ptr() == gs[0].data_ptr() for g in gs) and gs[0].numel() == 1 and gs[0].dtype == torch.float32:
            gout, gstride = gs[0], 0                       # the terms were summed: ONE upstream gradient, read in place
        else:
            gout, gstride = torch.cat([g.reshape(1).float() for g in gs]), 1
        dl = torch.empty((R, V), device=logits.device, dtype=grad_dtype)
        check(lib.bist_xent_smooth_bwd(logits.data_ptr(), lse.data_ptr(), target.data_ptr(), M, R, gout.data_ptr(), gstride, _ptr(denom), dl.data_ptr(),
                                       dtype_code(grad_dtype), V, smoothing, pad, _stream()), "bist_xent_smooth_bwd")
        if grad_dtype == torch.float32:
            return dl, None, None, None, None, None, None
        holder = torch.empty((R, V), device=logits.device, dtype=torch.float32)       # never read: the consumer takes _bist_dz
        holder._bist_dz = (dl, 0.0, 0)
        return holder, None, None, None, None, None, None


class SwitchLogitsFn(Function):
    """lin(cat(parts, -1)) of the pointer generators' switch (generator.py:69-71, 119-121) without the concatenation: one launch forward,
    two backward (bist_switch_logits_fwd / _bwd); parts: n <= 4 tensors [rows, d]; returns f32 [rows, ns].  Under the trainer the weight
    gradient accumulates into the flat gradient view and the bias gradient into its fp32 accumulator, like LinearFn."""

    @staticmethod
    def forward(ctx, w, bias, out_dtype, *parts):
        ctx.shapes = [tuple(p.shape) for p in parts]
        parts = [p.reshape(-1, p.shape[-1]) for p in parts]
        parts = [p if p.is_contiguous() else p.contiguous() for p in parts]
        rows, d = parts[0].shape
        ns, n = w.shape[0], len(parts)
        out = torch.empty((rows, ns), device=w.device, dtype=out_dtype or torch.float32)
        arr = (C.c_void_p * n)(*[p.data_ptr() for p in parts])
        check(lib.bist_switch_logits_fwd(arr, n, w.data_ptr(), w.stride(0), _ptr(bias), out.data_ptr(), dtype_code(out.dtype), rows, d, ns,
                                         dtype_code(w.dtype), _stream()), "bist_switch_logits_fwd")
        ctx.save_for_backward(w, *parts)
        ctx.w_dst = getattr(w, "_grad_view", None)
        ctx.b_dst = getattr(bias, "_acc32", None) if bias is not None else None
        ctx.bias_dtype = bias.dtype if bias is not None else None
        return out

    @staticmethod
    def backward(ctx, dsw):
        w, *parts = ctx.saved_tensors
        rows, d = parts[0].shape
        ns, n = w.shape[0], len(parts)
        dsw = dsw.contiguous() if dsw.dtype in (torch.float32, w.dtype) else dsw.float().contiguous()
        need = ctx.needs_input_grad
        dparts = [torch.empty_like(p) if need[3 + j] else None for j, p in enumerate(parts)]
        arr = (C.c_void_p * n)(*[p.data_ptr() for p in parts])
        darr = (C.c_void_p * n)(*[_ptr(g) for g in dparts])
        dw = db = None
        has_bias = ctx.bias_dtype is not None
        if ctx.w_dst is not None:
            wd, acc_w = ctx.w_dst, 1
        else:
            wd, acc_w = (torch.empty_like(w) if need[0] else None), 0
            dw = wd
        if has_bias and need[1] and wd is not None:
            bd, acc_b = (ctx.b_dst, 1) if ctx.b_dst is not None else (torch.empty((ns,), device=w.device, dtype=torch.float32), 0)
        else:
            bd, acc_b = None, 0
        check(lib.bist_switch_logits_bwd(arr, n, w.data_ptr(), w.stride(0), dsw.data_ptr(), dtype_code(dsw.dtype),
                                        darr if any(g is not None for g in dparts) else None, None,
                                        _ptr(wd), wd.stride(0) if wd is not None else 0, dtype_code(wd.dtype) if wd is not None else 0, acc_w,
                                        _ptr(bd), acc_b, rows, d, ns, dtype_code(w.dtype), _stream()), "bist_switch_logits_bwd")
        if has_bias and need[1] and ctx.b_dst is None and bd is not None:
            db = bd.to(ctx.bias_dtype)
        return (dw, db, None, *[g.view(sh) if g is not None else None for g, sh in zip(dparts, ctx.shapes)])


class PointerAttnFn(Function):
    """(p, tv) of a pointer attention (generator.py:106-118): p = softmax(q.k^T / sqrt(d), masked -1e9) f32 [B,Lt,L] over one head of d
    channels, tv = p . enc (the text vector).  mask [B or 1, L] uint8; text int64 [B,L] or None (then position t also needs
    text[b,t] != unk, generator.py:106-107).  One launch forward, one backward (bist_pointer_attn_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, q, k, enc, m8, text, unk):
        B, Lt, d = q.shape
        L = k.shape[1]
        q, k, enc = q.contiguous(), k.contiguous(), enc.contiguous()
        p = torch.empty((B, Lt, L), device=q.device, dtype=torch.float32)
        tv = torch.empty((B, Lt, d), device=q.device, dtype=enc.dtype)
        scale = 1.0 / math.sqrt(d)
        check(lib.bist_pointer_attn_fwd(q.data_ptr(), k.data_ptr(), enc.data_ptr(), _ptr(m8), (L if (m8 is not None and m8.shape[0] > 1) else 0),
                                        _ptr(text), int(unk), p.data_ptr(), tv.data_ptr(), B, Lt, L, d, scale, dtype_code(q.dtype), _stream()),
              "bist_pointer_attn_fwd")
        ctx.save_for_backward(q, k, enc, p)
        ctx.scale = scale
        return p, tv

    @staticmethod
    def backward(ctx, dp, dtv):
        q, k, enc, p = ctx.saved_tensors
        B, Lt, d = q.shape
        L = k.shape[1]
        dq, dk, denc = torch.empty_like(q), torch.empty_like(k), torch.empty_like(enc)
        if dp is None and dtv is None:
            return dq.zero_(), dk.zero_(), denc.zero_(), None, None, None
        if dp is not None:
            dp = dp.contiguous() if dp.dtype == torch.float32 else dp.float().contiguous()
        if dtv is not None:
            dtv = dtv.contiguous() if dtv.dtype == enc.dtype else dtv.to(enc.dtype).contiguous()
        else:
            denc.zero_()
        check(lib.bist_pointer_attn_bwd(q.data_ptr(), k.data_ptr(), enc.data_ptr(), p.data_ptr(), _ptr(dp), _ptr(dtv), dq.data_ptr(), dk.data_ptr(),
                                        denc.data_ptr(), B, Lt, L, d, ctx.scale, dtype_code(q.dtype), _stream()), "bist_pointer_attn_bwd")
        return dq, dk, denc, None, None, None


class LogSoftmaxFn(Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.log_softmax(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        V = y.shape[-1]
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        check(lib.bist_log_softmax_bwd(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel() // V, V, _stream()), "bist_log_softmax_bwd")
        return dx


class LabelSmoothingLossFn(Function):
    """sum_rows KL(smoothed target || exp(logp)) / denom  -> device scalar [1]."""

    @staticmethod
    def forward(ctx, logp, target, denom, smoothing, pad):
        rows = ops.label_smoothing_rows(logp, target, smoothing, pad)
        ctx.save_for_backward(target.contiguous(), denom)
        ctx.cfg = (tuple(logp.shape), smoothing, pad)
        return ops.sum_div(rows, denom)

    @staticmethod
    def backward(ctx, gout):
        target, denom = ctx.saved_tensors
        (rows, V), smoothing, pad = ctx.cfg
        gout = gout.contiguous().float()
        dlogp = torch.empty((rows, V), device=gout.device, dtype=torch.float32)
        check(lib.bist_label_smoothing_bwd(target.data_ptr(), gout.data_ptr(), _ptr(denom), dlogp.data_ptr(), rows, V, smoothing,
                                           pad, _stream()), "bist_label_smoothing_bwd")
        return dlogp, None, None, None, None
