"""Tensor-level wrappers over the C ABI of libbist_hip.so.

PyTorch is plumbing here: it owns device memory and the current HIP stream; every arithmetic
step of the hot path is a kernel of libbist_hip.so launched on that stream.  No wrapper has a
PyTorch/CPU fallback: a CPU tensor, a missing library or an unsupported dtype raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, BF16, F32, BistColSum, BistDrop, BistGemm, BistLnBwdSet, BistLnGrad, BistLnSet, check, lib

Tensor = torch.Tensor

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dtype_code(t: torch.dtype) -> int:
    try:
        return _DT[t]
    except KeyError:
        raise TypeError(f"bist_amd: unsupported dtype {t} (float32 or bfloat16 only)") from None


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The HIP stream torch would launch on now, as the integer the C ABI takes.  Through torch's raw accessors when they exist: building a
    torch.cuda.Stream object per launch costs ~7 us of host time, a third of an eager launch (the capture passes of a new decode geometry)."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return _RAW_STREAM(_RAW_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts: Optional[Tensor]) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("bist_amd: operands must live on the MI355X (got a CPU tensor); there is no CPU path")


def _ptr(t: Optional[Tensor], off_elems: int = 0) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr() + off_elems * t.element_size()


def gemm_desc(a: Tensor, b: Tensor, c: Tensor, *, M: int, N: int, K: int, a_rs: int, b_rs: int, ldc: int,
              a_ks: int = 1, b_ks: int = 1, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None, ldr: int = 0,
              alpha: float = 1.0, act: int = ACT_NONE, batch: Tuple[int, int] = (1, 1),
              a_bs: Tuple[int, int] = (0, 0), b_bs: Tuple[int, int] = (0, 0), c_bs: Tuple[int, int] = (0, 0),
              r_bs: Tuple[int, int] = (0, 0), bias_bs2: int = 0, bias_bs1: int = 0, res_map: Tuple[int, int] = (0, 0),
              a_off: int = 0, b_off: int = 0, c_off: int = 0, bias_off: int = 0, r_off: int = 0,
              drop_p: float = 0.0, drop_seed: int = 0) -> BistGemm:
    _dev(a, b, c, bias, residual)
    if a.dtype != b.dtype:
        raise TypeError("bist_amd.gemm: A and B dtypes differ")
    if bias is not None and bias.dtype != a.dtype:
        raise TypeError("bist_amd.gemm: bias dtype must equal the operand dtype")
    if residual is not None and residual.dtype != c.dtype:
        raise TypeError("bist_amd.gemm: residual dtype must equal the output dtype")
    g = BistGemm()
    g.A, g.B, g.C = _ptr(a, a_off), _ptr(b, b_off), _ptr(c, c_off)
    g.bias, g.residual = _ptr(bias, bias_off), _ptr(residual, r_off)
    g.M, g.N, g.K = M, N, K
    g.a_rs, g.a_ks, g.b_rs, g.b_ks, g.ldc, g.ldr = a_rs, a_ks, b_rs, b_ks, ldc, ldr
    g.batch1, g.batch2 = batch
    g.a_bs1, g.a_bs2 = a_bs
    g.b_bs1, g.b_bs2 = b_bs
    g.c_bs1, g.c_bs2 = c_bs
    g.r_bs1, g.r_bs2 = r_bs
    g.bias_bs2 = bias_bs2
    g.bias_bs1 = bias_bs1
    g.alpha, g.act = alpha, act
    g.res_outer, g.res_inner = res_map
    g.in_dtype, g.out_dtype = dtype_code(a.dtype), dtype_code(c.dtype)
    g.drop_p, g.drop_seed = drop_p, drop_seed
    g.drop_ctr = DROP_CTR.data_ptr() if (DROP_CTR is not None and drop_p > 0) else None
    ws = _workspace(a.device)
    g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    return g


DROP_CTR = None   # optional device int64 step counter mixed into every dropout seed (set by the trainer)
_WS = {}
WORKSPACE_BYTES = 64 << 20


def _workspace(device) -> Tensor:
    """Per-(device, stream) fp32 scratch for split-K partial tiles (allocated once; the library never
    allocates).  One buffer per stream: GEMMs of concurrent branches must not share partial slabs."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        # only the head (split-K ticket counters, 4 KiB) must start as zeros; a captured hipGraph replays this fill, so it is
        # kept to the head instead of the whole 64 MiB (10 us per replay per stream)
        ws = torch.empty(WORKSPACE_BYTES // 4, device=device, dtype=torch.float32)
        ws[:4096].zero_()
        _WS[key] = ws
    return ws


def gemm(a: Tensor, b: Tensor, c: Tensor, **kw) -> Tensor:
    """C = epilogue(alpha * A.B^T) with explicit element strides (see include/bist_hip.h)."""
    g = gemm_desc(a, b, c, **kw)
    if GEMM_TIMING is not None and (GEMM_TIMING_SHAPE is None or GEMM_TIMING_SHAPE == (g.M, g.N, g.K)):
        # bench.py: HIP events around the launch, on the launch stream.  An eager pass is host-bound (~10 us of Python per
        # launch), so the device would reach e0 long before the kernel is even submitted and the interval would include
        # that host latency; a ~100 us spin queued first lets the host run ahead, and e0 / kernel / e1 execute back to back.
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200_000)
        e0.record()
        check(lib.bist_gemm(C.byref(g), _stream()), "bist_gemm")
        e1.record()
        GEMM_TIMING.append(((g.M, g.N, g.K, g.batch1 * g.batch2), e0, e1, int(lib.bist_gemm_is_fast(C.byref(g)))))
        return c
    check(lib.bist_gemm(C.byref(g), _stream()), "bist_gemm")
    return c


WGRAD_STREAM = None       # trainer: side stream of the weight-gradient GEMMs (autograd._WeightGradStream); None = in line
WGRAD_KEEP = []           # their operands, kept alive until the trainer joins that stream
LEAF_STREAM = None        # trainer: the stream of the backward pass's leaf products (autograd._Leaf); None = in line
LEAF_MASK = 0             # which leaves go there (bist_amd/train.py: LEAF_MASK)
def gemm_pair(ga: BistGemm, gb: BistGemm) -> None:
    """Two independent products (descriptors from gemm_desc) in one call; one launch when they are the small dX / dW
    pair of a linear layer's backward (bist_gemm_pair)."""
    check(lib.bist_gemm_pair(C.byref(ga), C.byref(gb), _stream()), "bist_gemm_pair")


COLSUM_QUEUE = None       # trainer: list of (dz, acc32 view, M, N) bias-gradient jobs, flushed by col_sum_flush()


def col_sum_flush() -> None:
    """Run every queued bias gradient (autograd.LinearFn) in batched launches on the current stream."""
    global COLSUM_QUEUE
    q = COLSUM_QUEUE
    if not q:
        return
    by_dtype = {}
    for dz, acc, M, N in q:
        by_dtype.setdefault(dz.dtype, []).append((dz, acc, M, N))
    for dt, jobs in by_dtype.items():
        arr = (BistColSum * len(jobs))()
        for i, (dz, acc, M, N) in enumerate(jobs):
            arr[i].x, arr[i].out, arr[i].M, arr[i].N, arr[i].ldx = dz.data_ptr(), acc.data_ptr(), M, N, N
        check(lib.bist_col_sum_multi(arr, len(jobs), dtype_code(dt), _stream()), "bist_col_sum_multi")
    q.clear()


LNGRAD_QUEUE = None       # trainer: list of (dy, x, gain, da, db, eps) LayerNorm parameter-gradient jobs, see lngrad_flush()


def lngrad_flush() -> None:
    """Sum the queued LayerNorm gain/offset gradients (autograd._ln_backward) in batched launches."""
    q = LNGRAD_QUEUE
    if not q:
        return
    groups = {}
    for job in q:
        groups.setdefault((job[0].dtype, job[1].shape[1]), []).append(job)
    for (dt, d), jobs in groups.items():
        arr = (BistLnGrad * len(jobs))()
        for i, (dy, x, a, da, db, eps) in enumerate(jobs):
            arr[i].dy, arr[i].x, arr[i].a, arr[i].da, arr[i].db = dy.data_ptr(), x.data_ptr(), a.data_ptr(), da.data_ptr(), db.data_ptr()
            arr[i].rows, arr[i].lddy, arr[i].ldx, arr[i].eps = x.shape[0], dy.stride(0), x.stride(0), eps
        check(lib.bist_layernorm_param_grad_multi(arr, len(jobs), d, dtype_code(dt), _stream()), "bist_layernorm_param_grad_multi")
    q.clear()


GEMM_TIMING = None        # set to a list to collect ((M,N,K,batch), start_event, end_event) per GEMM launch
GEMM_TIMING_SHAPE = None  # optional (M,N,K): only launches of this shape are bracketed


def linear(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, *, act: int = ACT_NONE, residual: Optional[Tensor] = None,
           res_map: Tuple[int, int] = (0, 0), alpha: float = 1.0, out: Optional[Tensor] = None,
           out_dtype: Optional[torch.dtype] = None, accumulate: bool = False, drop_p: float = 0.0, drop_seed: int = 0,
           ln=None, ln_out=None) -> Tensor:
    """y = act(alpha * x.W^T + bias) (+ residual): nn.Linear on the MFMA GEMM.
    ``ln_out`` = (gain, offset, eps): y = LayerNorm(act(alpha x.W^T + bias)) over the N = 512 output columns in the same launch (the
    LayerNorm EPILOGUE of bist_gemm, ln_mode = 1: VidEncoder8's in_norm(relu(W fts))) when inside its envelope, else a LayerNorm launch
    after the product.
    ``ln`` (or the tag ``x._bist_ln``): x is a PENDING LayerNorm output (layernorm(lazy=True)): the product is taken of the
    un-normalised rows with the LayerNorm as the GEMM's prologue, which also fills x.

    x [..., K] with a contiguous last dim and uniform row stride; W [N, K] (rows may be strided).
    ``accumulate`` adds into ``out`` (residual = out).  ``res_map=(outer, inner)`` reads residual
    row (m // outer) * inner + m % inner.
    """
    K = x.shape[-1]
    if ln is None:
        ln = getattr(x, "_bist_ln", None)
        if ln is not None:
            x._bist_ln = None
    x2 = x.reshape(-1, K)
    M, N = x2.shape[0], w.shape[0]
    if x2.stride(1) != 1 or w.stride(1) != 1:
        raise ValueError("bist_amd.linear: last dims must be contiguous")
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=out_dtype or x.dtype)
    o2 = out.reshape(-1, N) if out.dim() != 2 else out
    if accumulate:
        residual = o2
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, N) if residual.dim() != 2 else residual
    if ln_out is not None:
        ga, gb, eps = ln_out
        g = gemm_desc(x2, w, o2, M=M, N=N, K=K, a_rs=x2.stride(0), b_rs=w.stride(0), ldc=o2.stride(0), bias=bias, alpha=alpha, act=act)
        g.ln_gain, g.ln_offset, g.ln_eps, g.ln_mode = ga.data_ptr(), gb.data_ptr(), eps, 1
        if residual is None and drop_p == 0 and LN_EPILOGUE and lib.bist_gemm_ln_ok(C.byref(g)):
            check(lib.bist_gemm(C.byref(g), _stream()), "bist_gemm")
            return out
        gemm(x2, w, o2, M=M, N=N, K=K, a_rs=x2.stride(0), b_rs=w.stride(0), ldc=o2.stride(0), bias=bias,
             residual=r2, ldr=r2.stride(0) if r2 is not None else 0, alpha=alpha, act=act, res_map=res_map, drop_p=drop_p, drop_seed=drop_seed)
        return layernorm(o2, ga, gb, eps, out=o2)
    if ln is not None:
        xp, ga, gb, eps = ln
        g = gemm_desc(xp, w, o2, M=M, N=N, K=K, a_rs=xp.stride(0), b_rs=w.stride(0), ldc=o2.stride(0), bias=bias,
                      residual=r2, ldr=r2.stride(0) if r2 is not None else 0, alpha=alpha, act=act, res_map=res_map,
                      drop_p=drop_p, drop_seed=drop_seed)
        g.ln_gain, g.ln_offset, g.ln_out, g.ln_ld, g.ln_eps = ga.data_ptr(), gb.data_ptr(), x2.data_ptr(), x2.stride(0), eps
        if lib.bist_gemm_ln_ok(C.byref(g)):
            check(lib.bist_gemm(C.byref(g), _stream()), "bist_gemm")
            return out
        layernorm(xp, ga, gb, eps, out=x2)           # outside the prologue's envelope: the LayerNorm as a launch of its own
    gemm(x2, w, o2, M=M, N=N, K=K, a_rs=x2.stride(0), b_rs=w.stride(0), ldc=o2.stride(0), bias=bias,
         residual=r2, ldr=r2.stride(0) if r2 is not None else 0, alpha=alpha, act=act, res_map=res_map,
         drop_p=drop_p, drop_seed=drop_seed)
    return out


# LayerNorm as the prologue of the projection that consumes it (bist_gemm's LayerNorm prologue): 84 fewer launches per training step,
# measured 11.55 / 11.72 vs 11.67 / 11.73 ms per step and 0.297 / 0.305 vs 0.290 / 0.287 ms for the inference region at B = 16 -- the
# fused kernel waits for all of its K tiles before it can normalise (no product under the loads) and that costs what the saved launch
# boundary (~1.2 us) gains.  Not adopted: opt-in with BIST_LAZY_LN=1.
LAZY_LN = os.environ.get("BIST_LAZY_LN", "0") != "0"
# LayerNorm as the EPILOGUE of the input projection (bist_gemm ln_mode = 1: the two column-tile workgroups of a row block exchange row
# statistics): saves the 227 MB LayerNorm pass of the B = 64 region but measured 0.766 / 0.764 ms against 0.754 / 0.748 ms with the pass
# as its own launch (B = 16: 0.285 / 0.283 vs 0.289 / 0.273) -- the pair hand-shake and the epilogue arithmetic at the end of every
# round cost more than the streaming pass.  Opt-in with BIST_LN_EPILOGUE=1.
LN_EPILOGUE = os.environ.get("BIST_LN_EPILOGUE", "0") != "0"


def ln_lazy_ok(x: Tensor, a: Tensor, b: Tensor) -> bool:
    """May LN(x) be left to the projection that consumes it (the LayerNorm prologue of bist_gemm)?  Shape / layout side of the
    envelope; the projection asks bist_gemm_ln_ok with its own operands and runs the LayerNorm itself otherwise."""
    d = x.shape[-1]
    return (LAZY_LN and x.is_cuda and x.dtype == torch.bfloat16 and d == 512 and x.stride(-1) == 1 and x.is_contiguous()
            and x.data_ptr() % 16 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and a.dtype == x.dtype and b.dtype == x.dtype)


def ensure_ln(t: Tensor) -> Tensor:
    """Materialise a pending LayerNorm output (a tensor tagged ``_bist_ln`` by layernorm(lazy=True)) for a consumer that is not
    a fusable projection."""
    tag = getattr(t, "_bist_ln", None)
    if tag is not None:
        t._bist_ln = None
        x2, a, b, eps = tag
        layernorm(x2, a, b, eps, out=t)
    return t


def layernorm(x: Tensor, a: Tensor, b: Tensor, eps: float = 1e-6, out: Optional[Tensor] = None, lazy: bool = False) -> Tensor:
    """The reference's LayerNorm (modules.py:28-31): unbiased std, eps outside the sqrt.
    lazy (caller checked ln_lazy_ok and hands the result straight to ONE ``linear``): no launch -- the returned tensor is
    uninitialised and tagged ``_bist_ln = (x rows, gain, offset, eps)``; the projection normalises the rows in its prologue and
    fills the tensor (bist_gemm's LayerNorm prologue), or runs this LayerNorm first when it is outside that envelope."""
    _dev(x, a, b)
    if lazy and out is None:
        out = torch.empty(x.shape, device=x.device, dtype=x.dtype)
        out._bist_ln = (x.reshape(-1, x.shape[-1]), a, b, eps)
        return out
    d = x.shape[-1]
    x2 = x.reshape(-1, d)
    if x2.stride(1) != 1:
        raise ValueError("bist_amd.layernorm: last dim must be contiguous")
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    o2 = out.reshape(-1, d)
    check(lib.bist_layernorm_fwd(x2.data_ptr(), a.data_ptr(), b.data_ptr(), o2.data_ptr(), x2.shape[0], d,
                                 x2.stride(0), o2.stride(0), eps, dtype_code(x.dtype), _stream()), "bist_layernorm_fwd")
    return out


def layernorm_multi(xs: Sequence[Tensor], gains: Sequence[Tensor], offsets: Sequence[Tensor], outs: Sequence[Tensor], eps: float = 1e-6) -> None:
    """len(xs) LayerNorms of one geometry in ONE launch (bist_layernorm_fwd_multi): outs[i] = LN(xs[i]; gains[i], offsets[i]); every
    xs[i] / outs[i] is [rows, d] with the same rows, d and row strides (the t2s and s2t instances of a sublayer's norm)."""
    _dev(*xs, *gains, *offsets, *outs)
    n = len(xs)
    x0, o0 = xs[0], outs[0]
    rows, d = x0.shape
    for x, o in zip(xs, outs):
        if x.shape != (rows, d) or o.shape != (rows, d) or x.stride() != x0.stride() or o.stride() != o0.stride() or x.stride(1) != 1 or o.stride(1) != 1 \
                or x.dtype != x0.dtype or o.dtype != x0.dtype:
            raise ValueError("bist_amd.layernorm_multi: the sets must share one geometry and dtype")
    arr = (BistLnSet * n)()
    for i in range(n):
        arr[i].x, arr[i].a, arr[i].b, arr[i].y = xs[i].data_ptr(), gains[i].data_ptr(), offsets[i].data_ptr(), outs[i].data_ptr()
    check(lib.bist_layernorm_fwd_multi(arr, n, rows, d, x0.stride(0), o0.stride(0), eps, dtype_code(x0.dtype), _stream()), "bist_layernorm_fwd_multi")


def layernorm_bwd_multi(sets, rows: int, d: int, lddy: int, ldx: int, lddx: int, eps: float, ldadd: int, zdrop, dtype: torch.dtype) -> None:
    """sets: list of (dy, x, a, dx, da or None, db or None, dx_add or None, dz or None, drop_row0) -- bist_layernorm_bwd_multi."""
    n = len(sets)
    arr = (BistLnBwdSet * n)()
    for i, (dy, x, a, dx, da, db, add, dz, row0) in enumerate(sets):
        _dev(dy, x, a, dx, da, db, add, dz)
        arr[i].dy, arr[i].x, arr[i].a, arr[i].dx = dy.data_ptr(), x.data_ptr(), a.data_ptr(), dx.data_ptr()
        arr[i].da, arr[i].db, arr[i].dx_add, arr[i].dz, arr[i].drop_row0 = _ptr(da), _ptr(db), _ptr(add), _ptr(dz), int(row0)
    check(lib.bist_layernorm_bwd_multi(arr, n, rows, d, lddy, ldx, lddx, eps, ldadd, zdrop, dtype_code(dtype), _stream()), "bist_layernorm_bwd_multi")


def drop_ref(drop):
    """(p, seed) or None -> BistDrop by reference (NULL when off); the device step counter rides along."""
    if drop is None or drop[0] <= 0.0:
        return None
    return C.byref(BistDrop(float(drop[0]), int(drop[1]) & 0xFFFFFFFFFFFFFFFF, _ptr(DROP_CTR)))


def mha_core(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor], h: int, *, want_p: bool = False,
             out: Optional[Tensor] = None, drop=None) -> Tuple[Tensor, Optional[Tensor]]:
    """softmax(QK^T/sqrt(dk), masked -1e9) V per head; q [N,Lq,d], k/v [N,Lk,d] (row-strided views ok).

    mask: bool/uint8 [N,1,Lk], [N,Lq,Lk], [1,Lq,Lk] or None (modules.py:59-60 semantics).
    """
    _dev(q, k, v, mask)
    N, Lq, d = q.shape
    Lk = k.shape[1]
    dk = d // h
    for t in (q, k, v):
        if t.stride(2) != 1:
            raise ValueError("bist_amd.mha_core: last dim must be contiguous")
    if out is None:
        out = torch.empty((N, Lq, d), device=q.device, dtype=q.dtype)
    p = torch.empty((N, h, Lq, Lk), device=q.device, dtype=torch.float32) if want_p else None
    mptr, mbs, mqs = None, 0, 0
    if mask is not None:
        m = mask if mask.dtype == torch.uint8 else mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
        if m.stride(-1) != 1:
            m = m.contiguous()
        mbs = m.stride(0) if m.shape[0] > 1 else 0
        mqs = m.stride(1) if m.shape[1] > 1 else 0
        mptr = m.data_ptr()
        mask = m  # keep alive
    check(lib.bist_mha_core_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), mptr, out.data_ptr(), _ptr(p),
                                N, Lq, Lk, h, dk, q.stride(1), k.stride(1), v.stride(1), out.stride(1),
                                q.stride(0), k.stride(0), v.stride(0), out.stride(0), mbs, mqs,
                                1.0 / math.sqrt(dk), drop_ref(drop), dtype_code(q.dtype), _stream()), "bist_mha_core_fwd")
    return out, p


def st_stage1_pv(scores: Tensor, v: Tensor, tmask: Optional[Tensor], *, B: int, T: int, S: int, Lq: int, h: int,
                 dk: int, direction: int, out: Optional[Tensor] = None, drop=None) -> Tensor:
    """Stage-1 softmax + P.V of t2s (direction 0) / s2t (direction 1); see include/bist_hip.h."""
    _dev(scores, v, tmask)
    d = h * dk
    G = S if direction == 0 else T
    if not scores.is_contiguous() or scores.numel() != B * Lq * h * T * S:
        raise ValueError("bist_amd.st_stage1_pv: scores must be contiguous [B, Lq*h, T*S]")
    if v.stride(-1) != 1:
        raise ValueError("bist_amd.st_stage1_pv: V last dim must be contiguous")
    ldv = v.stride(-2)
    if out is None:
        out = torch.empty((B, G, Lq, d), device=v.device, dtype=v.dtype)
    mptr = None
    if tmask is not None:
        tmask = tmask.reshape(B, T if direction == 0 else S)        # one entry per key
        tmask = (tmask.view(torch.uint8) if tmask.dtype == torch.bool else tmask.to(torch.uint8)).contiguous()
        mptr = tmask.data_ptr()
    check(lib.bist_st_stage1_pv_fwd(scores.data_ptr(), v.data_ptr(), mptr, out.data_ptr(), B, T, S, Lq, h, dk, ldv,
                                    direction, drop_ref(drop), dtype_code(scores.dtype), dtype_code(v.dtype), _stream()),
          "bist_st_stage1_pv_fwd")
    return out


ST1F_TIMING = None        # set to a list to collect ((B,T,S,Lq,direction), start_event, end_event) per fused stage-1 launch


def st_stage1_fused_ok(T: int, S: int, Lq: int, d: int, h: int, direction: int, dtype: torch.dtype) -> bool:
    """True when bist_st_stage1_fused_fwd covers the shape (bf16, d = 512, h = 8, Lq <= 32, at most 128 keys)."""
    return dtype in _DT and bool(lib.bist_st_stage1_fused_ok(T, S, Lq, d, h, direction, _DT[dtype]))


WEIGHTS_EPOCH = 0      # bumped by whoever rewrites parameters behind autograd's back (Trainer.step: raw Adam kernel on flat views)


def weights_key(*params: Tensor):
    """Identity of a set of parameter VALUES for derived-operand caches (packed / re-ordered weights): storage, in-place
    version counter and the epoch of out-of-band updates."""
    return (WEIGHTS_EPOCH,) + tuple((p.data_ptr(), p._version) for p in params)


def stage_inputs(jobs) -> None:
    """bist_stage_inputs: jobs = [(src, dst, pad)]; src / dst tensors of one dtype whose shapes differ at most in the LAST dimension (dst the
    longer: its tail takes `pad`), both contiguous; one launch for all of them (at most 16 per launch)."""
    from ._lib import BistStageJob
    for k in range(0, len(jobs), 16):
        part = jobs[k:k + 16]
        arr = (BistStageJob * len(part))()
        for jb, (src, dst, pad) in zip(arr, part):
            _dev(src, dst)
            es = src.element_size()
            if (src.dtype != dst.dtype or not src.is_contiguous() or not dst.is_contiguous() or src.shape[:-1] != dst.shape[:-1]
                    or src.shape[-1] > dst.shape[-1] or src.numel() == 0):
                raise ValueError("bist_amd.stage_inputs: same dtype, contiguous, equal leading dimensions, destination rows at least as long")
            jb.src, jb.dst = src.data_ptr(), dst.data_ptr()
            jb.rows = src.numel() // src.shape[-1]
            jb.src_row_bytes, jb.dst_row_bytes = src.shape[-1] * es, dst.shape[-1] * es
            jb.pad, jb.pad_bytes = int(pad or 0) & ((1 << (8 * es)) - 1), es
        check(lib.bist_stage_inputs(arr, len(part), _stream()), "bist_stage_inputs")


def _children_signature(mods) -> int:
    return sum(id(c) for m in mods for c in m._modules.values()) + len(mods)


def _module_list(root):
    """The modules under `root`, cached on it; rebuilt when any of them gained, lost or REPLACED a child (the sum of the children's
    identities moves)."""
    hit = root.__dict__.get("_bist_module_list")
    if hit is not None and _children_signature(hit[0]) == hit[1]:
        return hit[0]
    mods = list(root.modules())
    root.__dict__["_bist_module_list"] = (mods, _children_signature(mods))
    return mods


def module_parameters(root) -> list:
    """The Parameter objects under `root` as they are NOW (a replaced Parameter is seen: the modules' dicts are read at every call), without
    nn.Module.parameters()' recursive generators -- 0.3 instead of 1.7 ms for the whole model, paid at the head of every decode turn."""
    return [p for m in _module_list(root) for p in m._parameters.values() if p is not None]


def pack_frag_rows(w: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """[rows, cols] bf16 weight in MFMA-fragment order (bist_pack_frag_rows); `out` keeps a cached copy's address stable."""
    _dev(w)
    if not w.is_contiguous() or w.dim() != 2:
        raise ValueError("bist_amd.pack_frag_rows: contiguous [rows, cols] expected")
    if out is None:
        out = torch.empty_like(w)
    check(lib.bist_pack_frag_rows(w.data_ptr(), out.data_ptr(), w.shape[0], w.shape[1], dtype_code(w.dtype), _stream()),
          "bist_pack_frag_rows")
    return out


def pack_frag_rows_multi(ws: Sequence[Tensor], outs: Sequence[Tensor]) -> None:
    """pack_frag_rows of up to 32 same-sized weights in one launch (bist_pack_frag_rows_multi); outs keep their addresses."""
    _dev(*ws, *outs)
    n = len(ws)
    for w, o in zip(ws, outs):
        if w.shape != ws[0].shape or o.shape != w.shape or not w.is_contiguous() or not o.is_contiguous() or w.dtype != ws[0].dtype or o.dtype != w.dtype:
            raise ValueError("bist_amd.pack_frag_rows_multi: contiguous matrices of one shape and dtype")
    for o0 in range(0, n, 32):
        part_w, part_o = ws[o0:o0 + 32], outs[o0:o0 + 32]
        a = (C.c_void_p * len(part_w))(*[w.data_ptr() for w in part_w])
        b = (C.c_void_p * len(part_w))(*[o.data_ptr() for o in part_o])
        check(lib.bist_pack_frag_rows_multi(a, b, len(part_w), ws[0].shape[0], ws[0].shape[1], dtype_code(ws[0].dtype), _stream()),
              "bist_pack_frag_rows_multi")


def st_stage1_fused_train_ok(T: int, S: int, Lq: int, d: int, h: int, direction: int, dtype: torch.dtype) -> bool:
    return dtype in _DT and bool(lib.bist_st_stage1_fused_train_ok(T, S, Lq, d, h, direction, _DT[dtype]))


def st_stage1_fused_train(qf: Tensor, vft: Tensor, kmask: Optional[Tensor], wv: Tensor, bv: Tensor, wo: Tensor, bo: Tensor, xres: Tensor, *,
                          h: int, direction: int, attn_drop=None, sub_drop=None, want_v: bool = True):
    """Stage 1 of one direction in one launch, TRAINING form (bist_st_stage1_fused_train_fwd): -> (Y [B,G,Lq,d], V [B,T,S,d],
    P [B,G,h,Lq,KP] f32 probabilities before dropout, O [B,G,Lq,d] context rows); wv / wo in fragment order; kmask uint8 [B,K] or None."""
    _dev(qf, vft, kmask, wv, bv, wo, bo, xres)
    B, T, S, d = vft.shape
    Lq = xres.shape[1]
    G, K = (S, T) if direction == 0 else (T, S)
    for t_ in (qf, vft, wv, bv, wo, bo, xres):
        if not t_.is_contiguous() or t_.dtype != vft.dtype:
            raise ValueError("bist_amd.st_stage1_fused_train: operands must be contiguous and of one dtype")
    if qf.numel() != B * Lq * h * d or wv.shape != (d, d) or wo.shape != (d, d) or xres.shape != (B, Lq, d):
        raise ValueError("bist_amd.st_stage1_fused_train: operand shapes do not match [B,T,S,d] / Lq / h")
    KP = (K + 3) // 4 * 4
    y = torch.empty((B, G, Lq, d), device=vft.device, dtype=vft.dtype)
    v = torch.empty((B, T, S, d), device=vft.device, dtype=vft.dtype) if want_v else None
    p = torch.empty((B, G, h, Lq, KP), device=vft.device, dtype=torch.float32)
    o = torch.empty((B, G, Lq, d), device=vft.device, dtype=vft.dtype)
    check(lib.bist_st_stage1_fused_train_fwd(qf.data_ptr(), vft.data_ptr(), _ptr(kmask), wv.data_ptr(), bv.data_ptr(), wo.data_ptr(), bo.data_ptr(),
                                             xres.data_ptr(), y.data_ptr(), _ptr(v), p.data_ptr(), o.data_ptr(), drop_ref(attn_drop),
                                             drop_ref(sub_drop), B, T, S, Lq, d, h, direction, dtype_code(vft.dtype), _stream()),
          "bist_st_stage1_fused_train_fwd")
    return y, v, p, o


def st_stage1_fused(qf: Tensor, vft: Tensor, kmask: Optional[Tensor], wv: Tensor, bv: Tensor, wo: Tensor, bo: Tensor,
                    xres: Tensor, *, h: int, direction: int, out: Optional[Tensor] = None) -> Tensor:
    """Stage 1 of one direction in one launch (inference form): qf [B, Lq*h, d] folded query, vft [B,T,S,d], kmask [B,K] or
    None, wv / wo in fragment order (pack_frag_rows), xres [B,Lq,d] -> Y [B,G,Lq,d]; see include/bist_hip.h."""
    _dev(qf, vft, kmask, wv, bv, wo, bo, xres)
    B, T, S, d = vft.shape
    Lq = xres.shape[1]
    G, K = (S, T) if direction == 0 else (T, S)
    for t_ in (qf, vft, wv, bv, wo, bo, xres):
        if not t_.is_contiguous() or t_.dtype != vft.dtype:
            raise ValueError("bist_amd.st_stage1_fused: operands must be contiguous and of one dtype")
    if qf.numel() != B * Lq * h * d or wv.shape != (d, d) or wo.shape != (d, d) or xres.shape != (B, Lq, d):
        raise ValueError("bist_amd.st_stage1_fused: operand shapes do not match [B,T,S,d] / Lq / h")
    if out is None:
        out = torch.empty((B, G, Lq, d), device=vft.device, dtype=vft.dtype)
    mptr = None
    if kmask is not None:
        kmask = kmask.reshape(B, K)
        kmask = (kmask.view(torch.uint8) if kmask.dtype == torch.bool else kmask.to(torch.uint8)).contiguous()
        mptr = kmask.data_ptr()
    timed = ST1F_TIMING is not None
    if timed:       # bench.py: HIP events around the launch on the launch stream (see gemm())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200_000)
        e0.record()
    check(lib.bist_st_stage1_fused_fwd(qf.data_ptr(), vft.data_ptr(), mptr, wv.data_ptr(), bv.data_ptr(), wo.data_ptr(),
                                       bo.data_ptr(), xres.data_ptr(), out.data_ptr(), B, T, S, Lq, d, h, direction,
                                       dtype_code(vft.dtype), _stream()), "bist_st_stage1_fused_fwd")
    if timed:
        e1.record()
        ST1F_TIMING.append(((B, T, S, Lq, direction), e0, e1))
    return out


def decoder_stack_ok(R: int, d: int, h: int, Lk_max: int, dtype: torch.dtype) -> bool:
    return dtype in _DT and bool(lib.bist_decoder_stack_ok(R, d, h, Lk_max, _DT[dtype]))


def decoder_stack(desc: Tensor, n_layers: int, x_in: Tensor, bufs: dict, self_mask: Tensor, R: int, LkS: int, slot0: int = 0,
                  head_local: bool = False, lk_pad_max: int = 64) -> Tensor:
    """All decoder layers of one decode step in one persistent launch (bist_decoder_stack_fwd); desc: device bytes of n_layers
    BistDecLayer; bufs: the caller-owned scratch and the per-layer self-attention caches "kc" / "vc" [n_layers, 64, 512] (zero-initialised
    once); the R rows of x_in take the cache slots slot0 ..; returns the [R, 512] rows of the output buffer.  head_local: hand the kernel
    the partial buffer bufs["p"], which selects its head-local form for R <= 16 (measured slower; see decstack.hip).  lk_pad_max: the
    largest padded memory length in desc (above 64: the kernel instance with the chunked attention core)."""
    _dev(desc, x_in, self_mask)
    if bufs["kc"].shape[0] < n_layers or tuple(self_mask.shape) != (R, LkS) or self_mask.dtype != torch.uint8 or not self_mask.is_contiguous():
        raise ValueError("bist_amd.decoder_stack: caches / mask do not fit the call")
    check(lib.bist_decoder_stack_fwd(desc.data_ptr(), n_layers, x_in.data_ptr(), bufs["x0"].data_ptr(), bufs["x1"].data_ptr(),
                                     bufs["q"].data_ptr(), bufs["kc"].data_ptr(), bufs["vc"].data_ptr(), bufs["h"].data_ptr(),
                                     self_mask.data_ptr(), R, LkS, slot0, lk_pad_max, bufs["sync"].data_ptr(), _ptr(bufs.get("p") if head_local else None), dtype_code(x_in.dtype), _stream()),
          "bist_decoder_stack_fwd")
    return bufs["x0" if (5 * n_layers - 1) % 2 == 0 else "x1"][:R]       # the residual stream ping-pongs: write k lands in buffer (k - 1) % 2


def st_stage2(q2f: Tensor, y: Tensor, gmask: Optional[Tensor], *, h: int, out: Optional[Tensor] = None, drop=None,
              want_rowsum: bool = False):
    """Stage-2 attention over the stage-1 outputs with K/V folded out; q2f [B,Lq,h,d], y [B,G,Lq,d].
    With dropout (or want_rowsum) returns (PY, rowsum f32 [B,Lq,h]): the value bias must be scaled by rowsum."""
    _dev(q2f, y, gmask)
    B, G, Lq, d = y.shape
    if not (q2f.is_contiguous() and y.is_contiguous()):
        raise ValueError("bist_amd.st_stage2: q2f and Y must be contiguous")
    if out is None:
        out = torch.empty((B, Lq, h, d), device=y.device, dtype=y.dtype)
    mptr = None
    if gmask is not None:
        gmask = gmask.reshape(B, G)
        gmask = (gmask.view(torch.uint8) if gmask.dtype == torch.bool else gmask.to(torch.uint8)).contiguous()
        mptr = gmask.data_ptr()
    dref = drop_ref(drop)
    rs = torch.empty((B, Lq, h), device=y.device, dtype=torch.float32) if (dref is not None or want_rowsum) else None
    check(lib.bist_st_stage2_fwd(q2f.data_ptr(), y.data_ptr(), mptr, out.data_ptr(), _ptr(rs), B, G, Lq, h, d, dref,
                                 dtype_code(y.dtype), _stream()), "bist_st_stage2_fwd")
    return out if rs is None else (out, rs)


def scaled_bias(x: Tensor, s: Tensor, bias: Tensor, h: int, out: Optional[Tensor] = None) -> Tensor:
    """x[m, hh*dk+c] + s[m, hh] * bias[hh*dk+c]: the value bias of stage 2 under attention dropout."""
    _dev(x, s, bias)
    d = x.shape[-1]
    x2 = x.reshape(-1, d)
    if not x2.is_contiguous() or not s.is_contiguous() or s.dtype != torch.float32 or s.numel() != x2.shape[0] * h:
        raise ValueError("bist_amd.scaled_bias: bad operands")
    if out is None:
        out = torch.empty_like(x2)
    check(lib.bist_scaled_bias_fwd(x2.data_ptr(), s.data_ptr(), bias.data_ptr(), out.data_ptr(), x2.shape[0], h, d // h,
                                   dtype_code(x.dtype), _stream()), "bist_scaled_bias_fwd")
    return out.view(x.shape)


def scaled_bias_z(x: Tensor, s: Tensor, bias0: Tensor, bias_zs: int, h: int, nsets: int, out: Optional[Tensor] = None) -> Tensor:
    """scaled_bias over `nsets` stacked row blocks, block z with the bias at bias0 + z * bias_zs elements (bist_scaled_bias_fwd_z)."""
    _dev(x, s, bias0)
    d = x.shape[-1]
    x2 = x.reshape(-1, d)
    if not x2.is_contiguous() or not s.is_contiguous() or s.dtype != torch.float32 or s.numel() != x2.shape[0] * h or x2.shape[0] % nsets:
        raise ValueError("bist_amd.scaled_bias_z: bad operands")
    if out is None:
        out = torch.empty_like(x2)
    check(lib.bist_scaled_bias_fwd_z(x2.data_ptr(), s.data_ptr(), bias0.data_ptr(), out.data_ptr(), x2.shape[0], h, d // h, nsets, bias_zs,
                                     dtype_code(x.dtype), _stream()), "bist_scaled_bias_fwd_z")
    return out.view(x.shape)


def copy_into(dst: Tensor, src: Tensor) -> Tensor:
    """dst <- src (same shape, dtype, both contiguous) as a one-operand bist_add_n launch: device-side data movement only."""
    _dev(dst, src)
    if dst.shape != src.shape or dst.dtype != src.dtype or not dst.is_contiguous() or not src.is_contiguous():
        raise ValueError("bist_amd.copy_into: contiguous tensors of one shape and dtype")
    arr = (C.c_void_p * 1)(src.data_ptr())
    check(lib.bist_add_n(arr, 1, dst.data_ptr(), dst.numel(), dtype_code(dst.dtype), _stream()), "bist_add_n")
    return dst


def embed_pe(ids: Tensor, lut: Tensor, pe: Tensor, out: Optional[Tensor] = None, drop=None) -> Tensor:
    """lut[ids]*sqrt(d) + pe[:L]  (modules.py:121-123,141-144); ids int64 [B,L], pe f32 [>=L,d]."""
    _dev(ids, lut, pe)
    B, L = ids.shape
    d = lut.shape[1]
    if ids.dtype != torch.int64 or pe.dtype != torch.float32 or pe.shape[0] < L or pe.shape[1] != d:
        raise ValueError("bist_amd.embed_pe: bad ids/pe")
    ids = ids.contiguous()
    if out is None:
        out = torch.empty((B, L, d), device=lut.device, dtype=lut.dtype)
    check(lib.bist_embed_pe_fwd(ids.data_ptr(), lut.data_ptr(), pe.data_ptr(), out.data_ptr(), B * L, L, d, drop_ref(drop),
                                dtype_code(lut.dtype), _stream()), "bist_embed_pe_fwd")
    return out


def text_vector(p: Tensor, enc: Tensor) -> Tensor:
    """sum_t p[b,i,t] * enc[b,t,:] (generator.py:117-118), inference: p f32 [B,Lt,L], enc [B,L,d] -> [B,Lt,d] in enc's dtype."""
    _dev(p, enc)
    B, Lt, L = p.shape
    d = enc.shape[-1]
    if p.dtype != torch.float32 or enc.shape[0] != B or enc.shape[1] != L or not p.is_contiguous() or not enc.is_contiguous():
        raise ValueError("bist_amd.text_vector: bad operands")
    out = torch.empty((B, Lt, d), device=enc.device, dtype=enc.dtype)
    check(lib.bist_text_vector_fwd(p.data_ptr(), enc.data_ptr(), out.data_ptr(), B * Lt, Lt, L, d, dtype_code(enc.dtype), _stream()),
          "bist_text_vector_fwd")
    return out


def temporal_mask(fts: Tensor) -> Tensor:
    """(fts.sum(2).sum(-1) != 0).unsqueeze(-2) of data/dataset.py:79, computed on the device."""
    _dev(fts)
    B, T = fts.shape[0], fts.shape[1]
    f = fts.contiguous()
    m = torch.empty((B, 1, T), device=fts.device, dtype=torch.uint8)
    check(lib.bist_temporal_mask(f.data_ptr(), m.data_ptr(), B * T, f.numel() // (B * T), dtype_code(f.dtype), _stream()),
          "bist_temporal_mask")
    return m.view(torch.bool)


def fuse_modalities(score: Tensor, xs: Sequence[Tensor], out: Optional[Tensor] = None) -> Tensor:
    """sum_j softmax(score)[..., j] * xs[j]   (decoder.py:155-159); score [..., n], xs[j] [..., d]."""
    _dev(score, *xs)
    n, d = score.shape[-1], xs[0].shape[-1]
    xs = [x.contiguous() for x in xs]
    score = score.contiguous()
    rows = xs[0].numel() // d
    if out is None:
        out = torch.empty_like(xs[0])
    arr = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    check(lib.bist_fuse_modalities(score.data_ptr(), arr, out.data_ptr(), rows, n, d, dtype_code(score.dtype), _stream()),
          "bist_fuse_modalities")
    return out


def cast(x: Tensor, dtype: torch.dtype) -> Tensor:
    _dev(x)
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    check(lib.bist_cast(x.data_ptr(), out.data_ptr(), x.numel(), dtype_code(x.dtype), dtype_code(dtype), _stream()), "bist_cast")
    return out


def permute_ts(x: Tensor) -> Tensor:
    """[B,T,S,d] -> [B,S,T,d] (contiguous copy, bist_permute_ts)."""
    _dev(x)
    x = x.contiguous()
    B, T, S, d = x.shape
    y = torch.empty((B, S, T, d), device=x.device, dtype=x.dtype)
    check(lib.bist_permute_ts(x.data_ptr(), y.data_ptr(), B, T, S, d, dtype_code(x.dtype), _stream()), "bist_permute_ts")
    return y


ADD_N_MAX = 24


def add_n(xs, out: Optional[Tensor] = None) -> Tensor:
    """sum of same-shape contiguous tensors in one pass (bist_add_n); more than ADD_N_MAX terms go in rounds."""
    xs = [x.contiguous() for x in xs]
    _dev(*xs)
    if out is None:
        out = torch.empty_like(xs[0])
    first = True
    for o in range(0, len(xs), ADD_N_MAX - 1):
        part = xs[o:o + ADD_N_MAX - 1] if first else [out] + xs[o:o + ADD_N_MAX - 1]
        arr = (C.c_void_p * len(part))(*[t.data_ptr() for t in part])
        check(lib.bist_add_n(arr, len(part), out.data_ptr(), out.numel(), dtype_code(out.dtype), _stream()), "bist_add_n")
        first = False
    return out


def add(a: Tensor, b: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """a + b (b broadcast over leading dims when smaller) -- SublayerConnection's residual add."""
    _dev(a, b)
    if a.dtype != b.dtype:
        raise TypeError("bist_amd.add: dtype mismatch")
    a, b = a.contiguous(), b.contiguous()
    if a.numel() % b.numel() != 0:
        raise ValueError("bist_amd.add: b must tile a")
    if out is None:
        out = torch.empty_like(a)
    check(lib.bist_add_bcast(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), b.numel(), dtype_code(a.dtype), _stream()),
          "bist_add_bcast")
    return out


def add_dropout(a: Tensor, b: Tensor, drop, mode: int) -> Tensor:
    """mode 0: a + dropout(b); mode 1: dropout(a + b); b may be a broadcast operand whose numel divides a's (see bist_add_dropout_fwd)."""
    _dev(a, b)
    if not (a.is_contiguous() and b.is_contiguous()) or a.dtype != b.dtype or a.numel() % b.numel():
        raise ValueError("bist_amd.add_dropout: contiguous operands of one dtype, b broadcastable by period")
    out = torch.empty_like(a)
    check(lib.bist_add_dropout_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), b.numel(), mode, drop_ref(drop), dtype_code(a.dtype),
                                   _stream()), "bist_add_dropout_fwd")
    return out


def dropout_mask_grad(dy: Tensor, drop) -> Tensor:
    """dy * mask / (1 - p) with the mask of add_dropout (element index order of the contiguous tensor)."""
    _dev(dy)
    dy = dy.contiguous()
    dz = torch.empty_like(dy)
    N = dy.shape[-1]
    M = dy.numel() // N
    check(lib.bist_epilogue_bwd(dy.data_ptr(), None, dz.data_ptr(), M, N, N, N, N, ACT_NONE, float(drop[0]), int(drop[1]) & 0xFFFFFFFFFFFFFFFF,
                                _ptr(DROP_CTR), dtype_code(dy.dtype), _stream()), "bist_epilogue_bwd")
    return dz


def pointer_mix(logits: Tensor, switch_logits: Tensor, ptr_p: Sequence[Tensor], ptr_text: Sequence[Tensor], Lt: int,
                sigmoid_switch: bool = False) -> Tensor:
    """log of the pointer/vocabulary mixture (generator.py:84-127); logits f32 [rows,V] -> f32 [rows,V]."""
    _dev(logits, switch_logits, *ptr_p, *ptr_text)
    rows, V = logits.shape
    n = len(ptr_p)
    if logits.dtype != torch.float32 or switch_logits.dtype != torch.float32:
        raise TypeError("bist_amd.pointer_mix: logits and switch logits must be float32")
    logits, switch_logits = logits.contiguous(), switch_logits.contiguous()
    ps = [p.reshape(rows, -1).contiguous() for p in ptr_p]
    ts = [t.contiguous() for t in ptr_text]
    for p, t in zip(ps, ts):
        if p.dtype != torch.float32 or t.dtype != torch.int64 or p.shape[1] != t.shape[1]:
            raise TypeError("bist_amd.pointer_mix: pointer probabilities f32 [rows,L], text int64 [B,L]")
    out = torch.empty((rows, V), device=logits.device, dtype=torch.float32)
    pp = (C.c_void_p * n)(*[p.data_ptr() for p in ps])
    tt = (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    ll = (C.c_int32 * n)(*[p.shape[1] for p in ps])
    check(lib.bist_pointer_mix_fwd(logits.data_ptr(), switch_logits.data_ptr(), n, pp, tt, ll, out.data_ptr(), rows, Lt, V,
                                   1 if sigmoid_switch else 0, _stream()), "bist_pointer_mix_fwd")
    return out


def pointer_decode_mix(x: Tensor, tgt: Tensor, logits: Tensor, srcs: Sequence[dict], wsw: Tensor, bsw: Tensor, scale: float) -> Tensor:
    """The pointer heads of one decode step for rows that share one dialogue, one launch (bist_pointer_decode_mix_fwd): x / tgt
    [rows,d], logits f32 [rows,V], srcs: per source the turn's constants {"M" f32 [L,d], "c" f32 [L], "mask" u8 [L], "E" f32 [L,n+1],
    "text" int64 [L], optional "p" f32 [rows,L] to receive the pointer probabilities}; wsw = pointer_gen_W.weight, bsw its bias."""
    rows, d = x.shape
    V = logits.shape[-1]
    n = len(srcs)
    _dev(x, tgt, logits, wsw, bsw)
    if logits.dtype != torch.float32 or not (x.is_contiguous() and tgt.is_contiguous() and logits.is_contiguous()) or tgt.shape != x.shape \
            or tgt.dtype != x.dtype or wsw.dtype != x.dtype or bsw.dtype != x.dtype or wsw.shape != (n + 1, (n + 2) * d) or wsw.stride(1) != 1:
        raise ValueError("bist_amd.pointer_decode_mix: bad operands")
    arr = (_lib.BistPtrDecSrc * n)()
    for j, sj in enumerate(srcs):
        L = sj["text"].shape[0]
        M, c, mask, E, text, pout = sj["M"], sj["c"], sj["mask"], sj["E"], sj["text"], sj.get("p")
        _dev(M, c, mask, E, text)
        if M.dtype != torch.float32 or tuple(M.shape) != (L, d) or c.dtype != torch.float32 or c.numel() != L or mask.dtype != torch.uint8 \
                or mask.numel() != L or E.dtype != torch.float32 or tuple(E.shape) != (L, n + 1) or text.dtype != torch.int64 \
                or not all(t.is_contiguous() for t in (M, c, mask, E, text)) or (pout is not None and (pout.dtype != torch.float32 or tuple(pout.shape) != (rows, L) or not pout.is_contiguous())):
            raise ValueError("bist_amd.pointer_decode_mix: bad constants of source %d" % j)
        arr[j].M, arr[j].c, arr[j].mask, arr[j].E, arr[j].text = M.data_ptr(), c.data_ptr(), mask.data_ptr(), E.data_ptr(), text.data_ptr()
        arr[j].p_out, arr[j].L = (pout.data_ptr() if pout is not None else None), L
    out = torch.empty((rows, V), device=x.device, dtype=torch.float32)
    check(lib.bist_pointer_decode_mix_fwd(x.data_ptr(), tgt.data_ptr(), logits.data_ptr(), arr, n, wsw.data_ptr(), wsw.stride(0), bsw.data_ptr(),
                                          float(scale), out.data_ptr(), rows, d, V, dtype_code(x.dtype), _stream()), "bist_pointer_decode_mix_fwd")
    return out


def log_softmax(x: Tensor) -> Tensor:
    _dev(x)
    if x.dtype != torch.float32:
        raise TypeError("bist_amd.log_softmax: float32 only")
    V = x.shape[-1]
    x2 = x.reshape(-1, V).contiguous()
    out = torch.empty_like(x2)
    check(lib.bist_log_softmax_fwd(x2.data_ptr(), out.data_ptr(), x2.shape[0], V, _stream()), "bist_log_softmax_fwd")
    return out.view(x.shape)


def label_smoothing_rows(logp: Tensor, target: Tensor, smoothing: float, pad: int) -> Tensor:
    """Per-row KLDiv(sum) against the smoothed one-hot (label_smoothing.py:20-30); logp f32 [rows,V]."""
    _dev(logp, target)
    rows, V = logp.shape
    logp, target = logp.contiguous(), target.contiguous()
    if logp.dtype != torch.float32 or target.dtype != torch.int64:
        raise TypeError("bist_amd.label_smoothing_rows: logp f32, target int64")
    out = torch.empty((rows,), device=logp.device, dtype=torch.float32)
    check(lib.bist_label_smoothing_fwd(logp.data_ptr(), target.data_ptr(), out.data_ptr(), rows, V, smoothing, pad, _stream()),
          "bist_label_smoothing_fwd")
    return out


def sum_div(x: Tensor, denom: Optional[Tensor] = None, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """sum(x) / denom (device int64 scalar) as a device f32 scalar, fixed summation order."""
    _dev(x, denom, out)
    x = x.contiguous()
    if x.dtype != torch.float32 or (denom is not None and denom.dtype != torch.int64):
        raise TypeError("bist_amd.sum_div: x f32, denom int64")
    if out is None:
        out = torch.empty((1,), device=x.device, dtype=torch.float32)
        accumulate = False
    check(lib.bist_sum_div(x.data_ptr(), x.numel(), _ptr(denom), out.data_ptr(), 1 if accumulate else 0, _stream()), "bist_sum_div")
    return out
