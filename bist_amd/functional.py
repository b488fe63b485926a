"""Composite steps of the hot path, expressed on the kernels of libbist_hip.so.

The algebra used here (checked against the oracle in tests/):
  * the key projection is folded into the query:  (Q_h W_k,h) X^T == Q_h (X W_k,h^T)^T and the key
    bias is constant along the softmax axis, so it cancels -- K is never materialised;
  * in stage 2 the value projection is applied AFTER the weighted sum:
    P (Y W_v^T + b_v) == (P Y) W_v^T + b_v because each row of P sums to one.
Every function takes and returns device tensors; nothing here touches the CPU.  With autograd
enabled each step is a torch.autograd.Function whose backward also runs on the HIP kernels
(bist_amd/autograd.py); with autograd disabled the kernels are called directly.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import torch

from . import autograd as ag
from . import ops
from .ops import ACT_NONE, ACT_RELU  # noqa: F401

Tensor = torch.Tensor

linear = ag.linear


def _grad() -> bool:
    return torch.is_grad_enabled()


def linear_pair(x1: Tensor, w1: Tensor, b1: Optional[Tensor], x2: Tensor, w2: Tensor, b2: Optional[Tensor]):
    """Two independent plain projections in one launch (see autograd.LinearPairFn / bist_gemm_pair)."""
    if _grad():
        y1, y2 = ag.LinearPairFn.apply(x1, w1, b1, x2, w2, b2)
    else:
        y1, y2 = ag.LinearPairFn.forward(_NoCtx(), x1, w1, b1, x2, w2, b2)
    return y1.view(*x1.shape[:-1], -1), y2.view(*x2.shape[:-1], -1)


def _tag_ln(y: Tensor, x: Tensor, a: Tensor, b: Tensor, eps: float) -> Tensor:
    y._bist_ln = (x.detach().reshape(-1, x.shape[-1]), a.detach(), b.detach(), eps)       # (set on the tensor the CALLER holds)
    return y


def layernorm(x: Tensor, a: Tensor, b: Tensor, eps: float = 1e-6, lazy: bool = False) -> Tensor:
    """lazy: the caller hands the result to exactly ONE ``linear`` and nothing else reads it -- the LayerNorm then runs as that
    projection's prologue (ops.layernorm / bist_gemm's LayerNorm prologue) instead of a launch of its own."""
    lazy = lazy and ops.ln_lazy_ok(x, a, b)
    if _grad():
        y = ag.LayerNormFn.apply(x, a, b, eps, getattr(x, "_bist_drop", None), lazy)
        return _tag_ln(y, x, a, b, eps) if lazy else y
    return ops.layernorm(x, a, b, eps, lazy=lazy)


def layernorm_res(x: Tensor, a: Tensor, b: Tensor, eps: float = 1e-6, lazy: bool = False):
    """(LN(x), x) for x + sublayer(LN(x)): hand the second value to the residual add, so that its gradient
    is folded into the LayerNorm backward kernel instead of a separate accumulation pass.  lazy: see layernorm."""
    if _grad() and x.requires_grad:
        lazy = lazy and ops.ln_lazy_ok(x, a, b)
        y, r = ag.LayerNormResFn.apply(x, a, b, eps, getattr(x, "_bist_drop", None), lazy)
        return (_tag_ln(y, x, a, b, eps) if lazy else y), r
    return layernorm(x, a, b, eps, lazy=lazy), x


def head_fold(q: Tensor, wk: Tensor, h: int, alpha: float) -> Tensor:
    """Qf[m, hh*d + n] = alpha * sum_c q[m, hh*dk + c] * wk[hh*dk + c, n]     q [M,d], wk [d,d] -> [M, h*d]
    (one batched GEMM over the heads; wk is read "NN": row stride 1, k stride d)."""
    if _grad():
        return ag.HeadFoldFn.apply(q, wk, h, alpha)
    return ag.HeadFoldFn.forward(_NoCtx(), q, wk, h, alpha)


def head_unfold(py: Tensor, wv: Tensor, bv: Tensor, h: int) -> Tensor:
    """O[m, hh*dk + c] = sum_n py[m, hh*d + n] * wv[hh*dk + c, n] + bv[hh*dk + c]    py [M,h*d] -> [M,d]."""
    if _grad():
        return ag.HeadUnfoldFn.apply(py, wv, bv, h)
    return ag.HeadUnfoldFn.forward(_NoCtx(), py, wv, bv, h)


def st_scores(qf: Tensor, vft: Tensor) -> Tensor:
    """scores[b, r, ts] = qf[b, r, :] . vft[b, ts, :]   qf [B,R,d], vft [B,TS,d] -> f32 [B,R,TS]."""
    if _grad():
        return ag.StScoresFn.apply(qf, vft)
    return ag.StScoresFn.forward(_NoCtx(), qf, vft)


def bmm_nn(a: Tensor, b: Tensor) -> Tensor:
    """C[z] = A[z] . B[z]  (A [Z,M,K], B [Z,K,N]) -- the pointer generator's text vector (generator.py:117-118)."""
    if _grad():
        return ag.BmmNNFn.apply(a, b)
    return ag.BmmNNFn.forward(_NoCtx(), a, b)


def mha_packed(a: Tensor, b: Optional[Tensor], c: Optional[Tensor], mode: str, mask: Optional[Tensor], h: int, want_p: bool = False,
               drop=None):
    """Attention core over packed projections (see autograd.MhaCoreFn); returns (ctx [N,Lq,d], P or None).
    drop = (p, seed): dropout of the probabilities (modules.py:62-63)."""
    if _grad():
        return ag.MhaCoreFn.apply(a, b, c, mode, mask, h, want_p, drop)
    q, k, v = ag.MhaCoreFn._views(a, b, c, mode)
    return ops.mha_core(q, k, v, mask, h, want_p=want_p, drop=drop)


def st_stage1_pv(scores, v, tmask, *, B, T, S, Lq, h, dk, direction, drop=None):
    if _grad():
        return ag.StStage1PvFn.apply(scores, v, tmask, (B, T, S, Lq, h, dk), direction, drop)
    return ops.st_stage1_pv(scores, v, tmask, B=B, T=T, S=S, Lq=Lq, h=h, dk=dk, direction=direction, drop=drop)


FUSED_TRAIN = os.environ.get("BIST_FUSED_TRAIN", "1") != "0"      # tuning aid: 0 = the training forward of stage 1 as four launches (value / score products, core, output projection)


# Who projects the values of the fused training launch: 0 = a product of its own per direction (t2s: ahead on the caption stream, s2t: on
# the main stream, as in the four-launch form); 1 = the launch itself, both directions (the value projection's two backward products then
# sit on each direction's critical chain: measured 13.0 vs 11.8 ms per step at BASELINE configs[1], although 1.5 ms of kernel time are
# gone); 2 = the launch itself for the s2t direction only -- that direction's value products leave the MAIN stream (which carries the
# longer t2s chain: 49 groups against 32) for the shorter s2t chain's own stream: measured 18.8 vs 19.7 ms per step at T = 128 but
# 10.02 vs 9.88 ms at T = 32, so -1 (default) = mode 2 from 64 frames, mode 0 below.
FUSED_TRAIN_OWN_V = int(os.environ.get("BIST_FUSED_TRAIN_OWN_V", "-1"))


def fused_train_own_v(T: int) -> int:
    return FUSED_TRAIN_OWN_V if FUSED_TRAIN_OWN_V >= 0 else (2 if T >= 64 else 0)


# tuning aid: 1 = the t2s stage-1 backward products that feed nothing on its chain (output projection's weight gradient, video gradient through
# the scores: 54 us per layer) go to the caption / decoder stream, their consumer waiting on an event.  Measured 11.57 vs 9.94 ms per step
# (T = 128: 20.9 vs 18.8): a fork per layer and six extra event waits cost far more than the launches they move -- every cross-stream
# dependency of the replayed hipGraph is paid on the critical path.  Off.
OFFLOAD_T2S = os.environ.get("BIST_OFFLOAD_T2S", "0") != "0"


def st_stage1_fused_train(qf, x, vft_a, vft_b, tmask, attn, frag, *, h, direction, attn_drop=None, sub_drop=None, v=None, offload=False):
    """Training stage 1 of one direction as one launch forward (autograd.St1FusedTrainFn); attn: the MultiHeadedAttention holding
    linears[2] (values) and linears[3] (output); frag = (W_v, W_o) in fragment order; v: the value projection when the caller runs it
    as a product of its own (vft_b is then unused).  -> (y [B,G,Lq,d], x' = x again for its next consumer)"""
    return ag.St1FusedTrainFn.apply(qf, x, vft_a, vft_b, v, tmask, attn.linears[2].weight, attn.linears[2].bias, attn.linears[3].weight,
                                    attn.linears[3].bias, frag[0], frag[1], (h, direction, attn_drop, sub_drop, bool(offload and OFFLOAD_T2S)))


def st_stage2(q2f, y, gmask, *, h, drop=None):
    """(PY, rowsum or None); rowsum only under dropout -- scale the value bias with it (scaled_bias)."""
    if _grad():
        return ag.StStage2Fn.apply(q2f, y, gmask, h, drop)
    res = ops.st_stage2(q2f, y, gmask, h=h, drop=drop)
    return res if isinstance(res, tuple) else (res, None)


def scaled_bias(x, s, bias, h):
    return ag.ScaledBiasFn.apply(x, s, bias, h) if _grad() else ops.scaled_bias(x, s, bias, h)


def attn_drop(module):
    """(p, seed) for the dropout of attention probabilities held by a MultiHeadedAttention (modules.py:79,95)."""
    p = float(getattr(getattr(module, "dropout", None), "p", 0.0))
    if module.training and p > 0.0:
        return (p, next_seed("attn", module))
    return None


def embed_pe(ids, lut, pe, drop=None):
    return ag.EmbedFn.apply(ids, lut, pe, drop) if _grad() else ops.embed_pe(ids, lut, pe, drop=drop)


def fuse_modalities(score, xs: Sequence[Tensor]):
    return ag.FuseFn.apply(score, *xs) if _grad() else ops.fuse_modalities(score, xs)


def fuse_dyn(W, bias, parts, xs_idx):
    """softmax(cat(parts) W^T + bias)-weighted sum of parts[xs_idx[j]] (decoder.py:142-159) as one autograd node (autograd.FuseDynFn)."""
    return ag.FuseDynFn.apply(W, bias, tuple(xs_idx), *parts)


def add(a, b):
    if _grad() and (a.requires_grad or b.requires_grad):
        return ag.AddFn.apply(a, b)
    return ops.add(a, b)


def add_dropout(a, b, drop, mode: int):
    """mode 0: a + dropout(b); mode 1: dropout(a + b) -- the stand-alone dropout sites of modules.py:44 / :144."""
    return ag.AddDropoutFn.apply(a, b, drop, mode) if _grad() else ops.add_dropout(a.contiguous(), b.contiguous(), drop, mode)


def permute_ts(x):
    """The video tensor in region-major order [B,S,T,d] (one copy per step, shared by all layers; see bist_permute_ts)."""
    return ag.PermuteTSFn.apply(x) if (_grad() and x.requires_grad) else ops.permute_ts(x)


# Tensors that cross from one stream's chain to another's in a training step (the decoder / caption layers on the caption stream read the
# encoded texts, the fused memories and the decoder input of the main stream and save them for their backward; the value projections
# issued ahead on the caption stream are consumed on the main one; every layer boundary hands tensors between the chains).  The caching
# allocator gives a block back to its HOME stream's pool as soon as the last host-side reference is dropped -- possibly long before the
# foreign stream's kernels have run: in a replayed graph the streams really run side by side.  Found when the main stream stopped
# waiting for the caption stream's last backward launches: a weight-gradient product there read a recycled `encoded_his`, and the
# value projections' outputs were recycled under the main stream's stage-1 nodes (each a full Adam step of difference in a few weight
# matrices between the replayed and the eager step).  Such tensors are kept alive here until the streams have been joined (Trainer:
# after the backward pass; release_taken() for other training loops).  (record_stream at the same places does as well.)
_TAKEN = []
KEEP_TAKEN = os.environ.get("BIST_KEEP_TAKEN", "1") != "0"


def _keep_taken(t):
    if KEEP_TAKEN and torch.is_tensor(t) and t.is_cuda and _grad():
        if len(_TAKEN) > 16384:              # a training loop that never joins: bounded
            del _TAKEN[:8192]
        _TAKEN.append(t)
    return t


def release_taken() -> None:
    """Drop the references to the cross-stream tensors of a step whose streams have been joined."""
    _TAKEN.clear()


class Fan:
    """Hands out the aliases of FanOutFn one by one; once they are used up (or without autograd) the tensor itself."""

    def __init__(self, x, n: int, join: bool = False):
        self.x = x
        self._it = iter(ag.FanOutFn.apply(x, n, join)) if (n > 1 and _grad() and x.requires_grad and x.is_cuda) else iter(())

    def take(self):
        return _keep_taken(next(self._it, self.x))


def fan_take(ft, key):
    """The next alias of ft[key] when the layer loop set a fan up for it (training: every consumer gets an alias of its own and the
    gradients are summed in ONE bist_add_n launch by FanOutFn instead of pairwise by autograd), else ft[key] itself."""
    fans = ft.get("_bist_fans")
    f = fans.get(key) if fans else None
    return f.take() if f is not None else _keep_taken(ft[key])


def fan_set(ft, key, n: int, join: bool = False):
    """ft[key] will be read by up to n consumers (unused aliases cost nothing).  join: made on the main stream with consumers on the side
    streams too -- the gradient sum then waits for every side stream (autograd.FanOutFn)."""
    if _grad() and ft.get(key) is not None and ft[key].requires_grad and ft[key].is_cuda:
        ft.setdefault("_bist_fans", {})[key] = Fan(ft[key], n, join)


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    return ag.CastFn.apply(x, dtype) if _grad() else ops.cast(x, dtype)


def pointer_mix(logits, sw, ps, texts, Lt, sigmoid_switch=False):
    if _grad():
        return ag.PointerMixFn.apply(logits, sw, Lt, sigmoid_switch, len(ps), *ps, *texts)
    return ops.pointer_mix(logits, sw, ps, texts, Lt, sigmoid_switch)


def sum_terms(terms) -> Tensor:
    """The sum of the loss terms (device scalars [1], f32) as one launch instead of a chain of additions."""
    terms = list(terms)
    if len(terms) == 1:
        return terms[0]
    if all(t.is_cuda and t.dtype == torch.float32 and t.shape == terms[0].shape and t.data_ptr() % 16 == 0 for t in terms) and len(terms) <= 8:
        return ag.SumTermsFn.apply(*terms) if _grad() else ops.add_n([t.contiguous() for t in terms])
    out = terms[0]
    for t in terms[1:]:
        out = out + t
    return out


def stack_rows(xs) -> Tensor:
    if _grad():
        return ag.StackRowsFn.apply(*xs)
    return ag.StackRowsFn.forward(_NoCtx(), *xs)


def xent_smooth_losses(logits: Tensor, target: Tensor, denom, smoothing: float, pad: int, G: int, grad_dtype):
    """G device scalars [1]: label-smoothed KL of softmax(logits) per group of rows / denom (autograd.XentSmoothLossFn)."""
    if _grad():
        if grad_dtype != torch.float32 and type(getattr(logits, "grad_fn", None)).__name__ == "LinearFnBackward":
            # the logits' gradient will travel as an attribute of a never-initialised placeholder (autograd.XentSmoothLossFn): the producing
            # projection must refuse the placeholder if the attribute got lost on the way (hook, retain_grad, a re-wrapping view)
            logits.grad_fn._bist_expect_dz = True
        return ag.XentSmoothLossFn.apply(logits, target, denom, smoothing, pad, G, grad_dtype)
    return ag.XentSmoothLossFn.forward(_NoCtx(), logits, target, denom, smoothing, pad, G, grad_dtype)


def switch_logits_ok(w: Tensor, bias: Optional[Tensor], parts) -> bool:
    d = parts[0].shape[-1]
    return (POINTER_ATTN and w.is_cuda and w.dtype in (torch.bfloat16, torch.float32) and 1 <= len(parts) <= 4 and w.shape[0] <= 4
            and w.shape[1] == len(parts) * d and d % 8 == 0 and w.stride(1) == 1 and w.stride(0) % 8 == 0 and w.data_ptr() % 16 == 0
            and all(p.dtype == w.dtype and p.shape[-1] == d and p.numel() == parts[0].numel() for p in parts) and (bias is None or bias.dtype == w.dtype))


def switch_logits(w: Tensor, bias: Optional[Tensor], parts, out_dtype=None) -> Tensor:
    """lin(cat(parts, -1)) as [rows, ns] in f32 (default) or the operand dtype (autograd.SwitchLogitsFn)."""
    if _grad():
        return ag.SwitchLogitsFn.apply(w, bias, out_dtype, *parts)
    return ag.SwitchLogitsFn.forward(_NoCtx(), w, bias, out_dtype, *parts)


POINTER_ATTN = os.environ.get("BIST_POINTER_ATTN", "1") != "0"      # tuning aid: 0 = the pointer attentions on the generic attention core + a text-vector product


def pointer_attn_ok(q: Tensor, k: Tensor, enc: Tensor) -> bool:
    """Inside the envelope of bist_pointer_attn_fwd / _bwd: <= 32 query rows per sequence, <= 128 positions, channels a multiple of 64."""
    return (POINTER_ATTN and q.is_cuda and q.dim() == 3 and q.dtype == k.dtype == enc.dtype and q.dtype in (torch.bfloat16, torch.float32)
            and q.shape[1] <= 32 and k.shape[1] <= 128 and q.shape[2] % 64 == 0 and k.shape == enc.shape)


def pointer_attn(q: Tensor, k: Tensor, enc: Tensor, mask: Optional[Tensor], text: Optional[Tensor], unk: int = 0):
    """(p f32 [B,Lt,L], text vector [B,Lt,d]) of one pointer attention (autograd.PointerAttnFn); mask [B,1,L] / [B,L] / [1,L] boolean."""
    m8 = None
    if mask is not None:
        m8 = ag._mask_u8(mask.reshape(-1, mask.shape[-1]))
        m8 = m8 if m8.is_contiguous() else m8.contiguous()
    if text is not None:
        text = text.contiguous()
    if _grad():
        return ag.PointerAttnFn.apply(q, k, enc, m8, text, unk)
    return ag.PointerAttnFn.forward(_NoCtx(), q, k, enc, m8, text, unk)


def log_softmax(x):
    return ag.LogSoftmaxFn.apply(x) if _grad() else ops.log_softmax(x)


def label_smoothing_loss(logp, target, denom, smoothing, pad):
    """sum_rows KL(row) / denom as a device scalar [1] (denom: device int64 [1] or None)."""
    if _grad():
        return ag.LabelSmoothingLossFn.apply(logp, target, denom, smoothing, pad)
    return ops.sum_div(ops.label_smoothing_rows(logp, target, smoothing, pad), denom)


# ---- dropout seeds: every dropout site draws a fresh 64-bit seed; the mask itself is a pure function of
# (seed, element index) generated inside the kernels (forward epilogue and its backward).
_SEED = [0x5EED, 0]


def manual_seed(seed: int) -> None:
    _SEED[0], _SEED[1] = int(seed) & 0xFFFFFFFF, 0


SEED_LOG = None           # tests: a list that receives (kind, module, seed) for every dropout site a training-mode forward pass draws


def next_seed(kind: Optional[str] = None, module=None) -> int:
    _SEED[1] += 1
    seed = ((_SEED[0] << 32) ^ (_SEED[1] * 0x9E3779B1)) & 0x7FFFFFFFFFFFFFFF
    if SEED_LOG is not None:
        SEED_LOG.append((kind, module, seed))
    return seed


def drop_args(module) -> dict:
    """kwargs for linear(): the dropout of a SublayerConnection / nn.Dropout holder in training mode."""
    p = float(getattr(module, "p", 0.0))
    if module.training and p > 0.0:
        return {"drop_p": p, "drop_seed": next_seed("sub", module)}
    return {}


# ---- branch concurrency: t2s / s2t / caption reasoning are independent chains of small kernels; each gets
# its own HIP stream (fork = side.wait_stream(main), join = main.wait_stream(side)).  Captured into a
# hipGraph the forks become parallel branches of the graph.
CONCURRENT = os.environ.get("BIST_CONCURRENT", "1") != "0"      # tuning aid: BIST_CONCURRENT=0 keeps one stream
_SIDE = {}


def side_stream(i: int):
    key = (torch.cuda.current_device(), i)
    st = _SIDE.get(key)
    if st is None:
        if not _streams_ready() or key not in _SIDE:      # (the split executor's stream set, else a stream of the framework's pool)
            _SIDE[key] = torch.cuda.Stream()
        st = _SIDE[key]
    return st





BRANCH_V = os.environ.get("BIST_BRANCH_V", "0") != "0"      # tuning aid: value projection of a direction inside its branch (training); measured slower (14.5 vs 13.0 ms)


PIPELINE_DECODER = os.environ.get("BIST_PIPELINE_DECODER", "1") != "0"      # tuning aid: decoder layer l under reasoning layer l+1


# Deferred optimiser (bist_amd/train.py): while a step applies the previous step's update piece by piece on its own stream, the model
# announces -- on the MAIN stream -- that it is about to read the parameters of piece k for the first time (piece 0: everything
# outside the layer stacks, piece 1 + l: reasoning / caption / decoder layer l); the trainer's gate waits for that piece's event.
PARAM_GATE = None


def param_gate(k: int) -> None:
    if PARAM_GATE is not None:
        PARAM_GATE(k)


# Overlapped gradient exchange (bist_amd/train.py): the layer loop marks the point in the forward pass behind which every launch belongs
# to layers >= k (their ahead-issued value projections included); when the backward pass has issued everything behind the mark, the
# trainer's callback signals "the gradients of layers >= k are final" on every stream of the step.
BUCKET_MARK = None               # (set of layer indices k that open a bucket, callback(k))


def bucket_mark(x: Tensor, k: int) -> Tensor:
    if BUCKET_MARK is None or not _grad() or k not in BUCKET_MARK[0] or not x.requires_grad:
        return x
    return ag.BucketMarkFn.apply(x, k, BUCKET_MARK[1])


# Streams a captured graph may span when it is replayed by the RUNTIME's graph executor (the capturing stream included).  Every replay
# crash seen so far (host-side segmentation fault inside hipGraphLaunch, round 3) was in a graph captured over FOUR streams -- an optimiser
# stream of its own beside the three of a training step, the four-stream first step of a decode turn -- and none in tens of thousands of
# replays of three-stream graphs; tests/test_soak_gpu.py holds the three-stream forms to that.  A capture for the runtime's executor that
# ends with more streams joined is refused.  (A ``Graph`` replayed by the split executor -- one single-queue graph per stream,
# bist_amd/graphsplit.py -- never hands a multi-branch graph to the runtime and is not limited.)
MAX_CAPTURE_STREAMS = 3
CAPTURE_FOR_RUNTIME = False      # True while a capture for the runtime's executor is being recorded: model code then keeps to three streams
SPLIT_GRAPHS = os.environ.get("BIST_SPLIT_GRAPH", "1") != "0"      # tuning aid: 0 = every graph through the runtime's executor


def _streams_ready() -> bool:
    """The package's streams on hardware queues AND dispatch pipes of their own (split executor usable).  The device has four pipes
    (graphsplit.distinct_streams): [capturing stream, side 0, side 1, side 2 = leaf] -- side 2 is an inference stream (the third chain of
    the fused reasoning layer), the leaf stream a training one (backward leaves); no step uses both."""
    from . import graphsplit as GS
    if not SPLIT_GRAPHS:
        return False
    dev = torch.cuda.current_device()
    if (dev, "cap") in _SIDE:
        return True
    if torch.cuda.is_current_stream_capturing() or not GS.usable():
        return False
    torch.cuda.synchronize()
    st = GS.distinct_streams(4)
    _SIDE[(dev, "cap")] = st[0]
    for i in range(3):
        _SIDE[(dev, i)] = st[1 + i]
    _SIDE[(dev, "leaf")] = st[3]
    return True


def main_stream():
    """The stream the package captures its graphs on -- and the split executor replays their main chain on -- or None when graphs go to the
    runtime's executor.  A caller that makes it the CURRENT stream (``torch.cuda.set_stream``; bench.py does) spares every replay the
    two event hops between its own stream and this one; from the NULL stream each of those also costs the runtime a walk over all
    streams of the process."""
    return _SIDE.get((torch.cuda.current_device(), "cap")) if _streams_ready() else None


def copy_stream():
    """A stream for host-to-device copies (bist_amd/data/feeder.py) that shares its hardware queue with none of the package's four streams --
    a copy and the cast ordered behind it hold their queue for milliseconds -- or None when the split executor is not in use."""
    if not _streams_ready():
        return None
    dev = torch.cuda.current_device()
    st = _COPY.get(dev)
    if st is None and not torch.cuda.is_current_stream_capturing():
        from . import graphsplit as GS
        torch.cuda.synchronize()
        # (on the dispatch pipe of the fourth stream: in a training step that is the caption chain, the one with slack)
        chains = [_SIDE[(dev, k)] for k in ("cap", 0, 1, 2)]
        st = GS.queue_distinct_stream(chains, pipe_with=chains[COPY_PIPE] if 0 <= COPY_PIPE < 4 else None)
        _COPY[dev] = st
        return st
    return _COPY.get(dev)


_COPY = {}
# whose dispatch pipe the feeder's copy queue shares (index into [capture stream, side 0, side 1, side 2]; -1: the first free queue)
COPY_PIPE = int(os.environ.get("BIST_COPY_PIPE", "2"))


def fourth_stream():
    """The package's fourth stream for work that may have a chain of its own -- or None while a capture for the runtime's graph executor is
    being recorded (MAX_CAPTURE_STREAMS) and when the split executor is not in use."""
    return None if CAPTURE_FOR_RUNTIME else leaf_stream()


def leaf_stream():
    """The stream of the backward pass's leaf products (weight gradients and video-tensor gradients of the frame-grid products: nothing
    on a direction's chain reads them), or None when the step is replayed by the runtime's executor (a fourth stream there: see
    MAX_CAPTURE_STREAMS)."""
    return _SIDE.get((torch.cuda.current_device(), "leaf"))


class Graph:
    """A captured launch sequence and its replay.  ``with capture_graph(g): ...`` then ``g.replay()``.  When the split executor is usable
    (bist_amd/graphsplit.py: its self-test passes in this process) the capture is labelled per stream and -- if it spans more than one --
    replayed as one single-queue hipGraph per stream tied by device flags; otherwise, and for one-stream captures, by the runtime."""

    def __init__(self, split: Optional[bool] = None):
        self.want_split = SPLIT_GRAPHS if split is None else bool(split)
        self.cuda_graph = None
        self.split = None
        self.streams = 0

    def replay(self) -> None:
        if self.split is not None:
            self.split.launch()
        else:
            self.cuda_graph.replay()

    def errors(self) -> int:
        return self.split.errors() if self.split is not None else 0

    def __del__(self):
        # Destroying the EXECUTABLE of a multi-stream capture (hipGraphExecDestroy, the runtime's multi-branch executor) corrupts host memory:
        # in the runtime-executor soak the size of a live tensor came back one less -- a reference count released in freed memory -- or the
        # process died, in 7 runs of 10, always in the first capture after a graph had been dropped; also with the device drained first;
        # never (6 of 6) when the dropped graphs were kept alive.  So such a graph is RETIRED, not destroyed: it stays referenced (with its
        # memory pool) in a module list.  Only the fallback executor is affected -- a split graph never instantiates the capture, and
        # destroying the captured hipGraph itself and single-stream executables is clean (300 captures in the split soak) -- and a graph
        # dies only when a batch geometry changes or its trainer / decoder goes; beyond RETIRED_MAX the oldest is destroyed after all.
        try:
            if self.cuda_graph is not None and self.split is None and self.streams > 1:
                _RETIRED.append(self.cuda_graph)
                if len(_RETIRED) > RETIRED_MAX:
                    if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                        torch.cuda.synchronize()
                    del _RETIRED[0]
        except Exception:
            pass


_RETIRED: list = []
RETIRED_MAX = int(os.environ.get("BIST_RETIRED_GRAPHS_MAX", "64"))      # multi-stream runtime-executor graphs kept alive instead of destroyed (see Graph.__del__)


class capture_graph:
    """``with capture_graph(graph):`` = torch.cuda.graph(graph, capture_error_mode="thread_local") with Python's cyclic collector paused.
    Cyclic garbage may own hipGraphs and device buffers of EARLIER captures (a model that decoded, a trainer: nn.Module trees are cycles, so
    they die whenever the collector happens to run); destroying those is a runtime call a stream capture does not allow, and the process
    aborts.  torch.cuda.graph stopped collecting before a capture (2.10: only with torch.compiler.config.force_cudagraph_gc), so the
    collector is simply not allowed to run between capture begin and end.  (thread_local: other threads -- the RCCL watchdog -- may touch
    the runtime during a capture.)  ``graph``: a ``Graph`` (split executor when usable) or a plain torch.cuda.CUDAGraph (runtime's
    executor: on exit the streams that joined the capture are counted, MAX_CAPTURE_STREAMS, RuntimeError beyond)."""

    LAST_STREAMS = 0          # streams of the most recent capture (tests)

    def __init__(self, graph, **kw):
        self.kw = {k: v for k, v in kw.items() if v is not None}
        self.g = graph if isinstance(graph, Graph) else None
        self.raw = None if self.g is not None else graph
        self.lab = None

    def __enter__(self):
        import gc
        from . import graphsplit as GS
        split = False
        if self.g is not None:
            split = self.g.want_split and CONCURRENT and _streams_ready()
            self.g.cuda_graph = torch.cuda.CUDAGraph(keep_graph=True) if split else torch.cuda.CUDAGraph()
            self.raw = self.g.cuda_graph
            if split:
                self.kw.setdefault("stream", _SIDE[(torch.cuda.current_device(), "cap")])
                self.lab = GS.Labels()
        self.split = split
        global CAPTURE_FOR_RUNTIME
        self._prev_cfr, CAPTURE_FOR_RUNTIME = CAPTURE_FOR_RUNTIME, not split
        self.ctx = torch.cuda.graph(self.raw, capture_error_mode="thread_local", **self.kw)
        self.was = gc.isenabled()
        gc.disable()
        try:
            if self.lab is not None:
                self.lab.__enter__()
            res = self.ctx.__enter__()
            self.origin = torch.cuda.current_stream().cuda_stream
            return res
        except BaseException:
            CAPTURE_FOR_RUNTIME = self._prev_cfr
            if self.lab is not None:
                self.lab.__exit__(None, None, None)
            if self.was:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc
        from . import graphsplit as GS
        global CAPTURE_FOR_RUNTIME
        CAPTURE_FOR_RUNTIME = self._prev_cfr
        n = 0
        try:
            if exc[0] is None:
                try:
                    n = 1 + len(live_side_streams()) if torch.cuda.is_current_stream_capturing() else 0
                except Exception:
                    n = 0
            res = self.ctx.__exit__(*exc)
        finally:
            if self.lab is not None:
                self.lab.__exit__(None, None, None)
            if self.was:
                gc.enable()
        capture_graph.LAST_STREAMS = n
        if self.g is not None:
            self.g.streams = n
        if exc[0] is None and self.split:
            sp = GS.SplitGraph(self.raw, self.lab, self.origin)
            if sp.n_chains > 1:
                self.g.split = sp
            # (a one-stream capture: the runtime replays it on its single-queue path as it is)
        elif exc[0] is None and n > MAX_CAPTURE_STREAMS:
            raise RuntimeError(f"bist_amd: a hipGraph capture spanning {n} streams for the runtime's graph executor (at most "
                               f"{MAX_CAPTURE_STREAMS}: replays of wider graphs have crashed inside hipGraphLaunch); it must not be replayed"
                               f" [streams of this package: {sorted((str(k[1]), hex(v.cuda_stream)) for k, v in _SIDE.items())}]")
        return res


VALUES_AHEAD = os.environ.get("BIST_VALUES_AHEAD", "1") != "0"      # tuning aid: value projections of layer l+1 on the caption stream


PERMUTED_T2S = os.environ.get("BIST_PERMUTED_T2S", "1") != "0"      # tuning aid: t2s stage 1 on the region-major copy (training)


FUSED_ST1 = os.environ.get("BIST_FUSED_ST1", "1") != "0"      # tuning aid: 0 = stage 1 of the inference path as separate launches


FUSED_DECODE = os.environ.get("BIST_FUSED_DECODE", "1") != "0"      # tuning aid: 0 = the decoder layers of a decode step as separate launches


EVAL_SCHED = int(os.environ.get("BIST_EVAL_SCHED", "1"))      # tuning aid: stream schedule of the fused inference layer (0: one fork after the input projection, 1: two chains forked ahead of it, 2: stage-1 launches on the main stream)


def live_side_streams():
    """The side streams of this device whose present position a stream may wait for: all of them, or -- while the current stream is
    being captured into a hipGraph -- those that have joined the capture (a wait on a stream outside it would leave the graph)."""
    cap = torch.cuda.is_current_stream_capturing()
    out = []
    seen = set()
    for (dev, _), st in _SIDE.items():
        if dev != torch.cuda.current_device() or st.cuda_stream in seen or st.cuda_stream == torch.cuda.current_stream().cuda_stream:
            continue
        seen.add(st.cuda_stream)
        if cap:
            with torch.cuda.stream(st):
                if not torch.cuda.is_current_stream_capturing():
                    continue
        out.append(st)
    return out


def step_streams(main) -> list:
    """`main` and every side stream of this device that carries (or may carry) launches of the running pass: all of them, or -- while
    `main` is being captured -- those that have joined the capture.  Unlike live_side_streams() this does not depend on the calling
    thread's current stream (the autograd thread's is whichever stream the node it runs was recorded on)."""
    with torch.cuda.stream(main):
        cap = torch.cuda.is_current_stream_capturing()
    out, seen = [main], {main.cuda_stream}
    for (dev, _), st in _SIDE.items():
        if dev != torch.cuda.current_device() or st.cuda_stream in seen:
            continue
        seen.add(st.cuda_stream)
        if cap:
            with torch.cuda.stream(st):
                if not torch.cuda.is_current_stream_capturing():
                    continue
        out.append(st)
    return out


def join_side_streams() -> None:
    """Order the current stream after everything queued on the side streams (end of a backward pass: the weight-gradient
    GEMMs of a side branch write the flat gradient directly, which autograd's own leaf-stream join does not see)."""
    cur = torch.cuda.current_stream()
    for st in live_side_streams():       # (during a capture: only the streams that joined it)
        cur.wait_stream(st)


def column_block(w: Tensor, j: int, d: int) -> Tensor:
    """w[:, j*d:(j+1)*d] of a concat-consuming weight (vc_combine_W, pointer_gen_W).  Under the trainer the block
    carries its own view of the flat gradient, so the weight-gradient GEMM accumulates in place and autograd never
    materialises a zero-padded full-size gradient per block (slice_backward: fill + copy + add)."""
    blk = w[:, j * d:(j + 1) * d]
    gv = getattr(w, "_grad_view", None)
    if gv is not None:
        blk._grad_view = gv[:, j * d:(j + 1) * d]
    return blk


def pack_rows(*ws: Tensor) -> Tensor:
    """Concatenate weight matrices / biases row-wise (device-side data movement only)."""
    return torch.cat(ws, dim=0)


class _NoCtx:
    """Stand-in for the autograd context when a Function's forward is called without autograd."""

    def save_for_backward(self, *a):
        pass

    def set_materialize_grads(self, *a):
        pass
