"""Text normalisation, video input projection and the bi-directional spatio-temporal reasoning
layer on the HIP kernels (reference: model/encoder.py).

``VidEncoderLayer4`` is the hot spot (87.8 % of the reference's forward).  The
reference materialises, per direction, a permuted copy of the video tensor, an S- or T-fold
expansion of the query, and K/V projections of both; here

  stage 1  scores = (LN(x) W_q^T + b_q) folded through W_k  x  raw video rows     (one batched GEMM)
           V      = video rows x [W_v(t2s); W_v(s2t)]^T                            (one GEMM for both directions)
           O      = softmax(scores, temporal mask) . V  per (clip, region/segment) (st_stage1_pv kernel)
           Y      = x (un-expanded, added per group in the epilogue) + O W_o^T     (GEMM)
  stage 2  q2f    = stage-2 query folded through its W_k                           (batched GEMM, tiny)
           PY     = softmax(q2f . Y) . Y   per (clip, query position)              (st_stage2 kernel, reads Y once)
           out    = x + (PY W_v^T + b_v) W_o^T                                     (2 tiny GEMMs)

so K is never computed, nothing is permuted or expanded, and the video tensor is read in place.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops
from .. import stamps as STM
from .modules import LayerNorm, MultiHeadedAttention, PositionwiseFeedForward, SublayerConnection, clones

Tensor = torch.Tensor


class Encoder(nn.Module):
    """One LayerNorm per text stream, applied in argument order (reference: encoder.py:11-41)."""

    def __init__(self, size: int, nb_layers: int):
        super().__init__()
        self.norm = nn.ModuleList(LayerNorm(size) for _ in range(nb_layers))
        self.nb_layers = nb_layers

    def forward(self, *seqs):
        out, i = [], 0
        for s in seqs:
            if isinstance(s, list):
                grp = []
                for t in s:
                    grp.append(self.norm[i](t)); i += 1
                out.append(grp)
            elif s is None:
                out.append(None)
            else:
                out.append(self.norm[i](s)); i += 1
        return out


class VidEncoder8(nn.Module):
    """P0: ft['spatiotemporal_ft'] = LN(ReLU(W fts + b))   (reference: encoder.py:55-93).

    The ReLU is the GEMM epilogue; the features are read in place as a [B*T*S, C] matrix.
    """

    def __init__(self, W, a_W, vid_position, v_N, a_N, size, args):
        super().__init__()
        self.v_N, self.a_N, self.args = v_N, a_N, args
        if a_N > 0:
            raise NotImplementedError("audio features are outside the hot path (SURVEY.md 8a; the reference's audio "
                                      "branch reads the undefined args.noW_venc, encoder.py:84)")
        if v_N > 0:
            self.W = W
            self.vid_position = vid_position
            self.in_norm = LayerNorm(size)
            if vid_position is not None:
                raise NotImplementedError("vid_position is always None in the reference (mtn.py:106)")

    def forward(self, b, ft: Dict[str, Tensor]) -> Dict[str, Tensor]:
        if self.v_N > 0:
            fts = b.fts
            if fts.dtype != self.W.weight.dtype:
                fts = Fn.cast(fts, self.W.weight.dtype)
            B, T, S, C = fts.shape
            if fts.is_cuda and not torch.is_grad_enabled() and Fn.CONCURRENT:
                # inference: everything queued so far (the text encoders) is done at this point of the stream; the first reasoning
                # layer forks its query-side chains from HERE, so that they run under the input projection instead of after it.
                # (Recorded ONLY when a layer will wait on it: an event recorded during a hipGraph capture and never waited on stays in
                # the graph as an event-record node of an event object that dies with this call -- replays then crash in the runtime.)
                ev = torch.cuda.Event()
                ev.record()
                ft["_bist_pre_vid"] = ev
            if not torch.is_grad_enabled() and fts.is_cuda:
                # inference: LayerNorm(ReLU(W fts + b)) as ONE launch where the product's LayerNorm epilogue covers the shape (no
                # separate pass over the [B*T*S, 512] activations); training keeps both tensors for the backward pass
                n = self.in_norm
                y = ops.linear(fts.reshape(B * T * S, C), self.W.weight, self.W.bias, act=Fn.ACT_RELU, ln_out=(n.a_2, n.b_2, n.eps))
                ft["spatiotemporal_ft"] = y.view(B, T, S, -1)
            else:
                act = Fn.linear(fts.reshape(B * T * S, C), self.W.weight, self.W.bias, act=Fn.ACT_RELU)
                ft["spatiotemporal_ft"] = self.in_norm(act).view(B, T, S, -1)
        return ft


def _self_attention(sub: SublayerConnection, attn: MultiHeadedAttention, x: Tensor, mask: Optional[Tensor]) -> Tensor:
    """x + MHA(LN(x), LN(x), LN(x), mask): A0/A3 and every other self-attention sublayer."""
    xn, xr = sub.norm.with_residual(x, lazy=True)        # one consumer: the packed Q/K/V projection
    ctx = attn.context(xn, xn, xn, mask)
    return Fn.linear(ctx, attn.linears[3].weight, attn.linears[3].bias, residual=xr, out_shape=x.shape, **Fn.drop_args(sub))


def _cross_attention(sub: SublayerConnection, attn: MultiHeadedAttention, x: Tensor, mem: Tensor, mask: Optional[Tensor]) -> Tensor:
    """x + MHA(LN(x), mem, mem, mask) -- only the query stream is normalised (modules.py:44)."""
    xn, xr = sub.norm.with_residual(x)
    ctx = attn.context(xn, mem, mem, mask)
    return Fn.linear(ctx, attn.linears[3].weight, attn.linears[3].bias, residual=xr, out_shape=x.shape, **Fn.drop_args(sub))


def _feed_forward(sub: SublayerConnection, ff: PositionwiseFeedForward, x: Tensor) -> Tensor:
    """x + FFN(LN(x)); the residual is the second GEMM's epilogue."""
    xn, xr = sub.norm.with_residual(x, lazy=True)        # one consumer: w_1
    return ff(xn, residual=xr, out_drop=Fn.drop_args(sub))


class VidEncoderLayer4(nn.Module):
    """Bi-directional spatio-temporal reasoning layer (reference: encoder.py:95-201).

    Index order, fixed by the reference's run-time counters (encoder.py:173): with both
    directions on, attn 0..5 = A0,A1,A2,A3,A4,A5; sublayer 0..7 = A0,A1,A2,F0,A3,A4,A5,F1; ff 0,1.
    """

    _RUNTIME_ATTRS = ("_frag_train", "_frag_cache", "_vpack", "_v_ready", "_v_event", "_offload_main", "_x_next")

    def __getstate__(self):
        """torch.save / copy.deepcopy: derived device buffers (fragment-ordered weights, packed projections) and per-call stream state
        stay behind -- they are rebuilt on demand (the same filter as MTN / MultimodalDecoder8 apply to their _bist_* attributes)."""
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_bist_") and k not in self._RUNTIME_ATTRS}

    def __init__(self, size, attn, nb_attn, ff, nb_ff, dropout, args):
        super().__init__()
        self.size = size
        self.attn = clones(attn, nb_attn)
        self.ff = clones(ff, nb_ff)
        self.sublayer = clones(SublayerConnection(size, dropout), nb_attn + nb_ff)
        self.args = args
        self._v_ready = self._v_event = None
        if args.enc_st_combine in ("early_sum", "early_dyn"):
            raise NotImplementedError("enc_st_combine=%s cannot run for more than one layer in the reference "
                                      "(decoder.py:123-124 overwrites the video tensor)" % args.enc_st_combine)

    # -- stage 1 ------------------------------------------------------------------------------
    def frag_train(self, ai: int):
        """(W_v, W_o) of attention `ai` in MFMA-fragment order for the TRAINING launches: persistent buffers (captured hipGraphs hold
        their addresses) that the layer loop refreshes once per forward pass, all layers in one launch (decoder.py)."""
        bufs = self.__dict__.setdefault("_frag_train", {})
        out = []
        for j in (2, 3):
            w = self.attn[ai].linears[j].weight
            b_ = bufs.get((ai, j))
            if b_ is None or b_.device != w.device or b_.dtype != w.dtype or b_.shape != w.shape:
                b_ = bufs[(ai, j)] = torch.empty(w.shape, device=w.device, dtype=w.dtype)
            out.append(b_)
        return out

    def _stage1(self, ai: int, si: int, x: Tensor, vft: Tensor, v: Tensor, tmask: Optional[Tensor], direction: int,
                permuted: bool = False, train_fused: Optional[Tensor] = None) -> Tensor:
        """A1 (direction 0, encoder.py:110-123) / A4 (direction 1, encoder.py:142-150) -> [B,G,Lq,d].
        permuted (t2s only): vft and v are region-major [B,S,T,*] (Fn.permute_ts): the step is then the s2t form with the two
        axes exchanged plus the frame mask -- contiguous score runs and value tiles instead of 16-byte pieces."""
        attn, sub = self.attn[ai], self.sublayer[si]
        B, T, S, d = vft.shape              # permuted: (B, S, T, d) -- "T" counts the groups, "S" the keys
        if permuted:
            direction = 1
        Lq, h, dk = x.shape[1], attn.h, attn.d_k
        xn, xr = sub.norm.with_residual(x, lazy=True)
        q = Fn.linear(xn, attn.linears[0].weight, attn.linears[0].bias)                           # [B*Lq, d]
        qf = Fn.head_fold(q, attn.linears[1].weight, h, 1.0 / math.sqrt(dk)).view(B, Lq * h, d)   # rows (i, hh)
        if train_fused is not None:      # training: the same single launch, with the dropouts and the side outputs the backward pass reads
            adrop = Fn.attn_drop(attn)            # (seeds drawn in the order of the unfused path: probabilities, then sublayer output)
            kw = Fn.drop_args(sub)
            v_ready = self.__dict__.get("_v_ready")
            if v is not None and v_ready is not None:
                torch.cuda.current_stream().wait_stream(v_ready)
            v_event = self.__dict__.get("_v_event")
            if v is not None and v_event is not None:
                torch.cuda.current_stream().wait_event(v_event)
            STM.lmark("dir%d pre-st1 (Qf ready, V awaited)" % direction)
            y, xnext = Fn.st_stage1_fused_train(qf, xr, vft, train_fused if v is None else None, tmask, attn, self.frag_train(ai), h=h, direction=direction,
                                                attn_drop=adrop, sub_drop=(kw["drop_p"], kw["drop_seed"]) if kw else None, v=v,
                                                offload=direction == 0 and bool(self.__dict__.get("_offload_main")))
            self.__dict__["_x_next"] = xnext       # x again, for stage 2 (one consumer per tensor: no accumulation launch by autograd)
            return y
        if v is None:           # inference: value projection, scores, softmax, P.V, output projection and residual in one launch
            wv, wo = self._frag_weights(ai)
            return ops.st_stage1_fused(qf, vft, tmask, wv, attn.linears[2].bias, wo, attn.linears[3].bias, x, h=h, direction=direction)
        scores = Fn.st_scores(qf, vft.view(B, T * S, d))
        v_ready = self.__dict__.get("_v_ready")
        if v_ready is not None:
            torch.cuda.current_stream().wait_stream(v_ready)      # V comes from the value-projection stream
        v_event = self.__dict__.get("_v_event")
        if v_event is not None:
            torch.cuda.current_stream().wait_event(v_event)       # V was projected ahead of this layer on another stream
        o = Fn.st_stage1_pv(scores, v, tmask, B=B, T=T, S=S, Lq=Lq, h=h, dk=dk, direction=direction, drop=Fn.attn_drop(attn))
        G = o.shape[1]
        y = Fn.linear(o, attn.linears[3].weight, attn.linears[3].bias, residual=xr, res_map=(G * Lq, Lq), **Fn.drop_args(sub))
        return y.view(B, G, Lq, d)

    def _stage1_query(self, ai: int, si: int, x_in: Tensor, query_mask):
        """Query side of one direction up to the fused stage-1 launch: self-attention sublayer (A0 / A3), then LayerNorm, query
        projection and the fold through W_k of the stage-1 attention -> (x, Qf [B, Lq*h, d])."""
        x = _self_attention(self.sublayer[si], self.attn[ai], x_in, query_mask)
        attn, sub = self.attn[ai + 1], self.sublayer[si + 1]
        B, Lq, d = x.shape
        xn, _ = sub.norm.with_residual(x, lazy=True)
        q = Fn.linear(xn, attn.linears[0].weight, attn.linears[0].bias)
        return x, Fn.head_fold(q, attn.linears[1].weight, attn.h, 1.0 / math.sqrt(attn.d_k)).view(B, Lq * attn.h, d)

    def _stage1_fused(self, ai: int, x: Tensor, qf: Tensor, vft: Tensor, kmask, direction: int) -> Tensor:
        attn = self.attn[ai]
        wv, wo = self._frag_weights(ai)
        return ops.st_stage1_fused(qf, vft, kmask, wv, attn.linears[2].bias, wo, attn.linears[3].bias, x, h=attn.h, direction=direction)

    def _frag_weights(self, ai: int):
        """W_v and W_o of attention `ai` in MFMA-fragment order (ops.pack_frag_rows), re-packed INTO THE SAME BUFFERS whenever the
        parameters change (ops.weights_key covers in-place updates and the trainer's raw Adam kernel), so that captured hipGraphs
        that read them stay valid."""
        attn = self.attn[ai]
        wv, wo = attn.linears[2].weight, attn.linears[3].weight
        key = ops.weights_key(wv, wo)
        cache = self.__dict__.setdefault("_frag_cache", {})
        hit = cache.get(ai)
        if hit is None or hit[0] != key or hit[1].device != wv.device:
            bufs = (None, None) if hit is None or hit[1].device != wv.device or hit[1].dtype != wv.dtype else (hit[1], hit[2])
            hit = (key, ops.pack_frag_rows(wv.detach(), bufs[0]), ops.pack_frag_rows(wo.detach(), bufs[1]))
            cache[ai] = hit
        return hit[1], hit[2]

    # -- stage 2 ------------------------------------------------------------------------------
    def _stage2(self, ai: int, si: int, x: Tensor, y: Tensor, gmask: Optional[Tensor]) -> Tensor:
        """A2 (encoder.py:125-134) / A5 (encoder.py:152-165) -> [B,Lq,d]."""
        attn, sub = self.attn[ai], self.sublayer[si]
        B, G, Lq, d = y.shape
        h, dk = attn.h, attn.d_k
        xn, xr = sub.norm.with_residual(x, lazy=True)
        q = Fn.linear(xn, attn.linears[0].weight, attn.linears[0].bias)
        q2f = Fn.head_fold(q, attn.linears[1].weight, h, 1.0 / math.sqrt(dk)).view(B, Lq, h, d)
        py, rowsum = Fn.st_stage2(q2f, y, gmask, h=h, drop=Fn.attn_drop(attn))
        if rowsum is None:
            ctx = Fn.head_unfold(py.view(B * Lq, h * d), attn.linears[2].weight, attn.linears[2].bias, h)
        else:       # dropped probabilities do not sum to one: P'(Y W^T + b) = (P'Y) W^T + rowsum(P') b
            ctx = Fn.head_unfold(py.view(B * Lq, h * d), attn.linears[2].weight, None, h)
            ctx = Fn.scaled_bias(ctx, rowsum, attn.linears[2].bias, h)
        return Fn.linear(ctx, attn.linears[3].weight, attn.linears[3].bias, residual=xr, out_shape=(B, Lq, d), **Fn.drop_args(sub))

    def value_projection(self, vft: Tensor):
        """V of A1 and A4 in one GEMM over the video tensor: returns (v_t2s, v_s2t) column views."""
        both = self.args.t2s and self.args.s2t
        B, T, S, d = vft.shape
        if both:
            a1, a4 = self.attn[1], self.attn[4]
            if torch.is_grad_enabled():      # training: one GEMM per direction, so each dV is a whole tensor
                x2 = vft.view(B * T * S, d)
                return (Fn.linear(x2, a1.linears[2].weight, a1.linears[2].bias).view(B, T, S, d),
                        Fn.linear(x2, a4.linears[2].weight, a4.linears[2].bias).view(B, T, S, d))
            else:
                key = ops.weights_key(a1.linears[2].weight, a4.linears[2].weight, a1.linears[2].bias, a4.linears[2].bias)
                hit = self.__dict__.get("_vpack")
                if hit is None or hit[0] != key:
                    # re-packed INTO the same buffers (captured hipGraphs keep reading them) whenever the parameters changed --
                    # in place (optimizer.step, load_state_dict: _version) or behind autograd's back (Trainer: ops.WEIGHTS_EPOCH)
                    same = hit is not None and hit[1].device == a1.linears[2].weight.device and hit[1].dtype == a1.linears[2].weight.dtype
                    w_new = Fn.pack_rows(a1.linears[2].weight, a4.linears[2].weight)
                    b_new = Fn.pack_rows(a1.linears[2].bias, a4.linears[2].bias)
                    if same:
                        hit[1].copy_(w_new); hit[2].copy_(b_new)
                        hit = (key, hit[1], hit[2])
                    else:
                        hit = (key, w_new, b_new)
                    self.__dict__["_vpack"] = hit
                w, bb = hit[1], hit[2]
            v = Fn.linear(vft.view(B * T * S, d), w, bb).view(B, T, S, 2 * d)
            return v[..., :d], v[..., d:]
        a = self.attn[1]
        v = Fn.linear(vft.view(B * T * S, d), a.linears[2].weight, a.linears[2].bias).view(B, T, S, d)
        return (v, None) if self.args.t2s else (None, v)

    def train_value(self, vft_a: Tensor, ai: int):
        B, T, S, d = vft_a.shape
        a = self.attn[ai]
        return Fn.linear(vft_a.view(B * T * S, d), a.linears[2].weight, a.linears[2].bias, leaf=True).view(B, T, S, d)

    def train_values(self, vft_a: Tensor, vft_b: Tensor):
        """Training: one value GEMM per direction (each dV is a whole tensor) -> (v_t2s, v_s2t)."""
        B, T, S, d = vft_a.shape
        a1, a4 = self.attn[1], self.attn[4]
        return (Fn.linear(vft_a.view(B * T * S, d), a1.linears[2].weight, a1.linears[2].bias).view(B, T, S, d),
                Fn.linear(vft_b.view(B * T * S, d), a4.linears[2].weight, a4.linears[2].bias).view(B, T, S, d))

    def forward(self, in_ft: Dict[str, Tensor], ft: Dict[str, Tensor], b) -> Dict[str, Tensor]:
        vft = ft["spatiotemporal_ft"]
        fan = ft.get("_bist_vft_fan")                      # aliases whose gradients are summed in one pass (training)
        take = fan.take if fan is not None else (lambda: vft)
        fan_p = ft.get("_bist_vftp_fan")                   # training: aliases of the region-major copy [B,S,T,d] for the t2s direction
        permuted = fan_p is not None
        take_t2s = fan_p.take if permuted else take
        vft_t2s, vft_s2t = take_t2s(), take()
        t2s_on = (not hasattr(self.args, "t2s")) or self.args.t2s
        s2t_on = (not hasattr(self.args, "s2t")) or self.args.s2t
        concurrent = t2s_on and s2t_on and Fn.CONCURRENT and vft.is_cuda
        v_stream = None
        main = torch.cuda.current_stream() if concurrent else None

        def branch_v(ai):
            return self.train_value(take_t2s() if ai == 1 else take(), ai)

        train_fused = bool(ft.get("_bist_fused_train")) and torch.is_grad_enabled() and t2s_on and s2t_on
        own_mode = Fn.fused_train_own_v(vft.shape[1])
        own_v = train_fused and own_mode == 1                 # both directions' values projected (and saved) by the fused launch
        own_v_s2t = train_fused and own_mode in (1, 2)        # ... the s2t direction's
        per_branch_v = torch.is_grad_enabled() and t2s_on and s2t_on and Fn.BRANCH_V and "_bist_v_pre" not in ft and not own_v
        v_t2s = v_s2t = None
        pre = ft.pop("_bist_v_pre", None)                  # (v_t2s, v_s2t, event): projected ahead by the layer loop (decoder.py)
        self._v_event = None
        # inference at the production width: each direction's stage 1 is one fused launch that projects the values itself
        B_, T_, S_, d_ = vft.shape
        x0 = in_ft["t2s"] if t2s_on else in_ft["s2t"]
        fused = (Fn.FUSED_ST1 and not torch.is_grad_enabled() and vft.is_cuda and pre is None
                 and all(ops.st_stage1_fused_ok(T_, S_, x0.shape[1], d_, self.attn[0].h, dr, vft.dtype)
                         for dr, on in ((0, t2s_on), (1, s2t_on)) if on))
        if fused or own_v:
            pass
        elif pre is not None:
            v_t2s, v_s2t, ev = pre
            main.wait_event(ev) if main is not None else torch.cuda.current_stream().wait_event(ev)
            if v_s2t is None and not own_v_s2t:
                v_s2t = self.train_value(take(), 4)
        elif per_branch_v:
            # training: one value GEMM per direction (each dV is a whole tensor), issued INSIDE its branch, so that the
            # projection and its two backward products run on that branch's stream
            pass
        elif torch.is_grad_enabled() and t2s_on and s2t_on:
            v_t2s = self.train_value(take_t2s(), 1)
            v_s2t = None if own_v_s2t else self.train_value(take(), 4)
        elif concurrent and not (ft.get("_bist_cap_fork") and Fn.fourth_stream() is None):
            # (inference only: under autograd a third forked stream makes hipGraph capture of the training step crash in
            # the HIP runtime -- also with every side stream joined explicitly after backward; beside a forked caption layer it would be
            # the capture's FOURTH stream, which only the split executor's graphs may have: Fn.MAX_CAPTURE_STREAMS)
            # The value projections are the layer's big GEMMs and depend on the video tensor only: they run on
            # their own stream, under the query-side chains (self-attention, LayerNorm, Q projection, fold) of the
            # two directions, and are awaited just before the stage-1 cores.
            v_stream = Fn.side_stream(2)
            v_stream.wait_stream(main)
            with torch.cuda.stream(v_stream):
                v_t2s, v_s2t = self.value_projection(take())
        else:
            v_t2s, v_s2t = self.value_projection(take())
        self._v_ready = v_stream

        trace = self.__dict__.get("_bist_trace")      # debug hook: per-stage outputs (tests compare them with the reference's sublayer outputs)
        # training: (spatial_out_norm, temporal_out_norm) of the layer loop (decoder.py:127,129), applied at the end of each direction's chain on
        # that chain's stream; the stream then reaches the next layer THROUGH the norm's autograd node (the next layer's gradient is added
        # inside the norm's backward kernel: one consumer per tensor, and the s2t chain's backward never leaves its stream)
        out_norms = ft.get("_bist_out_norms") if (torch.is_grad_enabled() and t2s_on and s2t_on) else None

        def t2s_branch(ai, si, fi):
            x = _self_attention(self.sublayer[si], self.attn[ai], STM.through(in_ft["t2s"], "t2s in", True), b.query_mask)     # A0
            x = STM.through(x, "t2s A0", True)
            y = self._stage1(ai + 1, si + 1, x, vft_t2s, branch_v(ai + 1) if per_branch_v else v_t2s, b.temporal_mask, 0,
                             permuted=permuted, train_fused=(take() if own_v else True) if train_fused else None)     # A1
            x = self.__dict__.pop("_x_next", x)
            y = STM.through(y, "t2s st1", True)
            z = self._stage2(ai + 2, si + 2, x, y, None)                                          # A2
            z = STM.through(z, "t2s st2", True)
            in_ft["t2s"] = STM.through(_feed_forward(self.sublayer[si + 3], self.ff[fi], z), "t2s ff", True)                   # F0
            if trace is not None:
                trace.update(t2s_self=x, t2s_stage1=y, t2s_stage2=z, t2s_ff=in_ft["t2s"])
            if out_norms is not None:       # the layer loop's output norm of this stream, here on ITS stream; the stream goes on through the norm's node
                ft["spatial_ft"], in_ft["t2s"] = Fn.layernorm_res(in_ft["t2s"], out_norms[0].a_2, out_norms[0].b_2, out_norms[0].eps)

        def s2t_branch(ai, si, fi):
            x = _self_attention(self.sublayer[si], self.attn[ai], STM.through(in_ft["s2t"], "s2t in", True), b.query_mask)     # A3
            x = STM.through(x, "s2t A3", True)
            y = self._stage1(ai + 1, si + 1, x, vft_s2t, branch_v(ai + 1) if per_branch_v else v_s2t, None, 1,
                             train_fused=(take() if own_v_s2t else True) if train_fused else None)    # A4
            x = self.__dict__.pop("_x_next", x)
            y = STM.through(y, "s2t st1", True)
            z = self._stage2(ai + 2, si + 2, x, y, b.temporal_mask)                               # A5
            z = STM.through(z, "s2t st2", True)
            in_ft["s2t"] = STM.through(_feed_forward(self.sublayer[si + 3], self.ff[fi], z), "s2t ff", True)                   # F1
            if trace is not None:
                trace.update(s2t_self=x, s2t_stage1=y, s2t_stage2=z, s2t_ff=in_ft["s2t"])
            if out_norms is not None:
                ft["temporal_ft"], in_ft["s2t"] = Fn.layernorm_res(in_ft["s2t"], out_norms[1].a_2, out_norms[1].b_2, out_norms[1].eps)

        # stream schedule of the fused inference layer; beside a caption layer forked onto its own stream (decoder.py) schedule 1 would make
        # the capture span four streams, which the runtime's graph executor is not trusted with (Fn.MAX_CAPTURE_STREAMS): schedule 0 there,
        # unless the graph is replayed by the split executor
        sched = 0 if (ft.get("_bist_cap_fork") and Fn.fourth_stream() is None and Fn.EVAL_SCHED == 1) else Fn.EVAL_SCHED
        pre_vid = ft.pop("_bist_pre_vid", None)
        if pre_vid is not None and not (concurrent and fused and sched in (1, 2)):
            torch.cuda.current_stream().wait_event(pre_vid)      # no schedule below waits on it: consume it here (see VidEncoder8.forward)
        if trace is not None and fused:
            fused = False                                # the traced form keeps every stage's output: separate launches
            v_t2s, v_s2t = self.value_projection(take())
        if concurrent and fused and sched == 1:
            # both directions as whole chains on two side streams, forked ahead of the input projection when this is the first layer
            side, side2 = Fn.side_stream(0), Fn.side_stream(2)
            for st_ in (side, side2):
                if pre_vid is not None:
                    st_.wait_event(pre_vid)
                else:
                    st_.wait_stream(main)
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side2):
                xt, qft = self._stage1_query(0, 0, in_ft["t2s"], b.query_mask)
                side2.wait_event(ev)
                yt = self._stage1_fused(1, xt, qft, vft_t2s, b.temporal_mask, 0)
                in_ft["t2s"] = _feed_forward(self.sublayer[3], self.ff[0], self._stage2(2, 2, xt, yt, None))
            with torch.cuda.stream(side):
                xs, qfs = self._stage1_query(3, 4, in_ft["s2t"], b.query_mask)
                side.wait_event(ev)
                ys = self._stage1_fused(4, xs, qfs, vft_s2t, None, 1)
                in_ft["s2t"] = _feed_forward(self.sublayer[7], self.ff[1], self._stage2(5, 6, xs, ys, b.temporal_mask))
            main.wait_stream(side2); main.wait_stream(side)
        elif concurrent and fused and sched == 2:
            # Inference at the production width.  Main stream: the two fused stage-1 launches (each fills the chip), back to back;
            # side stream: the query-side chains of BOTH directions ahead of them (self-attention, LayerNorm, query projection,
            # fold: small launches that depend on the encoded query only -- in the first layer they are forked from BEFORE the
            # input projection and run under it), then the t2s tail (stage 2, feed-forward) under the s2t stage-1 launch.
            side = Fn.side_stream(0)
            if pre_vid is not None:
                side.wait_event(pre_vid)
            else:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                xt, qft = self._stage1_query(0, 0, in_ft["t2s"], b.query_mask)
                xs, qfs = self._stage1_query(3, 4, in_ft["s2t"], b.query_mask)
                for t_ in (xt, qft, xs, qfs):
                    t_.record_stream(main)
            main.wait_stream(side)
            yt = self._stage1_fused(1, xt, qft, vft_t2s, b.temporal_mask, 0)
            ev = torch.cuda.Event()
            ev.record(main)
            ys = self._stage1_fused(4, xs, qfs, vft_s2t, None, 1)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                yt.record_stream(side)
                zt = self._stage2(2, 2, xt, yt, None)
                in_ft["t2s"] = _feed_forward(self.sublayer[3], self.ff[0], zt)
            zs = self._stage2(5, 6, xs, ys, b.temporal_mask)
            in_ft["s2t"] = _feed_forward(self.sublayer[7], self.ff[1], zs)
            main.wait_stream(side)
        elif concurrent:
            side = Fn.side_stream(0)
            side.wait_stream(main)                    # fork: the two directions share only read-only inputs
            with torch.cuda.stream(side):
                s2t_branch(3, 4, 1)
            self.__dict__["_offload_main"] = True     # the t2s chain runs on the main stream: its off-chain backward products may leave it
            try:
                t2s_branch(0, 0, 0)
            finally:
                self.__dict__["_offload_main"] = False
            main.wait_stream(side)                    # join
            if v_stream is not None:
                main.wait_stream(v_stream)
            self._v_ready = None
            self._v_event = None
            if out_norms is not None:
                in_ft["_norms_done"] = True
        else:
            ai = si = fi = 0
            if t2s_on:
                t2s_branch(ai, si, fi)
                ai, si, fi = ai + 3, si + 4, fi + 1
            if s2t_on:
                s2t_branch(ai, si, fi)
            if out_norms is not None:
                in_ft["_norms_done"] = True
        return in_ft


class CapEncoderLayer(nn.Module):
    """Self-attention, attention to the encoded caption, FFN (reference: encoder.py:203-218)."""

    def __init__(self, size, attn, nb_attn, ff, dropout):
        super().__init__()
        self.size = size
        self.attn = clones(attn, nb_attn)
        self.ff = ff
        self.sublayer = clones(SublayerConnection(size, dropout), nb_attn + 1)

    def forward(self, in_ft, ft, b):
        c = _self_attention(self.sublayer[0], self.attn[0], in_ft["cap"], b.query_mask)
        c = _cross_attention(self.sublayer[1], self.attn[1], c, Fn.fan_take(ft, "encoded_cap"), b.cap_mask)
        in_ft["cap"] = _feed_forward(self.sublayer[2], self.ff, c)
        return in_ft


class AudioEncoderLayer(nn.Module):
    """Parameter container only (the reference builds one even with nb_aenc_blocks == 0, mtn.py:131);
    its forward is outside the hot path."""

    def __init__(self, size, attn, nb_attn, ff, dropout):
        super().__init__()
        self.size = size
        self.attn = clones(attn, nb_attn)
        self.ff = ff
        self.sublayer = clones(SublayerConnection(size, dropout), nb_attn + 1)

    def forward(self, in_ft, ft, b):
        raise NotImplementedError("audio reasoning is outside the hot path (SURVEY.md 8a)")
