"""Training step of the hot path: forward, losses, backward, gradient exchange, Adam -- the body of the
reference's ``run_epoch`` loop (train.py:29-37) + ``SimpleLossCompute`` (model/optimize.py:46-94) +
``NoamOpt.step`` (optimize.py:19-26) for one batch.

MI355X layout: all parameters live in ONE flat buffer in the compute dtype (the tensors the kernels
read; each ``nn.Parameter`` is a view into it), all gradients in ONE flat buffer of the same dtype, and
the fp32 master weights and Adam moments in three more flat buffers.  The layout is chosen for the
kernels, not for registration order:

  * region 0 (prefix): every 1-D parameter (biases, LayerNorm gains/offsets) and the shared embedding
    matrix -- the gradients that are reduced over rows with fp32 atomics.  A parallel fp32 accumulator
    ``acc32`` covers exactly this prefix; the backward kernels add into it directly and one kernel
    folds it into the gradient buffer per step.
  * the Q/K/V projection weights (and biases) of each attention are adjacent, so the packed [3d,d] /
    [2d,d] operands of the fused projection GEMMs are plain views -- no concatenation, and their
    gradients land in place.

Weight gradients are written by the backward GEMMs straight into the flat gradient buffer
(C = dY^T X + C), so a step is: two memsets, forward+backward, one fold kernel, one (optional) RCCL
all-reduce over xGMI of the whole gradient buffer, one fused Adam kernel that also refreshes the
low-precision weights.  No per-parameter optimiser launches, no per-tensor collectives.
"""
from __future__ import annotations

import copy
import os
import time
import re
import weakref
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from ._lib import check, lib
from .model.label_smoothing import LabelSmoothing
from .model.modules import MultiHeadedAttention
from .model.optimize import SimpleLossCompute
from . import functional as Fn
from . import graphsplit as GS
from . import stamps as STM
from . import ops, parallel
from .ops import _stream, dtype_code

ALIGN = 64      # elements; keeps every parameter view 16-byte aligned for the LDS-DMA GEMM path


ADAM_CLEARS = os.environ.get("BIST_ADAM_CLEARS", "0") != "0"      # tuning aid: 1 = Adam on 4 elements per thread that also clears the gradients it consumed, no memset of the gradient
                                                                  # buffer at the head of a replayed step (measured 11.83-11.91 vs 11.72-11.79 ms per step: not adopted)
ALWAYS_COPY_INPUTS = os.environ.get("BIST_ALWAYS_COPY_INPUTS", "0") != "0"      # 1 = every step copies the batch into the graph's static buffers (see Trainer.invalidate_inputs)
# The captured step replayed as one linear hipGraph per stream, tied by device-side flags (bist_amd/graphsplit.py) instead of through the
# runtime's multi-branch graph executor, which serialises independent branches.  0 = torch's CUDAGraph.replay().
SPLIT_GRAPH = os.environ.get("BIST_SPLIT_GRAPH", "1") != "0"
# tuning aid: which leaves of the backward pass go to the leaf chain (bit mask; ops.LEAF_MASK): 1 = the frame-grid products (output
# projection dW + video gradient of stage 1, value projection dX / dW), 2 = the fusion logits' weight gradient, 4 = the closing
# reductions in per-layer batches.  Chip-filling products beside the direction chains cost those chains more than they save (measured
# 10.0 vs 8.9 ms per step with mask 1); moving only the fusion logits' weight gradient (mask 2: seven 58 us launches on four workgroups) gains
# nothing measurable either (8.93-8.99 vs 8.90-8.96) although a critical-path model of the step puts them on the path -- the step has many
# near-critical paths (scripts/critical_path.py).  Default 0.
LEAF_MASK = int(os.environ.get("BIST_LEAF_MASK", "0"))
LEAF_OFFLOAD = LEAF_MASK != 0
# Several ranks: the matrices of the LAST layers (their backward runs first) are exchanged in this many buckets DURING the backward pass,
# each as soon as the step's streams have signalled that its gradients are final (bist_flag_signal into pinned memory, polled by the
# host); 0 = every all-reduce after the backward pass (the round-3 schedule).  The remaining matrices follow after the pass in
# EXCHANGE_CHUNKS pieces.
EXCHANGE_BUCKETS = int(os.environ.get("BIST_EXCHANGE_BUCKETS", "2"))
# One rank, replayed by the split executor: the same buckets are UPDATED during the backward pass -- Adam on a bucket as a background
# launch (ADAM_BG_BLOCKS workgroups) on the caption chain, which has the least backward work, once all chains have passed the bucket's
# mark -- so that the step's tail is the update of the remaining matrices only.  0 = the whole update at the tail.
ADAM_EARLY_BUCKETS = int(os.environ.get("BIST_ADAM_EARLY_BUCKETS", "2"))
ADAM_BG_BLOCKS = int(os.environ.get("BIST_ADAM_BG_BLOCKS", "256"))
EARLY_REDUCTIONS = os.environ.get("BIST_EARLY_REDUCTIONS", "1") != "0"      # ... together with the bias / LayerNorm-parameter reductions queued up to each bucket's mark
FLUSH_AT_0 = os.environ.get("BIST_FLUSH_AT_0", "1") != "0"      # ... and, reductions only, once more at the end of the layer stacks' backward (mark 0)
EXCHANGE_TIMEOUT_S = float(os.environ.get("BIST_EXCHANGE_TIMEOUT_S", "20"))       # a bucket whose flags do not arrive within this raises
EXCHANGE_CHUNKS = int(os.environ.get("BIST_EXCHANGE_CHUNKS", "4"))        # pieces of the flat gradient per step (multi-rank): all-reduce k+1 runs under Adam k


def _round(n: int) -> int:
    return (n + ALIGN - 1) // ALIGN * ALIGN


class Trainer:
    def __init__(self, model: torch.nn.Module, args, vocab_size: int, *, compute_dtype: torch.dtype = torch.bfloat16,
                 warmup: int = 4000, factor: float = 1.0, smoothing: float = 0.1, pad: int = 1,
                 betas=(0.9, 0.98), eps: float = 1e-9, process_group=None, use_graph: bool = False,
                 deferred_adam: Optional[bool] = None):
        self.model, self.args = model, args
        self.use_graph = use_graph
        self._graph = self._graph2 = self._graph_key = self._static_batch = self._static_terms = None
        self._split = self._split2 = None
        self._static_src = {}
        self.compute_dtype = compute_dtype
        self.warmup, self.factor, self.betas, self.eps = warmup, factor, betas, eps
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self._step = 0
        # one rank: the optimiser is part of backward() (and of its captured hipGraph); several ranks: the gradient exchange
        # comes first, Adam stays an eager launch per exchanged piece (step())
        self.exchanging = self.world > 1 or (os.environ.get("BIST_FORCE_EXCHANGE") == "1" and dist.is_available()
                                             and dist.is_initialized())          # (one-rank rehearsal aid, bench.py)
        self.adam_in_step = not self.exchanging and os.environ.get("BIST_ADAM_IN_STEP", "1") != "0"
        # Deferred optimiser (one rank, replayed step): the update of step t is applied at the HEAD of step t+1 -- piece by piece in
        # the order the forward pass first reads the parameters, on a stream of its own, beside that forward pass -- and by flush()
        # after the last step.  The tail of a step is then the backward pass and its closing reductions only.
        # Measured SLOWER at BASELINE configs[1] (12.05 vs 11.71 ms per step: the HBM-bound update competes with the input projection
        # and slows the forward pass's small launches more than it saves at the tail, where Adam already runs beside the closing
        # reductions), so opt-in.
        if deferred_adam is None:
            deferred_adam = os.environ.get("BIST_DEFERRED_ADAM", "0") != "0"
        self.deferred = bool(deferred_adam) and self.adam_in_step and use_graph
        self._pending_host = False

        params: List[torch.nn.Parameter] = []
        seen = set()
        for p in model.parameters():
            if id(p) not in seen:
                seen.add(id(p)); params.append(p)
        dev = params[0].device
        lut = model.query_embed[0].lut.weight
        attns = [m for m in model.modules() if isinstance(m, MultiHeadedAttention)]

        # ---- layout -------------------------------------------------------------------------------
        order: List[torch.nn.Parameter] = []
        placed = set()

        def place(p):
            if id(p) not in placed:
                placed.add(id(p)); order.append(p)
        for m in attns:                                   # region 0: packed attention biases first ...
            for j in range(3):
                place(m.linears[j].bias)
        for p in params:                                  # ... then every other 1-D parameter ...
            if p.dim() == 1:
                place(p)
        place(lut)                                        # ... and the shared embedding matrix
        n_prefix_params = len(order)
        # The matrices behind the prefix, grouped into PIECES in the order the forward pass first reads them: piece 0 = everything
        # outside the layer stacks (input projection, text encoders, generator), piece 1 + l = reasoning / caption / decoder layer l.
        # Within a piece the packed attention weights come first.  (The deferred optimiser updates piece by piece.)
        names = {}
        for nme, p in model.named_parameters():
            names.setdefault(id(p), nme)

        def piece_of(p) -> int:
            mt = re.match(r"mutlimodal_decoder\.(?:v_layers|c_layers|layers)\.(\d+)\.", names.get(id(p), ""))
            return 1 + int(mt.group(1)) if mt else 0
        units = []                                        # (piece, kind, sequence, [parameters placed back to back])
        for i, m in enumerate(attns):
            units.append((piece_of(m.linears[0].weight), 0, i, [m.linears[j].weight for j in range(3)]))
        packed_ids = {id(q) for u in units for q in u[3]}
        for i, p in enumerate(params):
            if id(p) not in placed and id(p) not in packed_ids:
                units.append((piece_of(p), 1, i, [p]))
        units.sort(key=lambda u: u[:3])
        first_of_piece = {}
        for piece, _, _, ps in units:
            first_of_piece.setdefault(piece, ps[0])
            for q in ps:
                place(q)

        offs, n = {}, 0
        packed_w, packed_b = {}, {}
        for idx, p in enumerate(order):
            offs[id(p)] = n
            is_pack_member = any(p is m.linears[j].bias or p is m.linears[j].weight for m in attns for j in (0, 1))
            # members 0 and 1 of a packed triple are NOT padded, so [W0;W1;W2] is contiguous
            n += p.numel() if is_pack_member else _round(p.numel())
            if idx == n_prefix_params - 1:
                self.n32 = n                               # end of the fp32-accumulated prefix
        self.numel = n
        starts = sorted(offs[id(q)] for q in first_of_piece.values())
        assert not starts or starts[0] == self.n32
        # (lo, hi) of the fp32-accumulated prefix, then of every piece, in flat elements
        self.pieces = [(0, self.n32)] + [(lo, hi) for lo, hi in zip(starts, starts[1:] + [n]) if hi > lo]
        # Overlapped exchange: bucket j = the matrices of layers >= cut_j (all three stacks), cuts descending -- final, in the backward
        # pass, when everything recorded behind the layer loop's mark `cut_j` has run (decoder.py, Fn.bucket_mark); flat ranges, in the
        # order they become final.  What is left (layers below the last cut, everything outside the stacks) is exchanged after the pass.
        piece_lo = {pc: offs[id(q)] for pc, q in first_of_piece.items()}
        n_layers = max(piece_lo) if piece_lo else 0
        self.buckets: List[Tuple[int, int, int]] = []          # (cut layer, lo, hi)
        self.early_adam = (self.adam_in_step and not self.deferred and use_graph and SPLIT_GRAPH and ADAM_EARLY_BUCKETS > 0 and not ADAM_CLEARS
                           and dev.type == "cuda" and Fn.CONCURRENT)
        nb = max(0, min(EXCHANGE_BUCKETS if self.exchanging else ADAM_EARLY_BUCKETS if self.early_adam else 0, n_layers - 1))
        hi_ = n
        cuts_env = os.environ.get("BIST_BUCKET_CUTS")          # tuning aid: explicit cut layers, descending ("3,1")
        cuts = [int(c) for c in cuts_env.split(",")] if (cuts_env and nb) else [n_layers - (j + 1) * n_layers // (nb + 1) for j in range(nb)]
        for cut in cuts:                                       # default, e.g. 6 layers, 2 buckets: cuts 4, 2
            lo_ = piece_lo.get(cut + 1)
            if lo_ is None or lo_ >= hi_ or cut < 0:
                break
            self.buckets.append((cut, lo_, hi_))
            hi_ = lo_
        self._bucket_end = hi_                                  # [n32, _bucket_end): exchanged after the backward pass
        for m in attns:                                   # a packed member must start 16-byte aligned
            for t in ("bias", "weight"):
                o0 = offs[id(getattr(m.linears[0], t))]
                nn_ = getattr(m.linears[0], t).numel()
                assert offs[id(getattr(m.linears[1], t))] == o0 + nn_ and offs[id(getattr(m.linears[2], t))] == o0 + 2 * nn_
                assert (o0 * 2) % 16 == 0 and (nn_ * 2) % 16 == 0, "attention width must keep packed views 16-byte aligned"

        # ---- buffers ------------------------------------------------------------------------------
        self.master = torch.zeros(n, device=dev, dtype=torch.float32)
        for p in order:
            o = offs[id(p)]
            self.master[o:o + p.numel()].copy_(p.detach().reshape(-1).float())
        self.flat_param = self.master if compute_dtype == torch.float32 else self.master.to(compute_dtype)
        self.flat_grad = torch.zeros(n, device=dev, dtype=compute_dtype)
        self.acc32 = torch.zeros(self.n32, device=dev, dtype=torch.float32)
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        # {lr, 1 - beta1^t, 1 - beta2^t, grad_scale} of the current step, derived ON THE DEVICE from the step counter (drop_ctr) by
        # bist_noam_hyper at the head of the step: no host buffer that a later step could rewrite while this one is still queued
        self.hyper = torch.zeros(8, device=dev, dtype=torch.float32)
        self.hyper[4] = 1.0                   # bist_adam_apply_dev: apply (the deferred form rewrites it at every head)
        self._dirty = False                   # the gradient buffer holds a gradient no optimiser step consumed (and cleared)
        self._skip_head_clear = False
        self.pending = torch.zeros(1, device=dev, dtype=torch.int64)     # deferred optimiser: the step whose update is still to be applied (0: none)
        self.deferred = self.deferred and dev.type == "cuda"
        # (the deferred optimiser's launches go to the caption / decoder stream, idle at the head of a step: a FOURTH stream in the captured
        # step made replays of small models crash now and then inside hipGraphLaunch -- three times on the last day of round 3, always with
        # a fourth stream in the graph, never with three)
        self._adam_stream = None
        if self.deferred:
            object.__setattr__(model, "_bist_trainer", weakref.ref(self))     # model.eval() / state_dict() flush the pending update

        for p in order:
            o, k = offs[id(p)], p.numel()
            p.data = self.flat_param[o:o + k].view(p.shape)
            p.grad = None
            p._grad_view = self.flat_grad[o:o + k].view(p.shape)       # GEMM backward accumulates here
            if o < self.n32:
                p._acc32 = self.acc32[o:o + k].view(p.shape)           # atomically reduced gradients go here
        for m in attns:                                                # packed projection operands as plain views
            d_out, d_in = m.linears[0].weight.shape
            ow, ob = offs[id(m.linears[0].weight)], offs[id(m.linears[0].bias)]
            pk = {}
            for idx in ((0, 1, 2), (1, 2)):
                r0, r1 = idx[0] * d_out, (idx[-1] + 1) * d_out
                w = self.flat_param[ow + r0 * d_in: ow + r1 * d_in].view(r1 - r0, d_in).requires_grad_(True)
                b = self.flat_param[ob + r0: ob + r1].requires_grad_(True)
                w._grad_view = self.flat_grad[ow + r0 * d_in: ow + r1 * d_in].view(r1 - r0, d_in)
                b._acc32 = self.acc32[ob + r0: ob + r1]
                pk[idx] = (w, b)
            m._pk = pk
        self.params = order
        # device-side step counter mixed into every dropout seed, so that a captured graph draws new masks
        # weight-gradient GEMMs on their own stream (autograd._WeightGradStream): measured SLOWER in a replayed hipGraph
        # (21.2 vs 18.3 ms/step, round 1: ~300 extra cross-stream edges cost more than the overlap gains), so opt-in
        self.wgrad_side_stream = os.environ.get("BIST_WGRAD_STREAM", "0") != "0"
        self._wgrad_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self.drop_ctr = torch.zeros(1, device=dev, dtype=torch.int64)
        ops.DROP_CTR = self.drop_ctr
        # ready flags of the buckets, in pinned host memory: 4 per bucket (one per stream of the step)
        self.early_adam = self.early_adam and bool(self.buckets)
        self._early_done = 0                                   # buckets updated inside the running backward pass
        self._early_armed = False                              # (set by backward(optimizer=True) around its pass)
        self._early_keep: list = []
        self.overlap = self.exchanging and bool(self.buckets) and dev.type == "cuda" and Fn.CONCURRENT
        self._flags = torch.zeros(4 * max(1, len(self.buckets)), dtype=torch.int64).pin_memory() if self.overlap else None
        self._flags_np = self._flags.numpy() if self.overlap else None
        self._flag_streams = [0] * len(self.buckets)           # streams that signalled bucket j in the last recorded / eager pass
        self.exchange_ready_s: List[float] = []                # last step: seconds from the step's launch to each bucket's flags (diagnostics)
        self._comm_stream = None
        self._main_stream = None
        self.criterion = LabelSmoothing(vocab_size, pad, smoothing)
        self.loss_compute = SimpleLossCompute(model.generator, model.ae_generator, self.criterion, opt=None, args=args)

    # NoamOpt.rate (optimize.py:28-34)
    def rate(self, step: Optional[int] = None) -> float:
        return parallel.noam_rate(self._step if step is None else step, self.args.d_model, self.factor, self.warmup)

    def forward_loss(self, batch):
        ft = self.model.forward(batch)
        terms, logp = self.loss_compute.terms(ft, batch)
        return Fn.sum_terms(terms.values()), terms

    def _adam_dev(self, lo: int, hi: int, max_blocks: int = 0) -> None:
        """Adam on flat elements [lo, hi) with this step's scalars read from ``self.hyper`` (device): capturable.  max_blocks > 0: as a
        background launch of that many workgroups (bist_adam_step_dev_bg)."""
        if hi <= lo:
            return
        work = None if self.compute_dtype == torch.float32 else self.flat_param
        gsz = self.flat_grad.element_size()
        check(lib.bist_adam_step_dev_bg(self.master.data_ptr() + 4 * lo, self.flat_grad.data_ptr() + gsz * lo, self.m.data_ptr() + 4 * lo,
                                        self.v.data_ptr() + 4 * lo, (work.data_ptr() + work.element_size() * lo) if work is not None else None,
                                        hi - lo, self.hyper.data_ptr(), self.betas[0], self.betas[1], self.eps,
                                        dtype_code(self.compute_dtype), dtype_code(self.compute_dtype), max_blocks, _stream()), "bist_adam_step_dev")

    def _adam_apply(self, lo: int, hi: int) -> None:
        """bist_adam_apply_dev on flat elements [lo, hi): Adam with ``self.hyper`` (skipped when nothing is pending) + clear of the gradient."""
        if hi <= lo:
            return
        work = None if self.compute_dtype == torch.float32 else self.flat_param
        gsz = self.flat_grad.element_size()
        check(lib.bist_adam_apply_dev(self.master.data_ptr() + 4 * lo, self.flat_grad.data_ptr() + gsz * lo, self.m.data_ptr() + 4 * lo,
                                      self.v.data_ptr() + 4 * lo, (work.data_ptr() + work.element_size() * lo) if work is not None else None,
                                      hi - lo, self.hyper.data_ptr(), self.betas[0], self.betas[1], self.eps,
                                      dtype_code(self.compute_dtype), dtype_code(self.compute_dtype), _stream()), "bist_adam_apply_dev")

    def _pending_hyper(self) -> None:
        check(lib.bist_noam_hyper_pending(self.pending.data_ptr(), self.hyper.data_ptr(), float(self.args.d_model), float(self.factor),
                                          float(self.warmup), self.betas[0], self.betas[1], 1.0, _stream()), "bist_noam_hyper_pending")

    def flush(self) -> None:
        """Deferred optimiser: apply the update the last step() left pending (a no-op otherwise).  Called by model.eval() and
        model.state_dict(); call it before reading parameter tensors directly after a step."""
        if not self.deferred or not self._pending_host:
            return
        self._pending_hyper()
        for lo, hi in self.pieces:
            self._adam_apply(lo, hi)
        self._pending_host = False
        ops.WEIGHTS_EPOCH += 1

    def _backward_deferred(self, batch):
        """One replayed step with the deferred optimiser.  Head: the scalars of the pending step, then -- on the caption / decoder
        stream, idle until the first caption layer -- Adam piece by piece (each also clears its gradients), an event after each piece; the forward pass waits for a
        piece just before it first reads it (Fn.param_gate, called by the model on the main stream).  Tail: backward, closing
        reductions, pending <- this step."""
        self._pending_hyper()
        main, ad = torch.cuda.current_stream(), Fn.side_stream(1)
        ad.wait_stream(main)
        events = []
        with torch.cuda.stream(ad):
            for lo, hi in self.pieces:
                self._adam_apply(lo, hi)
                ev = torch.cuda.Event()
                ev.record(ad)
                events.append(ev)
        waited = [0]

        def gate(k: int) -> None:            # the prefix and pieces 0 .. k are final (main stream only)
            upto = min(len(events), k + 2)
            while waited[0] < upto:
                main.wait_event(events[waited[0]])
                waited[0] += 1
        if not getattr(self.model, "_bist_param_gates", False):
            gate(len(events))                 # a model that does not announce its first reads waits for the whole update
        Fn.PARAM_GATE = gate
        try:
            terms = self._backward_open(batch, clear=False, after_forward=lambda: gate(len(events)))
        finally:
            Fn.PARAM_GATE = None
        self._backward_close()
        self.pending.copy_(self.drop_ctr)
        return terms

    def _flush_on_caption_chain(self) -> None:
        """The bias / LayerNorm-parameter reductions queued so far, on the caption chain behind a wait for the other chains (captured split
        step only): their operands were produced on those chains and stay referenced until the pass ends, so none is recycled meanwhile."""
        cap = Fn.fourth_stream()
        for st in Fn.step_streams(self._main_stream):
            if st.cuda_stream != cap.cuda_stream:
                cap.wait_stream(st)
        with torch.cuda.stream(cap):
            self._early_keep.extend(ops.COLSUM_QUEUE or [])
            self._early_keep.extend(ops.LNGRAD_QUEUE or [])
            ops.col_sum_flush()
            ops.lngrad_flush()

    def _bucket_ready(self, cut: int) -> None:
        """Backward-pass callback of the layer loop's mark `cut` (autograd thread): every launch of the backward pass of layers >= cut
        has been issued -- signal the bucket's flag on each stream of the step, behind those launches."""
        j = next(i for i, b_ in enumerate(self.buckets) if b_[0] == cut)
        streams = Fn.step_streams(self._main_stream)
        with torch.cuda.stream(self._main_stream):
            capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            # an eager pass: every side stream of the package may hold launches -- the main stream waits for them (free outside a
            # graph) and signals alone
            for st in streams[1:]:
                self._main_stream.wait_stream(st)
            streams = streams[:1]
        if len(streams) > 4:
            raise RuntimeError("bist_amd.Trainer: a captured step over more than four streams cannot signal its buckets (4 flags per wait)")
        for c, st in enumerate(streams):
            check(lib.bist_flag_signal(self._flags.data_ptr() + 8 * (4 * j + c), self.drop_ctr.data_ptr(), st.cuda_stream), "bist_flag_signal")     # (pinned: the host pointer is the device's)
        self._flag_streams[j] = len(streams)

    def _bucket_adam(self, cut: int) -> None:
        """One rank, captured step: backward-pass callback of the layer loop's mark `cut` (autograd thread).  Every launch of the backward
        pass of layers >= cut has been issued: the caption chain -- the one with the least backward work -- waits for the other chains at
        this point and updates the bucket as a background launch, beside the backward pass of the earlier layers (which reads none of the
        bucket's parameters)."""
        j = next((i for i, b_ in enumerate(self.buckets) if b_[0] == cut), None)
        cap = Fn.fourth_stream()
        if EARLY_REDUCTIONS or j is None:
            # (mark 0, FLUSH_AT_0: the layer stacks' backward is over -- what is left runs on the main stream: the video gradient's sum, the
            # input projection's and the text encoders' backward; the reductions queued so far leave the tail, the matrices stay with its update)
            self._flush_on_caption_chain()
        else:
            for st in Fn.step_streams(self._main_stream):
                if st.cuda_stream != cap.cuda_stream:
                    cap.wait_stream(st)
        if j is None:
            return
        with torch.cuda.stream(cap):
            STM.mark("early adam %d" % j) if STM.ENABLED else None
            self._adam_dev(self.buckets[j][1], self.buckets[j][2], ADAM_BG_BLOCKS)
        self._early_done = j + 1

    def _await_bucket(self, j: int) -> None:
        """Host: until every stream of the queued step has written this step's number into bucket j's flags."""
        n = self._flag_streams[j]
        if n == 0:
            raise RuntimeError("bist_amd.Trainer: bucket %d was never signalled (the model's layer loop does not call Fn.bucket_mark)" % j)
        fl, want = self._flags_np[4 * j: 4 * j + n], self._step
        t0 = time.monotonic()
        while (fl < want).any():
            if time.monotonic() - t0 > EXCHANGE_TIMEOUT_S:
                raise RuntimeError("bist_amd.Trainer: the gradients of bucket %d were not signalled within %.0f s (flags %s, step %d)"
                                   % (j, EXCHANGE_TIMEOUT_S, fl.tolist(), want))
            time.sleep(0)                  # (yields the interpreter lock: the feeder's thread keeps running)
        self.exchange_ready_s.append(time.monotonic() - self._launched_at)

    def _backward_open(self, batch, clear: bool = True, after_forward=None):
        """forward + backward up to the point where every gradient BEHIND the fp32-accumulated prefix (the big matrices, 98 %
        of the elements) is final; the bias / LayerNorm-parameter reductions stay queued for _backward_close()."""
        if clear:
            self.flat_grad.zero_()
        self.acc32.zero_()
        STM.mark("step head")
        self._early_done = 0
        early = self._early_armed and Fn.fourth_stream() is not None and torch.cuda.is_current_stream_capturing()
        flush0 = {0} if (FLUSH_AT_0 and EARLY_REDUCTIONS) else set()
        Fn.BUCKET_MARK = (({b_[0] for b_ in self.buckets}, self._bucket_ready) if self.overlap else
                          ({b_[0] for b_ in self.buckets} | flush0, self._bucket_adam) if early else None)
        try:
            loss, terms = self.forward_loss(batch)
        finally:
            Fn.BUCKET_MARK = None
        loss = STM.through(loss, "loss")
        if after_forward is not None:
            after_forward()
        ops.COLSUM_QUEUE = []                 # bias gradients: queued by LinearFn.backward, summed in a few launches by _backward_close
        ops.LNGRAD_QUEUE = []                 # LayerNorm gain/offset gradients: likewise
        wg = self._wgrad_stream if (self.wgrad_side_stream and loss.is_cuda) else None
        ops.WGRAD_STREAM = wg                 # weight-gradient GEMMs: off the critical path, on their own stream
        # the leaves of the backward pass (frame-grid weight / video gradients) on a chain of their own: only when the step is replayed by
        # the split executor (a fourth stream in a graph for the runtime's executor is not trusted: Fn.MAX_CAPTURE_STREAMS)
        ops.LEAF_STREAM = Fn.leaf_stream() if (LEAF_OFFLOAD and self.use_graph and SPLIT_GRAPH and loss.is_cuda and Fn.CONCURRENT) else None
        ops.LEAF_MASK = LEAF_MASK
        try:
            if self.overlap or early:
                self._main_stream = torch.cuda.current_stream()
            if self.overlap:
                self._flag_streams = [0] * len(self.buckets)
            loss.backward()
            STM.mark("backward issued (main)")
            if loss.is_cuda:
                Fn.join_side_streams()
            STM.mark("streams joined")
            if wg is not None:
                torch.cuda.current_stream().wait_stream(wg)
            for p in self.params:                # anything autograd still produced itself (views, fallbacks)
                if p.grad is not None:
                    p._grad_view.add_(p.grad)
                    p.grad = None
        except BaseException:
            ops.COLSUM_QUEUE = ops.LNGRAD_QUEUE = None
            raise
        finally:
            self._early_keep.clear()
            ops.WGRAD_STREAM = None
            ops.LEAF_STREAM = None
            ops.WGRAD_KEEP.clear()
            Fn.release_taken()
        return terms

    def _backward_close(self) -> None:
        """The deferred bias / LayerNorm-parameter reductions and the fold of the fp32 prefix into the gradient buffer."""
        try:
            ops.col_sum_flush()
            ops.lngrad_flush()
        finally:
            ops.COLSUM_QUEUE = None
            ops.LNGRAD_QUEUE = None
        check(lib.bist_add_f32_into(self.acc32.data_ptr(), self.flat_grad.data_ptr(), self.n32, dtype_code(self.compute_dtype),
                                    _stream()), "bist_add_f32_into")

    def backward(self, batch, optimizer: bool = False):
        """forward + backward; leaves the complete gradient in ``flat_grad``.  optimizer=True (single rank) also applies
        Adam with the scalars in ``self.hyper``: the big matrices on a side stream BESIDE the closing reductions, the
        prefix of biases / LayerNorm parameters after them."""
        if optimizer and self.deferred and self.flat_grad.is_cuda:
            return self._backward_deferred(batch)
        if optimizer:
            check(lib.bist_noam_hyper(self.drop_ctr.data_ptr(), self.hyper.data_ptr(), float(self.args.d_model), float(self.factor),
                                      float(self.warmup), self.betas[0], self.betas[1], 1.0, _stream()), "bist_noam_hyper")
        # one rank, optimiser in the step: Adam clears every gradient it consumes (bist_adam_apply_dev), so a REPLAYED step starts from a
        # clean buffer without a memset of its own (_graph_open clears it when something else left a gradient behind)
        self._early_armed = bool(optimizer and self.early_adam)
        try:
            terms = self._backward_open(batch, clear=not (optimizer and self._skip_head_clear))
        finally:
            self._early_armed = False
        big_end = self.buckets[self._early_done - 1][1] if self._early_done else self.numel      # (the buckets above were updated inside the pass)
        side = None
        if optimizer and self.flat_grad.is_cuda:
            main, side = torch.cuda.current_stream(), Fn.side_stream(0)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                (self._adam_apply if ADAM_CLEARS else self._adam_dev)(self.n32, big_end)
        self._backward_close()
        STM.mark("closing reductions done")
        if optimizer:
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
                STM.mark("adam (big) done")
                (self._adam_apply if ADAM_CLEARS else self._adam_dev)(0, self.n32)
                STM.mark("step end")
            else:
                self._adam_dev(0, self.numel)
        else:
            self._dirty = True
        return terms

    # ---- hipGraph path: forward + backward + gradient fold captured once per batch geometry ----------------
    _BATCH_FIELDS = ("query", "his", "cap", "trg", "trg_y", "fts", "query_mask", "query_mask2", "his_mask", "cap_mask", "temporal_mask",
                     "trg_mask", "trg_mean_mask", "ntokens", "qntokens")

    def _shape_key(self, batch):
        return tuple((f, tuple(getattr(batch, f).shape), getattr(batch, f).dtype) for f in self._BATCH_FIELDS
                     if getattr(batch, f, None) is not None)

    def _capture(self, batch, key):
        """Record the launches of forward+backward (+ Adam on one rank) into hipGraphs (the batch tensors become the graphs'
        static inputs).  Warm-up runs on a side stream first, as graph capture requires.  Several ranks: TWO graphs, the
        cut where the big matrices' gradients are final, so that their all-reduce starts under the closing reductions."""
        batch = self._own_copy(batch)             # the graphs read trainer-owned buffers, never the caller's (a feeder slot is rewritten in flight)
        graph = Fn.Graph(split=SPLIT_GRAPH and self.flat_grad.is_cuda)
        if graph.want_split and Fn._streams_ready():
            # the weight-gradient stream (BIST_WGRAD_STREAM=1) on a hardware queue of its own as well
            if self.wgrad_side_stream:
                self._wgrad_stream = Fn.leaf_stream()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self.backward(batch)              # warm-up passes leave the weights alone
        torch.cuda.current_stream().wait_stream(side)
        graph2 = None
        if self.exchanging:
            with Fn.capture_graph(graph):   # other threads (RCCL watchdog) may touch the runtime during capture
                terms = self._backward_open(batch)
            graph2 = Fn.Graph(split=False)
            with Fn.capture_graph(graph2):
                self._backward_close()
        else:
            self._skip_head_clear = self.adam_in_step and self.flat_grad.is_cuda and ADAM_CLEARS
            try:
                with Fn.capture_graph(graph):
                    terms = self.backward(batch, optimizer=self.adam_in_step)
            finally:
                self._skip_head_clear = False
        self._split = graph.split
        self._graph, self._graph2, self._graph_key, self._static_batch, self._static_terms = graph, graph2, key, batch, terms
        self._static_src = {}

    def _globally_normalised(self, batch):
        """The reference normalises each loss term by the token count of the WHOLE batch (optimize.py:48-50: ``/ batch.ntokens``,
        ``/ batch.qntokens``).  With the batch sharded over ranks, every rank divides its own sum by the GLOBAL counts (one
        all-reduce of two integers, asynchronous on the device) and the gradients are SUMMED: exactly the reference's update,
        also when the ranks hold different numbers of tokens.  (A mean of per-rank-normalised gradients would weight tokens of
        short shards more.)"""
        tok = torch.stack([batch.ntokens.reshape(()), batch.qntokens.reshape(())]).to(torch.int64)
        dist.all_reduce(tok, op=dist.ReduceOp.SUM, group=self.pg)
        gb = copy.copy(batch)
        gb.ntokens, gb.qntokens = tok[0], tok[1]
        return gb

    def _own_copy(self, batch):
        """A shallow copy of the batch whose tensor fields are clones owned by the trainer."""
        own = copy.copy(batch)
        for f in self._BATCH_FIELDS:
            v = getattr(batch, f, None)
            if v is not None:
                setattr(own, f, v.clone())
        return own

    def invalidate_inputs(self) -> None:
        """Forget which source tensors are resident in the replayed step's static batch: the next step copies every field again.
        The residency test of `_graph_open` sees torch's in-place writes (`_version`) and DeviceFeeder refills (`_bist_generation`); a
        producer that refills a REUSED batch tensor behind both -- a kernel of this library writing through a raw pointer, another
        process through IPC memory -- must call this (or hand over a new tensor object) before the step, or set
        BIST_ALWAYS_COPY_INPUTS=1."""
        self._static_src.clear()

    def _graph_open(self, batch):
        if ALWAYS_COPY_INPUTS:
            self._static_src.clear()
        key = self._shape_key(batch)
        if self._graph is None or key != self._graph_key:
            self._capture(batch, key)
        for f in self._BATCH_FIELDS:
            src = getattr(batch, f, None)
            if src is None:
                continue
            # a source the trainer has already copied and that was not written since (same tensor object, storage, in-place
            # version and feeder generation) is resident in the static buffers: a loop over one resident batch (bench.py) moves
            # no input bytes, a new batch costs one device-to-device copy per field
            # (the entry keeps the source tensor itself, so its identity cannot be recycled by a later tensor)
            last = self._static_src.get(f)
            sig = (src.data_ptr(), src._version, getattr(src, "_bist_generation", 0))
            if last is None or last[0] is not src or last[1] != sig:
                getattr(self._static_batch, f).copy_(src, non_blocking=True)
                self._static_src[f] = (src, sig)
        if self._dirty and self.adam_in_step:        # warm-up passes / a backward() without optimiser left a gradient behind
            self.flat_grad.zero_()
        self._dirty = False
        self._graph.replay()
        return self._static_terms

    def _exchange_stream(self):
        """An idle stream to issue the early buckets' all-reduces from (RCCL makes its own stream wait for the issuing one: from the
        stream the step runs on that would be the end of the step)."""
        if self._comm_stream is None:
            mine = {s_.cuda_stream for s_ in Fn.step_streams(torch.cuda.current_stream())}
            for _ in range(64):
                st = torch.cuda.Stream()
                if st.cuda_stream not in mine:
                    break
            self._comm_stream = st
        return self._comm_stream

    def step(self, batch) -> Dict[str, torch.Tensor]:
        """One optimiser step; returns the (detached, device-side) loss terms."""
        self._step += 1
        self.drop_ctr.fill_(self._step)
        ops.WEIGHTS_EPOCH += 1               # the parameters change behind autograd's back: derived operands (packed weights) are stale
        if self.adam_in_step:                # the optimiser scalars are derived from drop_ctr inside backward() (bist_noam_hyper)
            terms = self._graph_open(batch) if self.use_graph else self.backward(batch, optimizer=True)
            self._pending_host = self.deferred
            return {k: v.detach() for k, v in terms.items()}
        work = None if self.compute_dtype == torch.float32 else self.flat_param
        gsz = self.flat_grad.element_size()
        rate = self.rate()

        def adam(lo: int, hi: int, grad_scale: float):
            check(lib.bist_adam_step(self.master.data_ptr() + 4 * lo, self.flat_grad.data_ptr() + gsz * lo, self.m.data_ptr() + 4 * lo,
                                     self.v.data_ptr() + 4 * lo, (work.data_ptr() + work.element_size() * lo) if work is not None else None,
                                     hi - lo, rate, self.betas[0], self.betas[1], self.eps, self._step, grad_scale,
                                     dtype_code(self.compute_dtype), dtype_code(self.compute_dtype), _stream()), "bist_adam_step")
        if not self.exchanging:              # one rank, optimiser outside the step (BIST_ADAM_IN_STEP=0)
            terms = self._graph_open(batch) if self.use_graph else self.backward(batch)
            adam(0, self.numel, 1.0)
            return {k: v.detach() for k, v in terms.items()}
        # Several ranks.  The gradients behind the fp32-accumulated prefix (the big matrices) are final before the closing
        # bias / LayerNorm reductions: their all-reduce -- EXCHANGE_CHUNKS large pieces issued back to back (xGMI is
        # point-to-point: a few large messages keep all 7 links busy) -- starts under those reductions, the prefix follows,
        # and Adam runs on a piece as soon as its own all-reduce has finished, i.e. under the next piece's (the compute
        # stream waits on the collective's event, never the host).
        batch = self._globally_normalised(batch)
        self._launched_at = time.monotonic()
        terms = self._graph_open(batch) if self.use_graph else self._backward_open(batch)
        big, works = [], []
        if self.overlap:
            # The step is queued (a replayed graph returns at once; an eager pass has issued its launches).  Per bucket: poll the flags
            # the step's streams write when the bucket's gradients are final, then issue its all-reduce from an idle stream -- RCCL orders a
            # collective behind the stream it is issued from, so it starts at once, under the rest of the backward pass.  The flags carry
            # THIS step's number: the collective is behind this step's writes of its gradients, themselves behind the previous step's Adam.
            self.exchange_ready_s = []
            with torch.cuda.stream(self._exchange_stream()):
                for j, (_cut, lo, hi) in enumerate(self.buckets):
                    self._await_bucket(j)
                    big.append((lo, hi))
                    works += parallel.exchange_gradients_async(self.flat_grad, [(lo, hi)], self.pg)
        rest = [(self.n32 + lo, self.n32 + hi) for lo, hi in parallel.chunk_bounds((self._bucket_end if self.overlap else self.numel) - self.n32,
                                                                                      EXCHANGE_CHUNKS, ALIGN)]
        big += rest
        works += parallel.exchange_gradients_async(self.flat_grad, rest, self.pg)
        main = side = None
        if self.flat_grad.is_cuda and Fn.CONCURRENT:
            # the big pieces' updates on a side stream, each behind its own all-reduce: beside the closing reductions (as in the one-rank
            # step) instead of after them -- with a fast exchange (few ranks) the step no longer pays Adam's 0.8 ms in line
            main, side = torch.cuda.current_stream(), Fn.side_stream(0)
            side.wait_stream(main)           # (the fork: before the closing reductions are queued on the main stream)
        if self.use_graph:
            self._graph2.replay()
        else:
            self._backward_close()
        prefix = parallel.exchange_gradients_async(self.flat_grad, [(0, self.n32)], self.pg)
        if side is not None:
            with torch.cuda.stream(side):
                for (lo, hi), wk in zip(big, works):
                    wk.wait()                # (RCCL: orders the CURRENT stream behind the collective, the host does not block; gloo blocks the host)
                    adam(lo, hi, 1.0)
        if side is None:
            for (lo, hi), wk in zip(big, works):
                wk.wait()
                adam(lo, hi, 1.0)        # every rank divided by the GLOBAL token counts: the summed gradient is the reference's
        prefix[0].wait()
        adam(0, self.n32, 1.0)
        if side is not None:
            main.wait_stream(side)
        return {k: v.detach() for k, v in terms.items()}


def run_epoch(data, loader, vocab, epoch, model, loss_compute, eval=False, gen_valid_indices=None, report=None):
    """The reference's epoch loop (train.py:21-52) with its signature: for every batch ``model.forward`` then ``loss_compute``
    (a ``SimpleLossCompute``; with ``opt`` set it also runs backward and the optimiser step), accumulating the un-normalised
    losses and the token counts.  Batches must already live on the device (the reference calls ``batch.move_to_cuda()`` here,
    train.py:30; bist_amd.data.Batch objects built from device tensors need nothing).  ``report(j, losses, batch)`` replaces
    the reference's print / CSV logging (train.py:38-47).  Returns the same three per-token averages, without the
    reference's floor division on the spatial term (train.py:51, a logging slip)."""
    total_tokens = total_qtokens = 0
    total_loss = total_t = total_s = 0.0
    for j, batch in enumerate(loader):
        if hasattr(batch, "move_to_cuda") and not batch.query.is_cuda:
            batch.move_to_cuda()
        out = model.forward(batch)
        losses = loss_compute(out, batch)
        total_loss = total_loss + losses["out"]
        total_t = total_t + losses["temporal_ae"]
        total_s = total_s + losses["spatial_ae"]
        total_tokens = total_tokens + batch.ntokens
        total_qtokens = total_qtokens + batch.qntokens
        if report is not None and not eval:
            report(j, losses, batch)
    return {"out": total_loss / total_tokens.float(), "temporal_ae": total_t / total_qtokens.float(),
            "spatial_ae": total_s / total_qtokens.float()}
