"""Training step of the hot path: forward, losses, backward, gradient exchange, Adam -- the body of the
reference's ``run_epoch`` loop (train.py:29-37) + ``SimpleLossCompute`` (model/optimize.py:46-94) +
``NoamOpt.step`` (optimize.py:19-26) for one batch.

MI355X layout: all parameters live in ONE flat buffer in the compute dtype (the tensors the kernels
read; each ``nn.Parameter`` is a view into it), all gradients in ONE flat buffer of the same dtype
(each ``.grad`` is a view, autograd accumulates in place), and the fp32 master weights and Adam
moments in three more flat buffers.  A step is therefore one memset, one (optional) RCCL all-reduce
over xGMI of the whole gradient buffer and one fused Adam kernel that also refreshes the low-precision
copy -- no per-parameter launches, no NCCL-style per-tensor collectives.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import functional as Fn
from ._lib import check, lib
from .model.label_smoothing import LabelSmoothing
from .model.optimize import SimpleLossCompute
from .ops import _stream, dtype_code

ALIGN = 64      # elements; keeps every parameter view 16-byte aligned for the LDS-DMA GEMM path


class Trainer:
    def __init__(self, model: torch.nn.Module, args, vocab_size: int, *, compute_dtype: torch.dtype = torch.bfloat16,
                 warmup: int = 4000, factor: float = 1.0, smoothing: float = 0.1, pad: int = 1,
                 betas=(0.9, 0.98), eps: float = 1e-9, process_group=None):
        self.model, self.args = model, args
        self.compute_dtype = compute_dtype
        self.warmup, self.factor, self.betas, self.eps = warmup, factor, betas, eps
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self._step = 0
        params, seen = [], set()
        for p in model.parameters():
            if id(p) not in seen:
                seen.add(id(p)); params.append(p)
        dev = params[0].device
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = n
        self.master = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, o in zip(params, offs):
            self.master[o:o + p.numel()].copy_(p.detach().reshape(-1).float())
        self.flat_param = self.master if compute_dtype == torch.float32 else self.master.to(compute_dtype)
        self.flat_grad = torch.zeros(n, device=dev, dtype=compute_dtype)
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, o in zip(params, offs):
            p.data = self.flat_param[o:o + p.numel()].view(p.shape)
            p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
        for b in model.buffers():          # the PE table stays fp32
            pass
        self.params = params
        self.criterion = LabelSmoothing(vocab_size, pad, smoothing)
        self.loss_compute = SimpleLossCompute(model.generator, model.ae_generator, self.criterion, opt=None, args=args)

    # NoamOpt.rate (optimize.py:28-34)
    def rate(self, step: Optional[int] = None) -> float:
        step = self._step if step is None else step
        return self.factor * (self.args.d_model ** -0.5 * min(step ** -0.5, step * self.warmup ** -1.5))

    def forward_loss(self, batch):
        ft = self.model.forward(batch)
        terms, logp = self.loss_compute.terms(ft, batch)
        loss = None
        for t in terms.values():
            loss = t if loss is None else loss + t
        return loss, terms

    def step(self, batch) -> Dict[str, torch.Tensor]:
        """One optimiser step; returns the (detached, device-side) loss terms."""
        self.flat_grad.zero_()
        loss, terms = self.forward_loss(batch)
        loss.backward()
        if self.world > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
        self._step += 1
        work = None if self.compute_dtype == torch.float32 else self.flat_param
        check(lib.bist_adam_step(self.master.data_ptr(), self.flat_grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                 work.data_ptr() if work is not None else None, self.numel, self.rate(), self.betas[0],
                                 self.betas[1], self.eps, self._step, 1.0 / self.world, dtype_code(self.compute_dtype),
                                 dtype_code(self.compute_dtype), _stream()), "bist_adam_step")
        return {k: v.detach() for k, v in terms.items()}
