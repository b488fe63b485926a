"""Noam learning-rate wrapper and the loss/step driver (reference: model/optimize.py)."""
from __future__ import annotations

import os

import torch

from .. import functional as Fn
from .. import ops

AE_GROUPED = os.environ.get("BIST_AE_GROUPED", "1") != "0"      # tuning aid: 0 = one vocabulary product / log-softmax / loss chain per auto-encoder head


class NoamOpt:
    """lr = factor * d_model^-0.5 * min(step^-0.5, step * warmup^-1.5) (reference: optimize.py:9-34)."""

    def __init__(self, model_size, factor, warmup, optimizer):
        self.optimizer = optimizer
        self._step = 0
        self.warmup = warmup
        self.factor = factor
        self.model_size = model_size
        self._rate = 0

    def rate(self, step=None):
        step = self._step if step is None else step
        return self.factor * (self.model_size ** (-0.5) * min(step ** (-0.5), step * self.warmup ** (-1.5)))

    def step(self):
        self._step += 1
        rate = self.rate()
        for group in self.optimizer.param_groups:
            group["lr"] = rate
        self._rate = rate
        self.optimizer.step()


class SimpleLossCompute:
    """Response loss + query auto-encoder losses, backward and optimiser step
    (reference: optimize.py:36-94).  Every term is normalised on the device (no host sync)."""

    def __init__(self, generator, ae_generator, criterion, opt=None, l=1.0, args=None):
        self.generator = generator
        self.ae_generator = ae_generator
        self.criterion = criterion
        self.opt = opt
        self.l = l
        self.args = args

    def terms(self, ft, batch):
        """dict name -> device scalar [1], already divided by its token count (optimize.py:50-82)."""
        a = self.args
        keys = []
        if a.auto_encoder:
            if a.nb_cenc_blocks > 0:
                keys.append(("cap_ae", "cap_ft"))
            if a.nb_venc_blocks > 0 and a.enc_st_combine == "none":
                if a.s2t:
                    keys.append(("temporal_ae", "temporal_ft"))
                if a.t2s:
                    keys.append(("spatial_ae", "spatial_ft"))
        # (The auto-encoder heads stay on this stream, behind the response head.  They were tried on the idle s2t stream -- 9.84 -> 9.78 ms per
        # step -- but they share the embedding matrix with the response head's vocabulary product: the two weight-gradient products
        # then accumulate into ONE gradient region from two streams, a read-modify-write race that showed up as a rare mismatch of the
        # embedding after a few steps.)
        ae = {}
        q = batch.query.reshape(-1) if keys else None
        out = self.generator(ft, batch, a)
        V = out.size(-1)
        t = {"out": self.criterion.loss(out.reshape(-1, V), batch.trg_y.reshape(-1), batch.ntokens.reshape(1))}
        grouped = self._ae_grouped(ft, batch, keys, q) if (keys and AE_GROUPED) else None
        if grouped is not None:
            ae = grouped
        else:
            for name, key in keys:
                lp = self.ae_generator(ft, batch, a, key)
                ae[name] = self.criterion.loss(lp.reshape(-1, lp.size(-1)), q, batch.qntokens.reshape(1))
        t.update(ae)
        return t, out

    def _ae_grouped(self, ft, batch, keys, q):
        """The auto-encoder heads as ONE chain (optimize.py:66-82 runs one per output): their inputs stacked (one launch), one vocabulary
        product against the shared embedding, one pass from the logits to the row losses and one to the logits' gradient
        (bist_xent_smooth_*: log-softmax and label smoothing fused, the gradient written in the product's operand dtype) -- 4 launches
        forward and 3 backward where three separate heads take 12 and 15.  None when the heads do not share one projection / shape."""
        gen, crit = self.ae_generator, self.criterion
        if not (getattr(gen, "shared_W", False) and hasattr(crit, "smoothing")):
            return None
        xs = []
        for _, key in keys:
            xs.append(Fn.fan_take(ft, key))
        x0 = xs[0]
        if not (x0.is_cuda and all(x.shape == x0.shape and x.dtype == x0.dtype for x in xs) and 1 <= len(xs) <= 4
                and x0.dtype in (torch.bfloat16, torch.float32) and (x0.numel() * x0.element_size()) % 16 == 0
                and q.numel() * x0.shape[-1] == x0.numel()):
            return None
        G = len(xs)
        logits = Fn.linear(Fn.stack_rows(xs), gen.proj, None, out_dtype=torch.float32)                 # [G * M, V]
        if crit.size != logits.shape[-1]:
            return None
        losses = Fn.xent_smooth_losses(logits, q, batch.qntokens.reshape(1), crit.smoothing, crit.padding_idx, G, x0.dtype)
        return {name: losses[g] for g, (name, _) in enumerate(keys)}

    def __call__(self, ft, batch):
        t, _ = self.terms(ft, batch)
        loss = Fn.sum_terms(t.values())
        if self.opt is not None:
            loss.backward()
            self.opt.step()
            self.opt.optimizer.zero_grad()
        norm, qn = batch.ntokens.float(), batch.qntokens.float()
        zero = torch.zeros((), device=norm.device)
        return {"out": t["out"].detach()[0] * norm,
                "temporal_ae": t["temporal_ae"].detach()[0] * qn if "temporal_ae" in t else zero,
                "spatial_ae": t["spatial_ae"].detach()[0] * qn if "spatial_ae" in t else zero}
