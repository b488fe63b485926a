"""Label-smoothed KL loss on the HIP kernels (reference: model/label_smoothing.py)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as Fn


class LabelSmoothing(nn.Module):
    """KLDiv(sum) against confidence on the target, smoothing/(size-2) elsewhere, nothing on the
    padding column or on padded rows (reference: label_smoothing.py:9-30).  Returns a device
    scalar [1]; the smoothed target distribution is never materialised (its gradient, -target, is
    produced directly by the backward kernel)."""

    def __init__(self, size, padding_idx, smoothing=0.0):
        super().__init__()
        self.padding_idx = padding_idx
        self.confidence = 1.0 - smoothing
        self.smoothing = smoothing
        self.size = size
        self.true_dist = None

    def loss(self, x: torch.Tensor, target: torch.Tensor, denom=None) -> torch.Tensor:
        """sum over rows / denom (device int64 scalar, optional), differentiable."""
        assert x.size(1) == self.size
        return Fn.label_smoothing_loss(x, target, denom, self.smoothing, self.padding_idx)

    def forward(self, x, target):
        return self.loss(x, target, None)
