"""Label-smoothed KL loss on the HIP kernels (reference: model/label_smoothing.py)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class LabelSmoothing(nn.Module):
    """KLDiv(sum) against confidence on the target, smoothing/(size-2) elsewhere, nothing on the
    padding column or on padded rows (reference: label_smoothing.py:9-30).  Returns a device
    scalar [1]; the smoothed target distribution is never materialised."""

    def __init__(self, size, padding_idx, smoothing=0.0):
        super().__init__()
        self.padding_idx = padding_idx
        self.confidence = 1.0 - smoothing
        self.smoothing = smoothing
        self.size = size
        self.true_dist = None

    def row_losses(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        assert x.size(1) == self.size
        return ops.label_smoothing_rows(x, target, self.smoothing, self.padding_idx)

    def forward(self, x, target):
        return ops.sum_div(self.row_losses(x, target))
