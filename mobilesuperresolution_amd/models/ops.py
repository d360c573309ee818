"""BinaryConv2d / rounding of the NAS supernet (reference models/ops.py:7-43) on the hot path.

`BinaryConv2d` is a parameter holder with the reference's key (`weight`, shape (C,1,1,1), init U(0.5,1)).
Its forward value is a per-channel 0/1 mask with a straight-through gradient to `weight`
(weight - (weight.detach() - rounding(weight.detach()))); the multiplication itself is fused into the
NAS block kernels (csrc/nas_block.h), so the module exposes `effective()` instead of running a conv."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.init as init

__all__ = ["BinaryConv2d", "rounding"]


def rounding(weight: torch.Tensor, least_channel: int = 8) -> torch.Tensor:
    """mask = (w >= 0.5); if fewer than `least_channel` survive: mask = (w >= k-th largest w) (ties keep all);
    least_channel = 0 disables the fallback (reference models/ops.py:33-43)."""
    w = (weight >= 0.5).float()
    if least_channel > 0:
        # w >= k-th largest  <=>  fewer than k entries are strictly larger (ties at the k-th value all stay): two small
        # launches instead of topk's select + sort; and the same selection as the reference's `if torch.sum(w) >=
        # least_channel`, without the host sync it implies
        flat = weight.reshape(-1)
        topk = ((flat.unsqueeze(0) > flat.unsqueeze(1)).sum(1) < least_channel).float().view_as(weight)
        return torch.where(torch.sum(w) >= least_channel, w, topk)
    return w


class BinaryConv2d(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, dilation=1, groups=1,
                 bias=False, least_channel=8):
        super().__init__()
        if not (kernel_size == 1 and groups == in_channels == out_channels and not bias):
            raise NotImplementedError("hot path supports the depthwise 1x1 mask form of BinaryConv2d only")
        self.in_channels = self.out_channels = in_channels
        self.least_channel = least_channel
        self.weight = nn.Parameter(torch.empty(out_channels, 1, 1, 1))
        init.uniform_(self.weight, 0.5, 1)

    def init(self, value=0.5):
        init.constant_(self.weight, value)

    def effective(self, mask: torch.Tensor = None) -> torch.Tensor:
        """(C,) tensor: value = 0/1 mask, gradient = identity to `weight` (models/ops.py:18-24).  `mask`: rounding(weight)
        if the caller already has it."""
        w = self.weight.detach()
        if mask is None:
            mask = rounding(w, self.least_channel)
        return (self.weight - (w - mask)).reshape(-1)

    def forward(self, x, y=None):
        raise NotImplementedError("BinaryConv2d is fused into the NAS block kernels; use effective()")
