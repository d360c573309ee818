"""BasicVSR_origin on the MI355X hot path (reference: models/basicvsr_arch_origin.py:10-95).

Same constructor, state_dict keys and `forward(x, height, weight)`; the propagation loops (:61-82) run in HIP like
MotionVectorVSR's, and the two PixelShuffle(2) stages of the upsampler (:37,87-88) go through the standalone HIP
shuffle (csrc/pixel_shuffle.h, bit-exact).  The upsampler's convolutions (upconv1/2, conv_hr, conv_last) and the
bilinear base are plain library convolutions in the reference and stay in ATen.

Flows: `get_flow` (:42-51) runs SpyNet on every adjacent frame pair in both directions, as the reference does; SpyNet's 7x7
convolutions are MFMA kernels (models/spynet_arch.py, csrc/spynet_conv.h; inference only -- the reference's trainer keeps SPyNet
out of the optimizer).  Flows may also be GIVEN: `forward(x, height, weight, flows=(flows_forward, flows_backward))` with
(b, n-1, 2, h, w) tensors, or -- as MotionVectorVSR takes them -- as motion-vector channels 3..4 of a 5-channel input."""
from __future__ import annotations

import torch
from torch import nn as nn
from torch.nn import functional as F

from .. import _lib as L
from .basicvsr_arch import ConvResidualBlocks, propagate
from .spynet_arch import SpyNet, flow_warp

__all__ = ["BasicVSR_origin", "pixel_shuffle"]


class _PixelShuffle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r):
        n, c, h, w = x.shape
        if c % (r * r):
            raise ValueError(f"pixel_shuffle: {c} channels not divisible by {r * r}")
        out = torch.empty((n, c // (r * r), h * r, w * r), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            L.launch("sr_pixel_shuffle", L.lib().sr_pixel_shuffle, x.data_ptr(), out.data_ptr(), n, c // (r * r), h, w, r, 0,
                     L.stream_ptr(x.device))
        ctx.r = r
        return out

    @staticmethod
    def backward(ctx, g):
        r = ctx.r
        g = g.contiguous().float()
        n, c, hr, wr = g.shape
        dx = torch.empty((n, c * r * r, hr // r, wr // r), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            L.launch("sr_pixel_shuffle", L.lib().sr_pixel_shuffle, g.data_ptr(), dx.data_ptr(), n, c, hr // r, wr // r, r, 1,
                     L.stream_ptr(g.device))
        return dx, None


def pixel_shuffle(x: torch.Tensor, r: int) -> torch.Tensor:
    """nn.PixelShuffle(r) on the hot path (NCHW fp32, bit-exact); no CPU fallback"""
    if not x.is_cuda:
        raise L.HotpathError("pixel_shuffle (MI355X hot path) needs CUDA/HIP tensors; there is no CPU fallback")
    return _PixelShuffle.apply(x.contiguous().float(), int(r))


class BasicVSR_origin(nn.Module):

    def __init__(self, num_feat=64, num_block=15, spynet_path=None, hot_dtype=None):
        super().__init__()
        self.num_feat = num_feat
        self.spynet = SpyNet(spynet_path)            # alignment (basicvsr_arch_origin.py:25)
        self.scale = 4
        self.backward_trunk = ConvResidualBlocks(num_feat + 3, num_feat, num_block, hot_dtype=hot_dtype)
        self.forward_trunk = ConvResidualBlocks(num_feat + 3, num_feat, num_block, hot_dtype=hot_dtype)
        # reconstruction: same layers, same construction order as basicvsr_arch_origin.py:30-35
        self.fusion = nn.Conv2d(num_feat * 2, num_feat, 1, 1, 0, bias=True)
        self.upconv1 = nn.Conv2d(num_feat, num_feat * 4, 3, 1, 1, bias=True)
        self.upconv2 = nn.Conv2d(num_feat, 64 * 4, 3, 1, 1, bias=True)
        self.conv_hr = nn.Conv2d(64, 64, 3, 1, 1)
        self.conv_last = nn.Conv2d(64, 3, 3, 1, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)

    def get_flow(self, x):
        """basicvsr_arch_origin.py:42-51"""
        b, n, c, h, w = x.size()
        x_1 = x[:, :-1, :, :, :].reshape(-1, c, h, w)
        x_2 = x[:, 1:, :, :, :].reshape(-1, c, h, w)
        flows_backward = self.spynet(x_1, x_2).view(b, n - 1, 2, h, w)
        flows_forward = self.spynet(x_2, x_1).view(b, n - 1, 2, h, w)
        return flows_forward, flows_backward

    def forward(self, x, height, weight, flows=None):
        if flows is None:
            if x.shape[2] != 5:
                flows = self.get_flow(x)
            else:                                    # motion vectors ride in channels 3..4, as MotionVectorVSR takes them
                mv = x[:, :, 3:, :, :]
                x = x[:, :, :3, :, :]
                flows = (mv[:, 1:, :, :], mv[:, 1:, :, :] * (-1))
        flows_forward, flows_backward = flows
        b, n, _, h, w = x.size()
        feat_b, feat_f = propagate(x, flows_forward, flows_backward, self.backward_trunk, self.forward_trunk, flow_warp,
                                   num_feat=self.num_feat)
        out_l = []
        for i in range(n):
            out = torch.cat([feat_b[i], feat_f[i]], dim=1)
            out = self.lrelu(self.fusion(out))
            out = self.lrelu(pixel_shuffle(self.upconv1(out), 2))
            out = self.lrelu(pixel_shuffle(self.upconv2(out), 2))
            out = self.lrelu(self.conv_hr(out))
            out = self.conv_last(out)
            base = F.interpolate(x[:, i], scale_factor=4, mode='bilinear', align_corners=False)
            out = out + base
            out = F.interpolate(out, size=(height, weight), mode='bilinear')
            out_l.append(out)
        return torch.stack(out_l, dim=1)
