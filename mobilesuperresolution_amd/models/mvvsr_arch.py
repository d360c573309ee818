"""MotionVectorVSR on the MI355X hot path (reference: models/mvvsr_arch.py:11-109; the trainer's 'basic_mv' model,
train_video_superresolution.py:251 constructs `MotionVectorVSR(num_feat=20, num_block=8, spynet_path=...)`).

Same constructor, forward signature `forward(x_, height, weight)` and state_dict keys (`backward_trunk.main.*`,
`forward_trunk.main.*`, `fusion.*`, `upconv1.*`, `upconv2.*`, `conv_hr.*`, `conv_last.*`).  The hot path -- the two
recurrent propagation loops: flow_warp -> concat -> ConvResidualBlocks trunk, mvvsr_arch.py:72-93 -- runs in HIP
(csrc/conv3x3.h, csrc/flow_warp.h).  The reconstruction behind it (1x1 fusion, ConvTranspose2d x4, bilinear resize and
base add, :95-105) is SURVEY row K14, out of scope, and stays in ATen.  SPyNet is out of scope as well: the reference
constructs it and never calls it in this model (flows are the motion vectors in channels 3..4 of the input, :63-67);
`spynet.*` keys of a reference checkpoint are accepted and ignored."""
from __future__ import annotations

import torch
from torch import nn as nn
from torch.nn import functional as F

from .basicvsr_arch import ConvResidualBlocks, propagate
from .spynet_arch import flow_warp

__all__ = ["MotionVectorVSR"]


class _IgnoresSpynetKeys:
    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for k in [k for k in state_dict if k.startswith(prefix + "spynet.")]:
            del state_dict[k]                        # out-of-scope optical-flow prior of the reference checkpoint
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class MotionVectorVSR(_IgnoresSpynetKeys, nn.Module):

    def __init__(self, num_feat=64, num_block=15, spynet_path=None, hot_dtype=None):
        super().__init__()
        self.num_feat = num_feat
        self.scale = 4
        # propagation (hot path)
        self.backward_trunk = ConvResidualBlocks(num_feat + 3, num_feat, num_block, hot_dtype=hot_dtype)
        self.forward_trunk = ConvResidualBlocks(num_feat + 3, num_feat, num_block, hot_dtype=hot_dtype)
        # reconstruction (ATen; same layers, same construction order as mvvsr_arch.py:33-41)
        self.fusion = nn.Conv2d(num_feat * 2, num_feat * 2, 1, 1, 0, bias=True)
        self.upconv1 = nn.Conv2d(num_feat, num_feat * 4, 3, 1, 1, bias=True)
        self.upconv2 = nn.Conv2d(num_feat, num_feat * 4, 3, 1, 1, bias=True)
        self.conv_hr = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_last = nn.ConvTranspose2d(num_feat * 2, 3, 5, stride=self.scale)
        self.pixel_shuffle = nn.PixelShuffle(2)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)

    def forward(self, x_, height=1080, weight=1920):
        """x_: (b, n, 5, h, w) = RGB frames + motion vectors (mvvsr_arch.py:57-67) -> (b, n, 3, height, weight)"""
        x = x_[:, :, :3, :, :]
        mv = x_[:, :, 3:, :, :]
        flows_forward = mv[:, 1:, :, :]
        flows_backward = flows_forward * (-1)
        b, n, _, h, w = x.size()
        feat_b, feat_f = propagate(x, flows_forward, flows_backward, self.backward_trunk, self.forward_trunk, flow_warp,
                                   num_feat=self.num_feat)
        out_l = []
        for i in range(n):
            out = torch.cat([feat_b[i], feat_f[i]], dim=1)
            out = self.lrelu(self.fusion(out))
            out = self.conv_last(out)
            out = F.interpolate(out, size=(height, weight), mode='bilinear')
            base = F.interpolate(x[:, i], size=(height, weight), mode='bilinear', align_corners=False)
            out_l.append(out + base)
        return torch.stack(out_l, dim=1)
