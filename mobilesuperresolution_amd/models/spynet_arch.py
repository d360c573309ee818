"""SpyNet and flow_warp on the MI355X hot path (reference: models/spynet_arch.py -- the vendored, BasicSR-derived copy of the
mmedit functions the BasicVSR variants call at basicvsr_arch.py:24,74,85 / basicvsr_arch_origin.py:25).

`flow_warp` (:98-129): bilinear / zeros / align_corners=True in HIP (csrc/flow_warp.h), forward and backward.

`SpyNet` (:28-96): same constructor, parameter names (`basic_module.{level}.basic_module.{0,2,4,6,8}.{weight,bias}`, buffers
`mean`, `std`) and `forward(ref, supp)`.  The five 7x7 convolutions of every pyramid level -- where SPyNet's arithmetic is, 2.6
GFLOP per 64 x 64 frame pair -- run as implicit-GEMM MFMA kernels (csrc/spynet_conv.h, bf16 operands, fp32 accumulation, one
launch per layer, NHWC between layers); the pyramid's glue between them (average pooling, the x2 bilinear flow upsampling, the
border-padded warp of the 3-channel support image, normalisation, concatenation: a few KB to a few MB per level) is torch
tensor plumbing.  INFERENCE ONLY: the reference's trainer keeps SPyNet out of the optimizer
(train_video_superresolution.py:160-163), so its parameters never change and no gradient the optimizer uses flows through it;
`forward` runs under no_grad and returns a flow without history (the reference would also populate `spynet.*.grad`, which
nothing reads)."""
from __future__ import annotations

import math

import torch
from torch import nn as nn
from torch.nn import functional as F

from .. import _lib as L
from .. import packing as P

__all__ = ["flow_warp", "SpyNet", "BasicModule"]


class _FlowWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flow):
        n, c, h, w = x.shape
        with torch.cuda.device(x.device):
            out = torch.empty_like(x)
            L.launch("sr_flow_warp_fwd", L.lib().sr_flow_warp_fwd, x.data_ptr(), flow.data_ptr(), out.data_ptr(), n, c, h, w,
                     L.stream_ptr())
        ctx.save_for_backward(x, flow)
        return out

    @staticmethod
    def backward(ctx, g):
        x, flow = ctx.saved_tensors
        n, c, h, w = x.shape
        with torch.cuda.device(x.device):
            g = g.contiguous().float()
            dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
            df = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
            L.launch("sr_flow_warp_bwd", L.lib().sr_flow_warp_bwd, x.data_ptr(), flow.data_ptr(), g.data_ptr(),
                     dx.data_ptr() if dx is not None else None, df.data_ptr() if df is not None else None, n, c, h, w,
                     L.stream_ptr())
        return dx, df


def flow_warp(x, flow, interp_mode="bilinear", padding_mode="zeros", align_corners=True):
    """x: (n, c, h, w); flow: (n, h, w, 2).  Only the reference's defaults are on the hot path."""
    if (interp_mode, padding_mode, align_corners) != ("bilinear", "zeros", True):
        raise NotImplementedError("hot path flow_warp supports bilinear / zeros / align_corners=True only")
    if not x.is_cuda:
        raise L.HotpathError("flow_warp (MI355X hot path) needs CUDA/HIP tensors; there is no CPU fallback")
    if x.device != flow.device:
        raise L.HotpathError(f"x on {x.device}, flow on {flow.device}")
    assert x.shape[-2:] == flow.shape[1:3]
    return _FlowWarp.apply(x.contiguous().float(), flow.contiguous().float())


def _border_warp(x, flow_nchw):
    """flow_warp(x, flow.permute(0, 2, 3, 1), padding_mode='border') of spynet_arch.py:73-75 on a 3-channel pyramid image (glue:
    torch's grid_sample, as the reference calls it)"""
    n, _, h, w = x.shape
    gy, gx = torch.meshgrid(torch.arange(0, h, device=x.device, dtype=x.dtype), torch.arange(0, w, device=x.device, dtype=x.dtype),
                            indexing="ij")
    vx = 2.0 * (gx + flow_nchw[:, 0]) / max(w - 1, 1) - 1.0
    vy = 2.0 * (gy + flow_nchw[:, 1]) / max(h - 1, 1) - 1.0
    return F.grid_sample(x, torch.stack((vx, vy), dim=3), mode="bilinear", padding_mode="border", align_corners=True)


_LAYERS = ((8, 32, True), (32, 64, True), (64, 32, True), (32, 16, True), (16, 2, False))


class BasicModule(nn.Module):
    """spynet_arch.py:10-25: Conv 7x7 8 -> 32 -> 64 -> 32 -> 16 -> 2, ReLU between; keys basic_module.{0,2,4,6,8}.*"""

    def __init__(self):
        super().__init__()
        self.basic_module = nn.Sequential(
            nn.Conv2d(8, 32, 7, 1, 3), nn.ReLU(inplace=False), nn.Conv2d(32, 64, 7, 1, 3), nn.ReLU(inplace=False),
            nn.Conv2d(64, 32, 7, 1, 3), nn.ReLU(inplace=False), nn.Conv2d(32, 16, 7, 1, 3), nn.ReLU(inplace=False),
            nn.Conv2d(16, 2, 7, 1, 3))
        self._packed = None

    def _pack(self, dev):
        """the five layers' weights as MFMA fragments (bf16) + padded biases; re-made when a parameter changed"""
        convs = [self.basic_module[i] for i in (0, 2, 4, 6, 8)]
        key = tuple((c.weight.data_ptr(), c.weight._version, c.bias._version) for c in convs) + (str(dev),)
        if self._packed is None or self._packed[0] != key:
            out = []
            for c, (cin, cout, _) in zip(convs, _LAYERS):
                tab = P.conv7_tables(cin, cout)
                src = torch.cat([c.weight.detach().float().reshape(-1), torch.zeros(1, device=dev)])
                idx = torch.from_numpy(tab["idx"]).to(dev)
                bias = torch.zeros(tab["mt"] * 32, device=dev)
                bias[:cout] = c.bias.detach().float()
                out.append((src.index_select(0, idx).to(torch.bfloat16).contiguous(), bias))
            self._packed = (key, out)
        return self._packed[1]

    def forward(self, tensor_input):
        """(N, 8, H, W) fp32 -> (N, 2, H, W) fp32"""
        if not tensor_input.is_cuda:
            raise L.HotpathError("SpyNet (MI355X hot path) needs CUDA/HIP tensors; there is no CPU fallback")
        n, _, h, w = tensor_input.shape
        dev = tensor_input.device
        packed = self._pack(dev)
        x = tensor_input.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
        with torch.cuda.device(dev):
            for (cin, cout, relu), (wp, bias) in zip(_LAYERS, packed):
                last = cout == 2
                y = torch.empty((n, h, w, cout), dtype=torch.float32 if last else torch.bfloat16, device=dev)
                L.launch("sr_conv7_fwd", L.lib().sr_conv7_fwd, x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr(), n, h, w, cin,
                         cout, 1 if relu else 0, 1 if last else 0, L.stream_ptr(dev))
                x = y
        return x.permute(0, 3, 1, 2)


class SpyNet(nn.Module):
    """spynet_arch.py:28-96"""

    def __init__(self, load_path=None):
        super().__init__()
        self.basic_module = nn.ModuleList([BasicModule() for _ in range(6)])
        if load_path:
            self.load_state_dict(torch.load(load_path, map_location="cpu", weights_only=True)["params"])
        self.register_buffer("mean", torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def preprocess(self, tensor_input):
        return (tensor_input - self.mean) / self.std

    def process(self, ref, supp):
        ref = [self.preprocess(ref)]
        supp = [self.preprocess(supp)]
        for _ in range(5):
            ref.insert(0, F.avg_pool2d(input=ref[0], kernel_size=2, stride=2, count_include_pad=False))
            supp.insert(0, F.avg_pool2d(input=supp[0], kernel_size=2, stride=2, count_include_pad=False))
        flow = ref[0].new_zeros([ref[0].size(0), 2, int(math.floor(ref[0].size(2) / 2.0)), int(math.floor(ref[0].size(3) / 2.0))])
        for level in range(len(ref)):
            up = F.interpolate(input=flow, scale_factor=2, mode="bilinear", align_corners=True) * 2.0
            if up.size(2) != ref[level].size(2):
                up = F.pad(input=up, pad=[0, 0, 0, 1], mode="replicate")
            if up.size(3) != ref[level].size(3):
                up = F.pad(input=up, pad=[0, 1, 0, 0], mode="replicate")
            flow = self.basic_module[level](torch.cat([ref[level], _border_warp(supp[level], up), up], 1)) + up
        return flow

    @torch.no_grad()
    def forward(self, ref, supp):
        assert ref.size() == supp.size()
        h, w = ref.size(2), ref.size(3)
        w_floor = math.floor(math.ceil(w / 32.0) * 32.0)
        h_floor = math.floor(math.ceil(h / 32.0) * 32.0)
        ref = F.interpolate(input=ref, size=(h_floor, w_floor), mode="bilinear", align_corners=False)
        supp = F.interpolate(input=supp, size=(h_floor, w_floor), mode="bilinear", align_corners=False)
        flow = F.interpolate(input=self.process(ref, supp), size=(h, w), mode="bilinear", align_corners=False)
        flow[:, 0, :, :] *= float(w) / float(w_floor)
        flow[:, 1, :, :] *= float(h) / float(h_floor)
        return flow
