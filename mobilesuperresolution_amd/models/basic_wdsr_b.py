"""BASIC_MODEL on the MI355X hot path.

Same constructor fields, forward signature, `.scale` attribute and state_dict keys as the
reference's BASIC_MODEL (models/basic_wdsr_b.py:16-93): `head.*`, `body.{i}.body.{0,2,3}.*`,
`tail.*`, `skip.0.*`, each conv stored as `bias`, `weight_g`, `weight_v`.  The arithmetic runs in the
hand-written HIP kernels of libsr_hotpath.so: one launch for the head conv, one fused launch per
residual block, one fused launch for tail + skip + PixelShuffle + mean; backward mirrors it.  Only the
tiny weight-norm algebra (w = g v / ||v||, models/basic_wdsr_b.py:23) stays in PyTorch, so that
autograd carries the packed weight gradients back to `weight_g` / `weight_v`.

There is no CPU or ATen fallback: CPU tensors, unsupported widths or a missing library raise.
"""
from __future__ import annotations

import math
import os
from typing import List

import torch
import torch.nn as nn

from .. import hotpath as HP

__all__ = ["BASIC_MODEL", "Block"]

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}


def _hot_dtype(params) -> torch.dtype:
    name = getattr(params, "hot_dtype", None) or os.environ.get("SR_HOT_DTYPE", "fp32")
    if isinstance(name, torch.dtype):
        return name
    return _DTYPES[str(name).lower()]


class _WNConv(nn.Module):
    """Parameters of one weight-normalised conv, registered as the reference's
    torch.nn.utils.weight_norm(Conv2d(...)) registers them: bias, weight_g, weight_v."""

    def __init__(self, cin: int, cout: int, k: int, g_init: float):
        super().__init__()
        conv = nn.Conv2d(cin, cout, k)           # PyTorch-default init of weight_v, as in the reference
        self.bias = nn.Parameter(torch.zeros(cout))
        self.weight_g = nn.Parameter(torch.full((cout, 1, 1, 1), float(g_init)))
        self.weight_v = nn.Parameter(conv.weight.detach().clone())

    def weight(self) -> torch.Tensor:
        v = self.weight_v
        return v * (self.weight_g / v.flatten(1).norm(dim=1).view(-1, 1, 1, 1))


class Block(nn.Module):
    """Parameter container of one residual block (reference Block, models/basic_wdsr_b.py:96-144).
    body[1] is the parameter-free ReLU slot, so the keys are body.0 / body.2 / body.3."""

    def __init__(self, num_residual_units: int, kernel_size: int = 3, res_scale: float = 1.0):
        super().__init__()
        f = num_residual_units
        e, l = int(f * 6), int(f * 0.84)
        self.body = nn.ModuleList([_WNConv(f, e, 1, 2.0), nn.Identity(), _WNConv(e, l, 1, 2.0),
                                   _WNConv(l, f, kernel_size, res_scale)])


class BASIC_MODEL(nn.Module):

    def __init__(self, params):
        super().__init__()
        self.image_mean = float(params.image_mean)
        self.scale = int(params.scale)
        self.remain_blocks = params.num_blocks
        nin = int(params.num_channels)
        f = int(params.num_residual_units)
        self.num_residual_units = f
        if nin != 3 or f not in (24, 32) or self.scale not in (2, 3, 4):
            raise NotImplementedError(
                "MI355X hot path supports num_channels=3, num_residual_units in {24,32}, scale in {2,3,4} "
                f"(got {nin}, {f}, {self.scale}); there is no generic fallback")
        nout = self.scale * self.scale * nin
        self.hot_dtype = _hot_dtype(params)
        self.head = _WNConv(nin, f, 3, 1.0)
        self.body = nn.ModuleList([Block(f, 3, 1 / math.sqrt(params.num_blocks)) for _ in range(params.num_blocks)])
        self.tail = _WNConv(f, nout, 3, 1.0)
        self.skip = nn.ModuleList([_WNConv(nin, nout, 5, 1.0)])     # key skip.0.*, as nn.Sequential in the reference
        self.shuf = nn.Sequential()                                  # parameter-free; the shuffle is fused

    # ---- canonical (effective-weight) source vectors, differentiable w.r.t. the parameters ----
    def _sources(self):
        mean = self.image_mean
        src_head = HP.head_src(self.head.weight(), self.head.bias)
        btot = self.tail.bias + self.skip[0].bias + mean
        src_tail = HP.tail_src(self.tail.weight(), self.skip[0].weight(), btot)
        convs = [[blk.body[i] for blk in self.body] for i in (0, 2, 3)]
        ws, bs = [], []
        for layer in convs:
            v = torch.stack([c.weight_v for c in layer])
            g = torch.stack([c.weight_g for c in layer])
            ws.append(v * (g / v.flatten(2).norm(dim=2).view(v.shape[0], v.shape[1], 1, 1, 1)))
            bs.append(torch.stack([c.bias for c in layer]))
        src_body = HP.block_src(ws[0], ws[1], ws[2], bs[0], bs[1], bs[2])
        return src_head, src_body, src_tail

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise HP.L.HotpathError("BASIC_MODEL (MI355X hot path) needs a CUDA/HIP tensor; there is no CPU fallback")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected N x 3 x H x W input, got {tuple(x.shape)}")
        x = x.contiguous().float()
        src_head, src_body, src_tail = self._sources()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if x.requires_grad:
            raise NotImplementedError("gradient w.r.t. the input image is not on the hot path "
                                      "(the reference trainers never request it)")
        cfg = (self.hot_dtype, self.num_residual_units, self.scale, self.image_mean)
        if need_grad:
            return _WDSRFunction.apply(x, src_head, src_body, src_tail, cfg)
        return _wdsr_infer(x, src_head, src_body, src_tail, cfg)


def _wdsr_infer(x, src_head, src_body, src_tail, cfg):
    dtype, f, r, mean = cfg
    n, _, h, w = x.shape
    blob_h, blob_t = HP.pack_ends(src_head, src_tail, f, r, dtype)
    blob_b, cinit_b = HP.pack_blocks(src_body, f, dtype)
    a = torch.empty((n, h, w, f), dtype=dtype, device=x.device)
    b = torch.empty_like(a)
    HP.head_fwd(x, a, blob_h, mean)
    for i in range(src_body.shape[0]):
        HP.block_fwd(a, b, blob_b[i], cinit_b[i])
        a, b = b, a
    out = torch.empty((n, 3, r * h, r * w), dtype=torch.float32, device=x.device)
    HP.tail_fwd(a, x, out, blob_t, mean, r)
    return out


class _WDSRFunction(torch.autograd.Function):
    """Whole-network forward/backward on the HIP kernels.  Saves the block inputs (bf16 or fp32 NHWC);
    the E-wide and L-wide intermediates are recomputed in backward, never stored."""

    @staticmethod
    def forward(ctx, x, src_head, src_body, src_tail, cfg):
        dtype, f, r, mean = cfg
        n, _, h, w = x.shape
        nb = src_body.shape[0]
        blob_h, blob_t = HP.pack_ends(src_head, src_tail, f, r, dtype)
        blob_b, cinit_b = HP.pack_blocks(src_body, f, dtype)
        acts = torch.empty((nb + 1, n, h, w, f), dtype=dtype, device=x.device)
        HP.head_fwd(x, acts[0], blob_h, mean)
        for i in range(nb):
            HP.block_fwd(acts[i], acts[i + 1], blob_b[i], cinit_b[i])
        out = torch.empty((n, 3, r * h, r * w), dtype=torch.float32, device=x.device)
        HP.tail_fwd(acts[nb], x, out, blob_t, mean, r)
        ctx.save_for_backward(x, acts, blob_b, cinit_b, blob_t)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, dout):
        x, acts, blob_b, cinit_b, blob_t = ctx.saved_tensors
        dtype, f, r, mean = ctx.cfg
        nb = blob_b.shape[0]
        dout = dout.contiguous().float()
        grads = torch.empty_like(acts)                # grads[i] = dL/d acts[i]
        HP.tail_bwd_data(dout, grads[nb], blob_t, r)
        d_tail = HP.tail_wgrad(dout, acts[nb], x, mean, r)
        for i in range(nb - 1, -1, -1):
            HP.block_bwd_data(acts[i], grads[i + 1], grads[i], blob_b[i], cinit_b[i])
        d_body = HP.block_wgrad(acts[:nb], grads[1:], blob_b, cinit_b)
        d_head = HP.head_wgrad(grads[0], x, mean)
        return None, d_head, d_body, d_tail, None
