"""BasicVSR propagation trunk on the MI355X hot path.

`ConvResidualBlocks(num_in_ch, num_out_ch, num_block)` mirrors the reference class of the same name
(models/basicvsr_arch.py:108-124 with ResidualBlockNoBN :126-147; identical copies in
basicvsr_arch_origin.py and mvvsr_arch.py): conv3x3(num_in_ch -> F) + LeakyReLU(0.1), then
num_block x [x + conv2(relu(conv1(x)))]; plain convs with bias, PyTorch-default init; state_dict keys
`main.0.{weight,bias}`, `main.2.{i}.conv{1,2}.{weight,bias}`.  Input/output are NCHW fp32 like the
reference's (it is called on `torch.cat([x_i, feat_prop], 1)` inside the propagation loops,
basicvsr_arch.py:67-88); every convolution, activation, residual add and their backward run in
csrc/conv3x3.h.  Supported: num_out_ch = 24, num_in_ch in {24, 27}.  No CPU / ATen fallback.

`propagate(...)` restates the two recurrent loops of MotionVectorVSR.forward (mvvsr_arch.py:72-93) around
the trunk; flow_warp itself is the next row of SURVEY 8(f) and is taken as a callable.
"""
from __future__ import annotations

import os
from functools import lru_cache

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import packing as P

__all__ = ["ConvResidualBlocks", "ResidualBlockNoBN", "propagate"]

_DTYPES = {"fp32": torch.float32, "bf16": torch.bfloat16}


@lru_cache(maxsize=None)
def _tables(ci_real: int, device_index: int):
    t = P.c3_tables(ci_real)
    dev = torch.device("cuda", device_index)
    return (torch.from_numpy(t["w"]).to(dev), torch.from_numpy(t["grad"]).to(dev), t["off"]["size"])


def _pack(conv: nn.Conv2d, dtype):
    w = conv.weight
    idx, _, size = _tables(w.shape[1], w.device.index if w.device.index is not None else torch.cuda.current_device())
    src = torch.cat([w.detach().reshape(-1).float(), conv.bias.detach().float(), w.new_tensor([0.0, 1.0])])
    assert src.numel() == size
    return src.index_select(0, idx).to(dtype).contiguous()


class ResidualBlockNoBN(nn.Module):
    """parameter holder: conv1, conv2 (reference models/basicvsr_arch.py:126-147)"""

    def __init__(self, num_feat=64, res_scale=1, pytorch_init=False):
        super().__init__()
        if res_scale != 1:
            raise NotImplementedError("res_scale != 1 is not on the hot path (the reference always uses 1)")
        self.res_scale = res_scale
        self.conv1 = nn.Conv2d(num_feat, num_feat, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(num_feat, num_feat, 3, 1, 1, bias=True)


class ConvResidualBlocks(nn.Module):

    def __init__(self, num_in_ch=3, num_out_ch=64, num_block=15, hot_dtype=None):
        super().__init__()
        if num_out_ch != 24 or num_in_ch not in (24, 27):
            raise NotImplementedError("MI355X hot path supports ConvResidualBlocks(num_in_ch in {24,27}, 24, n) "
                                      f"(got {num_in_ch}, {num_out_ch}); there is no generic fallback")
        self.num_in_ch, self.num_feat, self.num_block = num_in_ch, num_out_ch, num_block
        name = hot_dtype or os.environ.get("SR_HOT_DTYPE", "fp32")
        self.hot_dtype = name if isinstance(name, torch.dtype) else _DTYPES[str(name).lower().replace("float32", "fp32").replace("bfloat16", "bf16")]
        self.main = nn.Sequential(nn.Conv2d(num_in_ch, num_out_ch, 3, 1, 1, bias=True), nn.Identity(),
                                  nn.Sequential(*[ResidualBlockNoBN(num_feat=num_out_ch) for _ in range(num_block)]))

    def _convs(self):
        out = [self.main[0]]
        for blk in self.main[2]:
            out += [blk.conv1, blk.conv2]
        return out

    def forward(self, fea: torch.Tensor) -> torch.Tensor:
        if not fea.is_cuda:
            raise L.HotpathError("ConvResidualBlocks (MI355X hot path) needs CUDA/HIP tensors; there is no CPU fallback")
        if fea.dim() != 4 or fea.shape[1] != self.num_in_ch:
            raise ValueError(f"expected N x {self.num_in_ch} x H x W input, got {tuple(fea.shape)}")
        params = [p for c in self._convs() for p in (c.weight, c.bias)]
        return _TrunkFunction.apply(fea, self, *params)


def _launch(name, *args):
    L.launch(name, getattr(L.lib(), name), *args, L.stream_ptr())


class _TrunkFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, fea, mod, *params):
        dt, nb, cin = mod.hot_dtype, mod.num_block, mod.num_in_ch
        code = L.DTYPE_CODE[dt]
        n, _, h, w = fea.shape
        ci0 = 32 if cin == 27 else 24
        x0 = fea.detach().permute(0, 2, 3, 1)
        if ci0 != cin:
            x0 = F.pad(x0, (0, ci0 - cin))
        x0 = x0.to(dt).contiguous()
        convs = mod._convs()
        blobs = [_pack(c, dt) for c in convs]
        acts = torch.empty((nb + 1, n, h, w, 24), dtype=dt, device=fea.device)      # a_0 .. a_nb
        mids = torch.empty((max(nb, 1), n, h, w, 24), dtype=dt, device=fea.device)  # t_i = relu(conv1(a_i))
        _launch("sr_c3_fwd", x0.data_ptr(), None, acts[0].data_ptr(), blobs[0].data_ptr(), n, h, w, ci0, 2, code)
        for i in range(nb):
            _launch("sr_c3_fwd", acts[i].data_ptr(), None, mids[i].data_ptr(), blobs[1 + 2 * i].data_ptr(), n, h, w, 24, 1, code)
            _launch("sr_c3_fwd", mids[i].data_ptr(), acts[i].data_ptr(), acts[i + 1].data_ptr(), blobs[2 + 2 * i].data_ptr(),
                    n, h, w, 24, 0, code)
        ctx.mod, ctx.x0, ctx.acts, ctx.mids, ctx.blobs = mod, x0, acts, mids, blobs
        ctx.need_dx = fea.requires_grad
        return acts[nb].permute(0, 3, 1, 2).float().contiguous()

    @staticmethod
    def backward(ctx, dy):
        mod, x0, acts, mids, blobs = ctx.mod, ctx.x0, ctx.acts, ctx.mids, ctx.blobs
        dt, nb, cin = mod.hot_dtype, mod.num_block, mod.num_in_ch
        code = L.DTYPE_CODE[dt]
        n, h, w, ci0 = x0.shape
        dev = x0.device
        wgs = 64
        g = dy.permute(0, 2, 3, 1).to(dt).contiguous()
        convs = mod._convs()
        parts = torch.empty((len(convs), wgs, 9 * 1024), dtype=torch.float32, device=dev)
        dtmp = torch.empty_like(g)
        for i in range(nb - 1, -1, -1):
            # a_{i+1} = a_i + conv2(t_i), t_i = relu(conv1(a_i))
            _launch("sr_c3_wgrad", mids[i].data_ptr(), g.data_ptr(), None, parts[2 + 2 * i].data_ptr(), wgs, n, h, w, 24, 0, code)
            _launch("sr_c3_bwd_data", g.data_ptr(), None, None, dtmp.data_ptr(), blobs[2 + 2 * i].data_ptr(), n, h, w, 24, 0, code)
            _launch("sr_c3_wgrad", acts[i].data_ptr(), dtmp.data_ptr(), mids[i].data_ptr(), parts[1 + 2 * i].data_ptr(), wgs,
                    n, h, w, 24, 1, code)
            gn = torch.empty_like(g)
            _launch("sr_c3_bwd_data", dtmp.data_ptr(), mids[i].data_ptr(), g.data_ptr(), gn.data_ptr(),
                    blobs[1 + 2 * i].data_ptr(), n, h, w, 24, 1, code)
            g = gn
        _launch("sr_c3_wgrad", x0.data_ptr(), g.data_ptr(), acts[0].data_ptr(), parts[0].data_ptr(), wgs, n, h, w, ci0, 2, code)
        dfea = None
        if ctx.need_dx:
            dx0 = torch.empty_like(x0)
            _launch("sr_c3_bwd_data", g.data_ptr(), acts[0].data_ptr(), None, dx0.data_ptr(), blobs[0].data_ptr(), n, h, w,
                    ci0, 2, code)
            dfea = dx0[..., :cin].permute(0, 3, 1, 2).float().contiguous()
        slabs = parts.sum(1)
        grads = []
        for k, c in enumerate(convs):
            _, gidx, _ = _tables(c.weight.shape[1], dev.index if dev.index is not None else torch.cuda.current_device())
            gv = slabs[k].index_select(0, gidx)
            nw = c.weight.numel()
            grads += [gv[:nw].view_as(c.weight), gv[nw:]]
        return (dfea, None, *grads)


def propagate(x, flows_forward, flows_backward, backward_trunk, forward_trunk, flow_warp, num_feat=24):
    """The two recurrent loops of the reference (mvvsr_arch.py:72-93 / basicvsr_arch.py:67-88):
    returns (backward features, forward features) per frame.  x: (b, n, 3, h, w); flows: (b, n-1, 2, h, w)."""
    b, n, _, h, w = x.shape
    out_b, out_f = [], []
    feat = x.new_zeros(b, num_feat, h, w)
    for i in range(n - 1, -1, -1):
        if i < n - 1:
            feat = flow_warp(feat, flows_backward[:, i].permute(0, 2, 3, 1))
        feat = backward_trunk(torch.cat([x[:, i], feat], dim=1))
        out_b.insert(0, feat)
    feat = torch.zeros_like(feat)
    for i in range(n):
        if i > 0:
            feat = flow_warp(feat, flows_forward[:, i - 1].permute(0, 2, 3, 1))
        feat = forward_trunk(torch.cat([x[:, i], feat], dim=1))
        out_f.append(feat)
    return out_b, out_f
