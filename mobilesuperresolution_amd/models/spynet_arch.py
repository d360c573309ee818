"""flow_warp on the MI355X hot path (reference: models/spynet_arch.py:98-129, the vendored copy of the
mmedit function the BasicVSR variants call at basicvsr_arch.py:74,85 / mvvsr_arch.py:79,90).
SpyNet itself (the optical-flow prior) is out of scope."""
from __future__ import annotations

import torch

from .. import _lib as L

__all__ = ["flow_warp"]


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
