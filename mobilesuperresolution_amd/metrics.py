"""`psnr` / `psnr_y` of the evaluation loop on the MI355X (reference: common/metrics.py:10-38, called by
utils/estimate.py:102-104,123-128).  Same names, arguments and return value (a 0-dim tensor: the per-image values SUMMED
over the batch), computed by csrc/metrics.h where the SR image already is instead of after `.to('cpu')`.  `ssim` wraps
skimage / mmedit in the reference and is out of scope.  No CPU fallback."""
from __future__ import annotations

import torch

from . import _lib as L

__all__ = ["psnr", "psnr_y"]


def _run(sr, hr, shave, luma):
    if not (sr.is_cuda and hr.is_cuda):
        raise L.HotpathError("psnr / psnr_y (MI355X hot path) need CUDA/HIP tensors; there is no CPU fallback")
    if sr.device != hr.device:
        raise L.HotpathError(f"sr on {sr.device}, hr on {hr.device}")
    if sr.shape != hr.shape or sr.dim() < 3:
        raise ValueError(f"sr {tuple(sr.shape)} vs hr {tuple(hr.shape)}")
    if hr.dtype != torch.float32:
        raise NotImplementedError("hr must be float32 (the reference casts sr to hr's dtype; its loaders produce float32)")
    c, h, w = sr.shape[-3:]
    if luma and not (sr.dim() == 4 and sr.shape[1] == 3):
        luma = -1                                    # metrics.py:29 tests shape[1]: anything else skips the luma filter
    s = sr.detach().to(hr.dtype).reshape(-1, c, h, w).contiguous()
    t = hr.detach().reshape(-1, c, h, w).contiguous()
    n = s.shape[0]
    wgs = max(1, min(64, (c * h * w + 4095) // 4096))
    with torch.cuda.device(sr.device):
        partial = torch.empty(n * wgs, dtype=torch.float32, device=sr.device)
        out = torch.empty((), dtype=torch.float32, device=sr.device)
        L.launch("sr_psnr", L.lib().sr_psnr, s.data_ptr(), t.data_ptr(), partial.data_ptr(), out.data_ptr(), n, c, h, w,
                 int(shave) if shave else 0, luma, wgs, L.stream_ptr())
    return out


def psnr(sr, hr, shave=4):
    """common/metrics.py:10-19"""
    return _run(sr, hr, shave, 0)


def psnr_y(sr, hr, shave=4):
    """common/metrics.py:22-38 (including the unused quantised copy: sr is clamped, not quantised)"""
    return _run(sr, hr, shave, 1)
