"""Full-image inference in tiles for the evaluation loop (reference: utils/estimate.py:44-49,112-128 runs `model(lr)` on
whole Set5 / DIV2K images).

`tiled_forward(model, lr, tile)` cuts the LR image into windows of `tile` + the model's receptive-field halo on every
side, runs ALL windows of a chunk as one batch through the model's ordinary forward (one `sr_wdsr_net_forward` call
per chunk: many workgroups per launch instead of one small image), and writes each window's core into the output.  Every
window has the same size; a window that would cross the image border is shifted inward instead, so that its outer edge
IS the image border (where the convolutions' zero padding is the real one) and its inner edges stay at least one halo
away from the pixels it contributes.  An output pixel therefore sees exactly the inputs it sees in the untiled run, in
the same order: the result is bit-identical to `model(lr)` (tests/test_gpu_model.py), in fp32 and in bf16."""
from __future__ import annotations

import torch

__all__ = ["tiled_forward", "window_plan"]


def _axis(size: int, tile: int, halo: int):
    """[(window origin, core start, core stop)] along one axis; window length min(size, tile + 2 halo)"""
    win = min(size, tile + 2 * halo)
    out, c0 = [], 0
    while c0 < size:
        c1 = min(size, c0 + tile)
        o = min(max(c0 - halo, 0), size - win)
        assert o <= c0 and c1 <= o + win and (o == 0 or c0 - o >= halo) and (o + win == size or o + win - c1 >= halo)
        out.append((o, c0, c1))
        c0 = c1
    return win, out


def window_plan(h: int, w: int, tile, halo: int):
    """(window height, window width, [(y origin, x origin, core y0, y1, x0, x1)])"""
    th, tw = (tile, tile) if isinstance(tile, int) else tile
    wh, ys = _axis(h, th, halo)
    ww, xs = _axis(w, tw, halo)
    return wh, ww, [(oy, ox, y0, y1, x0, x1) for (oy, y0, y1) in ys for (ox, x0, x1) in xs]


@torch.no_grad()
def tiled_forward(model, lr: torch.Tensor, tile=96, max_windows: int = 256):
    """lr (N, 3, H, W) -> (N, 3, scale H, scale W), identical to model(lr).  `tile`: core size in LR pixels (int or (h, w));
    `max_windows`: windows per forward call (bounds the activation memory: 16 blocks x 2 bytes x 24 channels per pixel)."""
    if lr.dim() != 4:
        raise ValueError(f"expected N x 3 x H x W, got {tuple(lr.shape)}")
    halo = model.receptive_halo()
    s = model.scale
    n, c, h, w = lr.shape
    wh, ww, plan = window_plan(h, w, tile, halo)
    if len(plan) == 1:
        return model(lr)
    out = torch.empty((n, c, h * s, w * s), dtype=torch.float32, device=lr.device)
    jobs = [(i,) + p for i in range(n) for p in plan]
    for k in range(0, len(jobs), max_windows):
        chunk = jobs[k:k + max_windows]
        batch = torch.stack([lr[i, :, oy:oy + wh, ox:ox + ww] for (i, oy, ox, *_rest) in chunk])
        sr = model(batch)
        for j, (i, oy, ox, y0, y1, x0, x1) in enumerate(chunk):
            out[i, :, y0 * s:y1 * s, x0 * s:x1 * s] = sr[j, :, (y0 - oy) * s:(y1 - oy) * s, (x0 - ox) * s:(x1 - ox) * s]
    return out
