"""Thin host layer over the C ABI: packs parameters into MFMA fragment blobs (one torch gather per
layer family) and enqueues the HIP kernels on torch's current stream.  No arithmetic of the SR path
happens in PyTorch here; torch is used for device memory, streams and the gather/sum plumbing."""
from __future__ import annotations

import ctypes
from functools import lru_cache
from typing import Tuple

import numpy as np
import torch

from . import _lib as L
from . import packing as P


KernelTimer, set_timer, _launch = L.KernelTimer, L.set_timer, L.launch



@lru_cache(maxsize=None)
def const01(device: torch.device, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """the [0, 1] constant slots of a canonical source vector, uploaded once per device
    (Tensor.new_tensor is a synchronous host-to-device copy: it drains the GPU queue on every call)"""
    return torch.tensor([0.0, 1.0], dtype=dtype, device=device)

def block_dims(F: int) -> Tuple[int, int, int]:
    """(F, E, L) of the reference Block: expand 6, linear 0.84 (models/basic_wdsr_b.py:105-106)."""
    return F, int(F * 6), int(F * 0.84)


@lru_cache(maxsize=None)
def _dev_tables(F: int, device_index: int):
    f, e, l = block_dims(F)
    tab = P.block_tables(f, e, l)
    gt = P.block_grad_tables(f, e, l)
    dev = torch.device("cuda", device_index)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    g = tab["geom"]
    o = g.off
    # canonical gradient vector order = w1 | w2 | w3 | b1 | b2 | b3 (src layout without constants)
    E_, L_ = g.E, g.L
    n_w1, n_w2, n_w3 = E_ * f, L_ * E_, f * L_ * 9
    ga, gb = gt["a"], gt["b"]
    # slab A order: w1, w2, b1, b2 ; slab B order: w3, b3  -> positions inside cat(slabA, slabB)
    a_w1, a_w2 = ga[:n_w1], ga[n_w1:n_w1 + n_w2]
    a_b1, a_b2 = ga[n_w1 + n_w2:n_w1 + n_w2 + E_], ga[n_w1 + n_w2 + E_:]
    b_w3, b_b3 = gb[:n_w3] + gt["a_size"], gb[n_w3:] + gt["a_size"]
    grad_idx = np.concatenate([a_w1, a_w2, b_w3, a_b1, a_b2, b_b3])
    assert grad_idx.size == o["zero"]
    return dict(w=t(tab["w"]), cinit=t(tab["cinit"]), grad=t(grad_idx), geom=g, nfrag=tab["nfrag"],
                slab_a=gt["a_size"], slab_b=gt["b_size"], src_size=o["size"])


def tables(F: int, device: torch.device):
    return _dev_tables(F, device.index if device.index is not None else torch.cuda.current_device())


def block_src(w1: torch.Tensor, w2: torch.Tensor, w3: torch.Tensor, b1: torch.Tensor, b2: torch.Tensor,
              b3: torch.Tensor) -> torch.Tensor:
    """Canonical per-block source vectors [NB, S] from stacked effective weights/biases
    (w1 [NB,E,F], w2 [NB,L,E], w3 [NB,F,L,3,3], b* [NB,...]); differentiable."""
    nb = w1.shape[0]
    const = const01(w1.device, w1.dtype).expand(nb, 2)
    return torch.cat([w1.reshape(nb, -1), w2.reshape(nb, -1), w3.reshape(nb, -1), b1, b2, b3, const], dim=1)


def pack_blocks(src_all: torch.Tensor, F: int, dtype: torch.dtype):
    """src_all [NB, S] fp32 -> (blob [NB, nfrag*512] dtype, cinit [NB, C] fp32)."""
    tb = tables(F, src_all.device)
    assert src_all.shape[1] == tb["src_size"], (src_all.shape, tb["src_size"])
    s = src_all.detach().float()
    blob = s.index_select(1, tb["w"]).to(dtype).contiguous()
    cinit = s.index_select(1, tb["cinit"]).contiguous()
    return blob, cinit


def block_fwd(x: torch.Tensor, y: torch.Tensor, blob_i: torch.Tensor, cinit_i: torch.Tensor):
    n, h, w, f = x.shape
    _launch("sr_wdsr_block_fwd", L.lib().sr_wdsr_block_fwd, L.ptr(x), L.ptr(y), L.ptr(blob_i), L.ptr(cinit_i), n, h, w, f,
                                      L.DTYPE_CODE[x.dtype], L.stream_ptr())


def block_bwd_data(x: torch.Tensor, dy: torch.Tensor, dx: torch.Tensor, blob_i: torch.Tensor,
                   cinit_i: torch.Tensor):
    n, h, w, f = x.shape
    _launch("sr_wdsr_block_bwd_data", L.lib().sr_wdsr_block_bwd_data, L.ptr(x), L.ptr(dy), L.ptr(dx), L.ptr(blob_i), L.ptr(cinit_i),
                                           n, h, w, f, L.DTYPE_CODE[x.dtype], L.stream_ptr())


def block_wgrad(xs: torch.Tensor, dys: torch.Tensor, blob: torch.Tensor, cinit: torch.Tensor,
                wgs_per_layer: int = 16) -> torch.Tensor:
    """xs, dys: [NB, N, H, W, F] (block inputs / output gradients); blob [NB, .], cinit [NB, .].
    Returns d_src [NB, S] (zeros at the two constant slots), fp32."""
    nb, n, h, w, f = xs.shape
    tb = tables(f, xs.device)
    pa = torch.empty((nb, wgs_per_layer, tb["slab_a"]), dtype=torch.float32, device=xs.device)
    pb = torch.empty((nb, wgs_per_layer, tb["slab_b"]), dtype=torch.float32, device=xs.device)
    assert xs.is_contiguous() and dys.is_contiguous() and blob.is_contiguous() and cinit.is_contiguous()
    _launch("sr_wdsr_block_wgrad", L.lib().sr_wdsr_block_wgrad, L.ptr(xs), L.ptr(dys), L.ptr(blob), L.ptr(cinit), L.ptr(pa), L.ptr(pb),
                                        nb, wgs_per_layer, n, h, w, f, L.DTYPE_CODE[xs.dtype],
                                        xs.stride(0), dys.stride(0), blob.stride(0), cinit.stride(0),
                                        L.stream_ptr())
    slab = torch.cat([pa.sum(1), pb.sum(1)], dim=1)
    g = slab.index_select(1, tb["grad"])
    return torch.cat([g, g.new_zeros(nb, 2)], dim=1)


# =====================================================================================
# head / tail (+ skip + PixelShuffle)
# =====================================================================================
@lru_cache(maxsize=None)
def _dev_ends_tables(F: int, R: int, device_index: int):
    tab = P.ends_tables(F, R)
    gt = P.ends_grad_tables(F, R)
    dev = torch.device("cuda", device_index)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    g = tab["geom"]
    return dict(head=t(tab["head"]), tail=t(tab["tail"]), head_grad=t(gt["head"]), tail_grad=t(gt["tail"]),
                tail_slab=gt["tail_size"], head_slab=gt["head_size"], geom=g,
                head_size=g.head_off["size"], tail_size=g.tail_off["size"])


def ends_tables(F: int, R: int, device: torch.device):
    return _dev_ends_tables(F, R, device.index if device.index is not None else torch.cuda.current_device())


def head_src(wh: torch.Tensor, bh: torch.Tensor) -> torch.Tensor:
    """canonical head source: wh (F,3,3,3) | bh (F) | 0 | 1"""
    return torch.cat([wh.reshape(-1), bh, const01(wh.device, wh.dtype)])


def tail_src(wt: torch.Tensor, ws: torch.Tensor, btot: torch.Tensor) -> torch.Tensor:
    """canonical tail source: wt (CO,F,3,3) | ws (CO,3,5,5) | bt + bs + mean (CO) | 0 | 1"""
    return torch.cat([wt.reshape(-1), ws.reshape(-1), btot, const01(wt.device, wt.dtype)])


def pack_ends(src_head: torch.Tensor, src_tail: torch.Tensor, F: int, R: int, dtype: torch.dtype):
    tb = ends_tables(F, R, src_head.device)
    assert src_head.numel() == tb["head_size"] and src_tail.numel() == tb["tail_size"]
    bh = src_head.detach().float().index_select(0, tb["head"]).to(dtype).contiguous()
    bt = src_tail.detach().float().index_select(0, tb["tail"]).to(dtype).contiguous()
    return bh, bt


def pack_head(src_head: torch.Tensor, F: int, dtype: torch.dtype):
    tb = ends_tables(F, 4, src_head.device)
    return src_head.detach().float().index_select(0, tb["head"]).to(dtype).contiguous()


def pack_tail(src_tail: torch.Tensor, F: int, R: int, dtype: torch.dtype):
    tb = ends_tables(F, R, src_tail.device)
    assert src_tail.numel() == tb["tail_size"]
    return src_tail.detach().float().index_select(0, tb["tail"]).to(dtype).contiguous()


def head_fwd(x: torch.Tensor, y: torch.Tensor, blob: torch.Tensor, mean: float):
    n, _, h, w = x.shape
    _launch("sr_head_fwd", L.lib().sr_head_fwd, L.ptr(x), L.ptr(y), L.ptr(blob), mean, n, h, w, y.shape[-1],
                                L.DTYPE_CODE[y.dtype], L.stream_ptr())


def tail_fwd(feat: torch.Tensor, x: torch.Tensor, out: torch.Tensor, blob: torch.Tensor, mean: float, R: int):
    n, h, w, f = feat.shape
    _launch("sr_tail_fwd", L.lib().sr_tail_fwd, L.ptr(feat), L.ptr(x), L.ptr(out), L.ptr(blob), mean, n, h, w, f, R,
                                L.DTYPE_CODE[feat.dtype], L.stream_ptr())


def tail_bwd_data(dout: torch.Tensor, dfeat: torch.Tensor, blob: torch.Tensor, R: int):
    n, h, w, f = dfeat.shape
    _launch("sr_tail_bwd_data", L.lib().sr_tail_bwd_data, L.ptr(dout), L.ptr(dfeat), L.ptr(blob), n, h, w, f, R,
                                     L.DTYPE_CODE[dfeat.dtype], L.stream_ptr())


def tail_wgrad(dout: torch.Tensor, feat: torch.Tensor, x: torch.Tensor, mean: float, R: int,
               wgs: int = 64) -> torch.Tensor:
    """returns d_src_tail (zeros at the constant slots)"""
    n, h, w, f = feat.shape
    tb = ends_tables(f, R, feat.device)
    part = torch.empty((wgs, tb["tail_slab"]), dtype=torch.float32, device=feat.device)
    _launch("sr_tail_wgrad", L.lib().sr_tail_wgrad, L.ptr(dout), L.ptr(feat), L.ptr(x), mean, L.ptr(part), wgs, n, h, w, f, R,
                                  L.DTYPE_CODE[feat.dtype], L.stream_ptr())
    g = part.sum(0).index_select(0, tb["tail_grad"])
    return torch.cat([g, g.new_zeros(2)])


def _tail_grad_vector(part: torch.Tensor, tb) -> torch.Tensor:
    g = part.sum(0).index_select(0, tb["tail_grad"])
    return torch.cat([g, g.new_zeros(2)])


def tail_bwd(dout: torch.Tensor, feat: torch.Tensor, x: torch.Tensor, blob: torch.Tensor, mean: float, R: int, wgs: int = 256):
    """tail_bwd_data + tail_wgrad in ONE launch (bf16: the gradient image is un-shuffled once for both): (dfeat, d_src_tail)"""
    n, h, w, f = feat.shape
    tb = ends_tables(f, R, feat.device)
    part = torch.empty((wgs, tb["tail_slab"]), dtype=torch.float32, device=feat.device)
    dfeat = torch.empty_like(feat)
    _launch("sr_tail_bwd", L.lib().sr_tail_bwd, L.ptr(dout), L.ptr(feat), L.ptr(x), mean, L.ptr(blob), L.ptr(dfeat), L.ptr(part), wgs,
            n, h, w, f, R, L.DTYPE_CODE[feat.dtype], L.stream_ptr())
    return dfeat, _tail_grad_vector(part, tb)


def tail_bwd_loss(out: torch.Tensor, hr: torch.Tensor, kind: int, gscale: float, feat: torch.Tensor, x: torch.Tensor,
                  blob: torch.Tensor, mean: float, R: int, wgs: int = 256):
    """the same with d(loss)/d(out) FORMED in the kernel from out and hr (kind 1: L1, 2: Charbonnier; gscale = upstream / numel):
    (dfeat, d_src_tail, per-workgroup loss sums)"""
    n, h, w, f = feat.shape
    tb = ends_tables(f, R, feat.device)
    part = torch.empty((wgs, tb["tail_slab"]), dtype=torch.float32, device=feat.device)
    loss_part = torch.empty(wgs, dtype=torch.float32, device=feat.device)
    dfeat = torch.empty_like(feat)
    _launch("sr_tail_bwd_loss", L.lib().sr_tail_bwd_loss, L.ptr(out), L.ptr(hr), kind, gscale, L.ptr(loss_part), L.ptr(feat), L.ptr(x),
            mean, L.ptr(blob), L.ptr(dfeat), L.ptr(part), wgs, n, h, w, f, R, L.DTYPE_CODE[feat.dtype], L.stream_ptr())
    return dfeat, _tail_grad_vector(part, tb), loss_part


def head_wgrad(dy0: torch.Tensor, x: torch.Tensor, mean: float, wgs: int = 64) -> torch.Tensor:
    n, h, w, f = dy0.shape
    tb = ends_tables(f, 4, dy0.device)
    part = torch.empty((wgs, tb["head_slab"]), dtype=torch.float32, device=dy0.device)
    _launch("sr_head_wgrad", L.lib().sr_head_wgrad, L.ptr(dy0), L.ptr(x), mean, L.ptr(part), wgs, n, h, w, f,
                                  L.DTYPE_CODE[dy0.dtype], L.stream_ptr())
    g = part.sum(0).index_select(0, tb["head_grad"])
    return torch.cat([g, g.new_zeros(2)])
