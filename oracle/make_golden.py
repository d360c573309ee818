#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (CPU, fp32).

Run in the build container only (the reference lives at /root/reference and
never travels to the GPU box):

    python oracle/make_golden.py

The fixtures are data (inputs, parameters, expected outputs/gradients); no
reference source is copied.  tests/test_oracle_golden.py pins oracle/wdsr_oracle.py
against them, and the -m gpu parity tests compare the HIP path with the same
files.  Fixture ids follow SURVEY.md section 8(c): G1..G9.
"""
import argparse
import hashlib
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("SR_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy().astype(np.float32)


def _params(**kw):
    ns = argparse.Namespace(image_mean=0.5, num_channels=3, scale=4, num_blocks=4,
                            num_residual_units=24, width_search=False, pretrained=False)
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


def g1_model(BASIC_MODEL):
    """C1: x4, 4 blocks / 24 units, batch 1, 48x48; fwd + L1 + bwd."""
    torch.manual_seed(0)
    m = BASIC_MODEL(_params()).train()
    # move weights off their init so that g, bias matter
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif n.endswith("weight_g"):
                p.mul_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
    x = torch.rand(1, 3, 48, 48, generator=torch.Generator().manual_seed(0), requires_grad=True)
    hr = torch.rand(1, 3, 192, 192, generator=torch.Generator().manual_seed(1))
    y = m(x)
    loss = torch.nn.functional.l1_loss(y, hr)
    loss.backward()
    d = {"x": _np(x), "hr": _np(hr), "y": _np(y), "loss": _np(loss), "dx": _np(x.grad)}
    for n, p in m.named_parameters():
        d["p/" + n] = _np(p)
        d["g/" + n] = _np(p.grad)
    # the survey's breadcrumb: untouched init, eval mode
    torch.manual_seed(0)
    m0 = BASIC_MODEL(_params()).eval()
    d["breadcrumb_mean"] = _np(m0(torch.rand(1, 3, 48, 48)).mean())
    np.savez_compressed(os.path.join(OUT, "g1_basic_model_c1.npz"), **d)
    print("G1 ok: y.mean", float(y.mean()), "loss", float(loss), "breadcrumb", float(d["breadcrumb_mean"]))


def g2_block(Block):
    for f in (24, 32):
        torch.manual_seed(10 + f)
        b = Block(num_residual_units=f, kernel_size=3, res_scale=0.5)
        g = torch.Generator().manual_seed(2)
        with torch.no_grad():
            for n, p in b.named_parameters():
                if n.endswith("bias"):
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
                elif n.endswith("weight_g"):
                    p.mul_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
        x = torch.randn(2, f, 20, 28, generator=g, requires_grad=True)
        dy = torch.randn(2, f, 20, 28, generator=g)
        seq = b.body
        h = seq[1](seq[0](x))
        hc = h.clone()          # ReLU is in-place in the reference
        t = seq[2](h)
        r = seq[3](t)
        y = r + x
        y.backward(dy)
        d = {"x": _np(x), "dy": _np(dy), "y": _np(y), "h": _np(hc), "t": _np(t), "r": _np(r),
             "dx": _np(x.grad)}
        for n, p in b.named_parameters():
            d["p/" + n] = _np(p)
            d["g/" + n] = _np(p.grad)
        y2 = b(x.detach())
        assert torch.equal(y2, y.detach())
        np.savez_compressed(os.path.join(OUT, f"g2_block_f{f}.npz"), **d)
        print(f"G2 F={f} ok")


def g3_pretrained(BASIC_MODEL):
    path = os.path.join(REF, "models", "pretrained_weights", "wdsr_b_x2_8_24.pt")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    m = BASIC_MODEL(_params(scale=2, num_blocks=8)).eval()
    m.load_state_dict(sd, strict=True)
    gen = torch.Generator().manual_seed(3)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 40), torch.linspace(0, 1, 56), indexing="ij")
    img = torch.stack([0.5 + 0.4 * torch.sin(9 * xx + 3 * yy), yy * xx, 0.5 + 0.5 * torch.cos(13 * yy)], 0)
    img = (img + 0.03 * torch.randn(img.shape, generator=gen)).clamp(0, 1)[None]
    with torch.no_grad():
        y = m(img)
    d = {"x": _np(img), "y": _np(y)}
    h = hashlib.sha256()
    for k in sorted(sd):
        d["p/" + k] = _np(sd[k])
        h.update(k.encode()); h.update(_np(sd[k]).tobytes())
    d["sha256"] = np.frombuffer(h.digest(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "g3_pretrained_x2_8_24.npz"), **d)
    print("G3 ok", y.shape, float(y.mean()))


def g4_pixel_shuffle():
    d = {}
    for r in (2, 3, 4):
        c, h, w = 3, 5, 7
        x = torch.arange(2 * c * r * r * h * w, dtype=torch.float32).reshape(2, c * r * r, h, w)
        d[f"x_r{r}"] = _np(x)
        d[f"y_r{r}"] = _np(torch.nn.PixelShuffle(r)(x))
    np.savez_compressed(os.path.join(OUT, "g4_pixel_shuffle.npz"), **d)
    print("G4 ok")


def g5_rounding(ops):
    cases = {
        "all_keep": torch.linspace(0.5, 1.0, 24),
        "straddle": torch.tensor([0.1, 0.49, 0.5, 0.51, 0.9, 0.2, 0.7, 0.3, 0.6, 0.4, 0.55, 0.45,
                                  0.8, 0.05, 0.95, 0.5, 0.499999, 0.500001, 0.0, 1.0, 0.65, 0.35, 0.75, 0.25]),
        "fallback": torch.tensor([0.1, 0.2, 0.3, 0.4, 0.45, 0.6, 0.7, 0.05, 0.15, 0.25, 0.35, 0.42,
                                  0.41, 0.33, 0.22, 0.11, 0.44, 0.43, 0.01, 0.02, 0.03, 0.04, 0.06, 0.07]),
        "ties": torch.tensor([0.3] * 10 + [0.2] * 10 + [0.9, 0.8, 0.4, 0.4]),
    }
    d = {}
    for name, w in cases.items():
        w4 = w.reshape(-1, 1, 1, 1).clone()
        for lc in (8, 0):
            d[f"{name}/w"] = _np(w4)
            d[f"{name}/mask_lc{lc}"] = _np(ops.rounding(w4, lc))
        conv = ops.BinaryConv2d(24, 24, groups=24, least_channel=8)
        with torch.no_grad():
            conv.weight.copy_(w4)
        x = torch.randn(2, 24, 6, 5, generator=torch.Generator().manual_seed(5), requires_grad=True)
        dy = torch.randn(2, 24, 6, 5, generator=torch.Generator().manual_seed(6))
        y = conv(x)
        y.backward(dy)
        d[f"{name}/x"], d[f"{name}/dy"], d[f"{name}/y"] = _np(x), _np(dy), _np(y)
        d[f"{name}/dx"], d[f"{name}/dw"] = _np(x.grad), _np(conv.weight.grad)
    np.savez_compressed(os.path.join(OUT, "g5_binary_mask.npz"), **d)
    print("G5 ok")


def g6_split_block(wdsr_b):
    for f in (24, 32):
        torch.manual_seed(60 + f)
        blk = wdsr_b.Split_Block(num_residual_units=f, kernel_size=3)
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            w = torch.rand(f, generator=g) * 0.6 + 0.2     # some channels below 0.5
            blk.split.weight.copy_(w.reshape(f, 1, 1, 1))
            for n, p in blk.named_parameters():
                if n.endswith("bias"):
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
        x = torch.randn(2, f, 14, 18, generator=g, requires_grad=True)
        dy = torch.randn(2, f, 14, 18, generator=g)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y = blk.forward_body(x)
        y.backward(dy)
        d = {"x": _np(x), "dy": _np(dy), "y": _np(y), "dx": _np(x.grad)}
        for n, p in blk.named_parameters():
            d["p/" + n] = _np(p)
            if p.grad is not None:
                d["g/" + n] = _np(p.grad)
        np.savez_compressed(os.path.join(OUT, f"g6_split_block_f{f}.npz"), **d)
        print(f"G6 F={f} ok; kept", int(ops_mask_count(blk)))


def ops_mask_count(blk):
    return (blk.split.weight.detach() >= 0.5).sum()


def g7_g8_vsr():
    # models/basicvsr_arch.py hard-imports mmedit (absent).  ConvResidualBlocks
    # itself is plain torch.nn; map the two imported names onto the vendored
    # copies in models/spynet_arch.py so the module body executes.
    import models.spynet_arch as sp
    for name in ("mmedit", "mmedit.models", "mmedit.models.common", "mmedit.models.backbones",
                 "mmedit.models.backbones.sr_backbones", "mmedit.models.backbones.sr_backbones.basicvsr_net",
                 "mmedit.utils", "mmedit.models.registry"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["mmedit.models.common"].flow_warp = sp.flow_warp
    sys.modules["mmedit.models.common"].PixelShufflePack = None            # unused by the trunk
    sys.modules["mmedit.models.backbones.sr_backbones.basicvsr_net"].SPyNet = sp.SpyNet
    sys.modules["mmedit.models.backbones.sr_backbones.basicvsr_net"].ResidualBlocksWithInputConv = None
    try:
        import models.basicvsr_arch as bv
    except Exception as e:  # ordinary python error: record and fall back
        print("G7: basicvsr_arch not importable here:", repr(e))
        bv = None
    if bv is not None:
        torch.manual_seed(70)
        trunk = bv.ConvResidualBlocks(27, 24, 8)
        g = torch.Generator().manual_seed(8)
        x = torch.randn(1, 27, 24, 20, generator=g, requires_grad=True)
        dy = torch.randn(1, 24, 24, 20, generator=g)
        y = trunk(x)
        y.backward(dy)
        d = {"x": _np(x), "dy": _np(dy), "y": _np(y), "dx": _np(x.grad)}
        for n, p in trunk.named_parameters():
            d["p/" + n] = _np(p)
            d["g/" + n] = _np(p.grad)
        np.savez_compressed(os.path.join(OUT, "g7_vsr_trunk.npz"), **d)
        print("G7 ok", y.shape)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 24, 16, 20, generator=g, requires_grad=True)
    flow = (torch.rand(2, 16, 20, 2, generator=g) * 6 - 3).requires_grad_(True)
    dy = torch.randn(2, 24, 16, 20, generator=g)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = sp.flow_warp(x, flow)
    y.backward(dy)
    np.savez_compressed(os.path.join(OUT, "g8_flow_warp.npz"), x=_np(x), flow=_np(flow), dy=_np(dy),
                        y=_np(y), dx=_np(x.grad), dflow=_np(flow.grad))
    print("G8 ok")


def main():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, REF)
    torch.set_num_threads(4)
    import models  # noqa: the reference package
    from models.basic_wdsr_b import BASIC_MODEL, Block
    import models.ops as ops
    import models.wdsr_b as wdsr_b
    g1_model(BASIC_MODEL)
    g2_block(Block)
    g3_pretrained(BASIC_MODEL)
    g4_pixel_shuffle()
    g5_rounding(ops)
    g6_split_block(wdsr_b)
    g7_g8_vsr()


if __name__ == "__main__":
    main()
