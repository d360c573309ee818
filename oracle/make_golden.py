#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (CPU, fp32).

Run in the build container only (the reference lives at /root/reference and
never travels to the GPU box):

    python oracle/make_golden.py

The fixtures are data (inputs, parameters, expected outputs/gradients); no
reference source is copied.  tests/test_oracle_golden.py pins oracle/wdsr_oracle.py
against them, and the -m gpu parity tests compare the HIP path with the same
files.  Fixture ids follow SURVEY.md section 8(c): G1..G9, then G10..G14 (whole NAS model, video models, Set5-shaped
images, the dataset item path).
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


def _mmedit_stubs():
    """models/{basicvsr_arch,basicvsr_arch_origin,mvvsr_arch}.py hard-import mmedit (absent here): map the
    imported names onto the vendored copies in models/spynet_arch.py so the module bodies execute."""
    import models.spynet_arch as sp
    for name in ("mmedit", "mmedit.models", "mmedit.models.common", "mmedit.models.backbones",
                 "mmedit.models.backbones.sr_backbones", "mmedit.models.backbones.sr_backbones.basicvsr_net",
                 "mmedit.utils", "mmedit.models.registry"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["mmedit.models.common"].flow_warp = sp.flow_warp
    sys.modules["mmedit.models.common"].PixelShufflePack = None
    sys.modules["mmedit.models.backbones.sr_backbones.basicvsr_net"].SPyNet = lambda pretrained=None: sp.SpyNet(None)
    sys.modules["mmedit.models.backbones.sr_backbones.basicvsr_net"].ResidualBlocksWithInputConv = None
    return sp


def g10_nas_model():
    """Whole reference NAS_MODEL (models/wdsr_b.py:30-137): forward + L1 + speed loss (search.py:71-81, utils/loss.py)
    + backward in train mode, forward in eval mode, one block gated off (alpha1 >= alpha2), partial global and split
    masks.  Two environment shims only, both about the CUDA-saved latency MLP that the forward never uses
    (speed_estimator.py:37-42): torch.load gets map_location='cpu', weights_only=True; Tensor.cuda is the identity
    for the `kernels` constant at speed_estimator.py:68."""
    import models.wdsr_b as wdsr_b
    from utils.loss import SpeedLoss
    from speed_models.helpers import get_ori_speed
    real_load, real_cuda = torch.load, torch.Tensor.cuda
    torch.load = lambda f, *a, **k: real_load(f, map_location="cpu", weights_only=True)
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        torch.manual_seed(100)
        ns = _params(num_blocks=4, num_residual_units=24, width_search=True, pretrained=False)
        m = wdsr_b.NAS_MODEL(ns).train()
        g = torch.Generator().manual_seed(101)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "speed_estimator" in n:
                    continue
                if n.endswith("bias"):
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
                elif n.endswith("weight_g"):
                    p.mul_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            m.mask.weight.copy_(torch.rand(24, 1, 1, 1, generator=g) * 0.7 + 0.25)      # some global channels off
            for blk in m.body:
                blk.split.weight.copy_(torch.rand(24, 1, 1, 1, generator=g) * 0.7 + 0.2)
            m.body[1].alpha1.fill_(1.5)                                                 # block 1 gated off
        x = torch.rand(2, 3, 20, 28, generator=g)
        hr = torch.rand(2, 3, 80, 112, generator=g)
        d = {"x": _np(x), "hr": _np(hr)}
        for k, v in m.state_dict().items():
            if "speed_estimator" not in k:
                d["p/" + k] = _np(v)
        ori = get_ori_speed(4, 24)
        target = 0.5 * ori
        crit = SpeedLoss(scale=float(ori - target))
        out, speed = m(x)
        l1 = torch.nn.functional.l1_loss(out, hr)
        ls = crit(speed=speed, target=target, gamma=0.1, method="clamp")
        (1.0 * l1 + ls).backward()
        d.update(out_train=_np(out), speed_train=_np(speed), loss_l1=_np(l1), loss_speed=_np(ls),
                 ori_speed=np.float32(ori), speed_target=np.float32(target))
        for n, p in m.named_parameters():
            if "speed_estimator" in n or p.grad is None:
                continue
            d["g/" + n] = _np(p.grad)
        for k, v in m.state_dict().items():                    # beta1 / beta2 are rewritten inside forward (:534)
            if k.endswith(("beta1", "beta2")):
                d["after/" + k] = _np(v)
        m.eval()
        with torch.no_grad():
            oe, se = m(x)
        d.update(out_eval=_np(oe), speed_eval=_np(se), current_blocks=np.int32(m.get_current_blocks()),
                 block_status=np.array(m.get_block_status(), dtype=np.int32))
        np.savez_compressed(os.path.join(OUT, "g10_nas_model.npz"), **d)
        print("G10 ok: speed", float(speed), "eval speed", float(se), "blocks", m.get_current_blocks(),
              "grads", sum(1 for k in d if k.startswith("g/")))
    finally:
        torch.load, torch.Tensor.cuda = real_load, real_cuda


def g11_g12_video_models():
    """G11: the reference MotionVectorVSR (models/mvvsr_arch.py:11-109, the trainer's 'basic_mv' model,
    train_video_superresolution.py:251: num_feat=20, num_block=8) on a 2 x 4-frame clip with motion vectors: output,
    per-frame trunk features of both directions (forward hooks), input gradient, every parameter gradient.
    G12: BasicVSR_origin (models/basicvsr_arch_origin.py:10-95) with its SPyNet flow estimate (out of scope) replaced
    by the same given flows: pins the propagation loops with 24 features and the PixelShuffle(2) x 2 upsampler."""
    _mmedit_stubs()
    import models.mvvsr_arch as mv
    import models.basicvsr_arch_origin as bo
    import warnings
    torch.manual_seed(110)
    m = mv.MotionVectorVSR(num_feat=20, num_block=2, spynet_path=None).train()
    g = torch.Generator().manual_seed(111)
    b, n, h, w = 2, 3, 12, 16
    frames = torch.rand(b, n, 3, h, w, generator=g)
    mvs = torch.rand(b, n, 2, h, w, generator=g) * 5 - 2.5
    x = torch.cat([frames, mvs], dim=2).requires_grad_(True)
    feats = {"backward_trunk": [], "forward_trunk": []}
    hooks = [getattr(m, k).register_forward_hook(lambda mod, i, o, k=k: feats[k].append(o.detach().clone())) for k in feats]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = m(x, 4 * h, 4 * w)
    tgt = torch.rand(out.shape, generator=g)
    loss = torch.sqrt((out - tgt) ** 2 + 1e-12).mean()          # Charbonnier, train_video_superresolution.py:43-53
    loss.backward()
    for hk in hooks:
        hk.remove()
    d = {"x": _np(x), "target": _np(tgt), "out": _np(out), "loss": _np(loss), "dx": _np(x.grad),
         "feat_backward": _np(torch.stack(feats["backward_trunk"], 1)),      # in call order: frame n-1 .. 0
         "feat_forward": _np(torch.stack(feats["forward_trunk"], 1))}
    for k, p in m.named_parameters():
        if k.startswith("spynet"):
            continue
        d["p/" + k] = _np(p)
        if p.grad is not None:
            d["g/" + k] = _np(p.grad)
    np.savez_compressed(os.path.join(OUT, "g11_mvvsr.npz"), **d)
    print("G11 ok", tuple(out.shape), "loss", float(loss))

    torch.manual_seed(120)
    m2 = bo.BasicVSR_origin(num_feat=24, num_block=2, spynet_path=None).train()
    ff = torch.rand(b, n - 1, 2, h, w, generator=g) * 4 - 2
    fb = torch.rand(b, n - 1, 2, h, w, generator=g) * 4 - 2
    m2.get_flow = lambda x: (ff, fb)                              # SPyNet is out of scope: flows are given
    x2 = frames.clone().requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out2 = m2(x2, 4 * h, 4 * w)
    tgt2 = torch.rand(out2.shape, generator=g)
    loss2 = torch.sqrt((out2 - tgt2) ** 2 + 1e-12).mean()
    loss2.backward()
    d = {"x": _np(x2), "flows_forward": _np(ff), "flows_backward": _np(fb), "target": _np(tgt2), "out": _np(out2),
         "loss": _np(loss2), "dx": _np(x2.grad)}
    for k, p in m2.named_parameters():
        if k.startswith("spynet"):
            continue
        d["p/" + k] = _np(p)
        if p.grad is not None:
            d["g/" + k] = _np(p.grad)
    np.savez_compressed(os.path.join(OUT, "g12_basicvsr_origin.npz"), **d)
    print("G12 ok", tuple(out2.shape), "loss", float(loss2))


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from set5_like import SET5_SHAPES, set5_like_hr  # noqa: E402


def g15_spynet():
    """G15: the reference's vendored SpyNet (models/spynet_arch.py:28-96) at its seeded default init (torch.manual_seed(150); 1.44 M
    parameters are re-created from the seed by the tests, checksums stored) on two frame pairs: 64 x 64 (C4's size) and
    40 x 56 (resized to 64 x 64 inside, flows scaled back).  Stored: inputs, output flows, per-level module outputs of pair 0."""
    sys.path.insert(0, REF)
    import importlib
    sp = importlib.import_module("models.spynet_arch")
    torch.manual_seed(150)
    net = sp.SpyNet().eval()
    d = {"w_sum": np.float64(sum(v.double().sum().item() for k, v in net.state_dict().items())),
         "w_abs": np.float64(sum(v.double().abs().sum().item() for k, v in net.state_dict().items()))}
    g = torch.Generator().manual_seed(151)
    for k, (n, h, w) in enumerate([(2, 64, 64), (1, 40, 56)]):
        base = torch.rand(n, 3, h // 8 + 2, w // 8 + 2, generator=g)
        img = F_interp(base, (h + 8, w + 8))
        ref, supp = img[:, :, 4:-4, 4:-4].contiguous(), img[:, :, 2:-6, 5:-3].contiguous()      # the same scene shifted by (2, -1) px
        ref = (ref + 0.05 * torch.rand(ref.shape, generator=g)).clamp(0, 1)
        with torch.no_grad():
            flow = net(ref, supp)
        d[f"ref_{k}"], d[f"supp_{k}"], d[f"flow_{k}"] = _np(ref), _np(supp), _np(flow)
        print(f"G15 pair set {k}: flow mean {flow.mean().item():.4f} abs max {flow.abs().max().item():.4f}")
    np.savez_compressed(os.path.join(OUT, "g15_spynet.npz"), **d)


def F_interp(t, size):
    return torch.nn.functional.interpolate(t, size=size, mode="bicubic", align_corners=False)


def _absent_third_party_stubs():
    """sys.modules stand-ins for THIRD-PARTY packages that are not installed here (skimage, mmedit, torchvision, h5py-free
    paths): they carry no arithmetic of the reference.  `torchvision.transforms.functional.to_tensor` is restated for the
    one input kind the dataset path feeds it (H x W x C uint8 ndarray -> C x H x W float32 / 255: torchvision's documented
    behaviour); everything else raises if touched."""
    def mod(name, **attrs):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    def absent(*a, **k):
        raise RuntimeError("third-party function absent in this container (stub)")

    def to_tensor(pic):
        a = np.asarray(pic)
        assert a.dtype == np.uint8 and a.ndim == 3
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div(255)

    mod("skimage", img_as_float=absent, img_as_ubyte=absent)
    mod("skimage.io", imread=absent)
    mod("skimage.metrics", structural_similarity=absent)
    mod("mmedit")
    mod("mmedit.core")
    mod("mmedit.core.evaluation")
    mod("mmedit.core.evaluation.metrics", psnr=absent, ssim=absent)
    tf = mod("torchvision.transforms.functional", to_tensor=to_tensor)
    tt = mod("torchvision.transforms", functional=tf)
    mod("torchvision", transforms=tt)


def _reference_metrics():
    """the reference's own common/metrics.py (psnr, psnr_y), imported with skimage / mmedit stubbed"""
    _absent_third_party_stubs()
    import importlib
    if REF not in sys.path:
        sys.path.insert(0, REF)
    return importlib.import_module("common.metrics")


def g9_metrics():
    """G9: common/metrics.py:10-38 itself on the hand cases of tests/test_oracle_golden.py and on random images (batch > 1:
    the SUM over the batch; 1-channel input: no luma filter; shave 0 / 4 / 10)."""
    M = _reference_metrics()
    g = torch.Generator().manual_seed(90)
    d = {}
    cases = []
    for k, (n, c, h, w, shave) in enumerate([(1, 3, 40, 52, 4), (3, 3, 33, 47, 10), (2, 1, 24, 24, 4), (2, 3, 20, 28, 0)]):
        hr = torch.rand(n, c, h, w, generator=g)
        sr = (hr + 0.05 * torch.randn(n, c, h, w, generator=g)) * 1.1 - 0.03          # leaves [0,1] here and there: clamps matter
        cases.append((k, sr, hr, shave))
    for k, sr, hr, shave in cases:
        d[f"sr_{k}"], d[f"hr_{k}"], d[f"shave_{k}"] = _np(sr), _np(hr), np.int64(shave)
        d[f"psnr_{k}"] = np.float64(M.psnr(sr, hr, shave=shave).item())
        d[f"psnr_y_{k}"] = np.float64(M.psnr_y(sr, hr, shave=shave).item())
        print(f"G9 case {k}: psnr {d[f'psnr_{k}']:.5f} psnr_y {d[f'psnr_y_{k}']:.5f}")
    d["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "g9_metrics.npz"), **d)


def g14_patches():
    """G14: the reference's own training-item path, datasets/_isr.py:56-121 -- `ImageSuperResolutionDataset.__getitem__` in
    TRAIN mode (`_load_item` through PIL on PNG files written here, `_sample_patch`, `_augment`, `to_tensor`) under a seeded
    `random`.  Stored: the decoded images, the parameters, every item's LR / HR patch as uint8 (exact: to_tensor only divides
    by 255) and `random.random()` drawn right after the last item (pins the number of draws)."""
    import importlib
    import random
    import tempfile
    from PIL import Image
    _absent_third_party_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    isr = importlib.import_module("datasets._isr")
    modes = importlib.import_module("common.modes")
    d = {}
    cfgs = [(4, 12, 0, 3), (2, 16, 4, 1), (3, 11, 2, 2)]
    d["cfgs"] = np.array(cfgs, dtype=np.int64)
    with tempfile.TemporaryDirectory() as tmp:
        for ci, (scale, P, ignored, num_patches) in enumerate(cfgs):
            g = np.random.default_rng(140 + ci)
            lr_files, hr_files = [], []
            n_img = 3
            for k in range(n_img):
                h, w = int(g.integers(40, 64)), int(g.integers(40, 64))
                lr = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
                hr = g.integers(0, 256, (h * scale + (k % 2), w * scale + (k % 3), 3), dtype=np.uint8)
                lp, hp = os.path.join(tmp, f"c{ci}_lr{k}.png"), os.path.join(tmp, f"c{ci}_hr{k}.png")
                Image.fromarray(lr).save(lp)
                Image.fromarray(hr).save(hp)
                lr_files.append((f"{k}.png", lp))
                hr_files.append((f"{k}.png", hp))
                d[f"c{ci}_lr{k}"], d[f"c{ci}_hr{k}"] = lr, hr
            params = argparse.Namespace(scale=scale, lr_patch_size=P, ignored_boundary_size=ignored, num_patches=num_patches)
            ds = isr.ImageSuperResolutionDataset(modes.TRAIN, params, lr_files, hr_files)
            idx = (list(range(len(ds))) * 64)[:64]
            random.seed(1400 + ci)
            lrs, hrs = [], []
            for i in idx:
                a, b = ds[i]
                lrs.append((a * 255).round().to(torch.uint8).numpy())
                hrs.append((b * 255).round().to(torch.uint8).numpy())
                assert torch.equal(torch.from_numpy(lrs[-1]).float().div(255), a)
            d[f"c{ci}_idx"] = np.array(idx, dtype=np.int64)
            d[f"c{ci}_seed"] = np.int64(1400 + ci)
            d[f"c{ci}_lr_items"], d[f"c{ci}_hr_items"] = np.stack(lrs), np.stack(hrs)
            d[f"c{ci}_next_random"] = np.float64(random.random())
            d[f"c{ci}_n_img"] = np.int64(n_img)
            print(f"G14 cfg {ci} (scale {scale}, P {P}, ignored {ignored}, num_patches {num_patches}): {len(idx)} items")
    np.savez_compressed(os.path.join(OUT, "g14_patches.npz"), **d)


def g13_set5_shaped(BASIC_MODEL):
    """Set5 is absent offline, so PSNR parity is pinned on five synthetic images with Set5's shapes.  LR is
    made by the reference's own numpy MATLAB-imresize clone (third_party/matlab_imresize/imresize.py:104-136) after
    cropping HR to multiples of the scale as datasets/_isr.py:216-221 does.  Two golden networks, both the reference
    BASIC_MODEL class: "x2" = the shipped trained checkpoint models/pretrained_weights/wdsr_b_x2_8_24.pt (realistic
    ~30 dB outputs; its tensors are already in fixture G3) and "x4" = 16 blocks / 24 units at the seeded init
    (torch.manual_seed(130); no x4 checkpoint ships).  Stored per image and network: LR, psnr / psnr_y of the reference
    output against HR (utils/estimate.py:123-125 shaves), output mean / mean-abs, and a 4 x 4-strided sample of the
    output (the full outputs would be 18 MB); HR is re-created from its seed by the tests (`set5_like_hr`, checksum
    stored), the x4 weights from theirs (checksum stored)."""
    sys.path.insert(0, os.path.join(REF, "third_party", "matlab_imresize"))
    from imresize import imresize
    M = _reference_metrics()                                 # the reference's own common/metrics.py
    nets = {}
    m2 = BASIC_MODEL(_params(num_blocks=8, num_residual_units=24, scale=2)).eval()
    m2.load_state_dict(torch.load(os.path.join(REF, "models", "pretrained_weights", "wdsr_b_x2_8_24.pt"), map_location="cpu",
                                  weights_only=True), strict=True)
    nets["x2"] = (m2, 2)
    torch.manual_seed(130)
    nets["x4"] = (BASIC_MODEL(_params(num_blocks=16, num_residual_units=24, scale=4)).eval(), 4)
    d = {"x4_weight_sum": np.float64(sum(v.double().sum().item() for v in nets["x4"][0].state_dict().values())),
         "x4_weight_abs": np.float64(sum(v.double().abs().sum().item() for v in nets["x4"][0].state_dict().values()))}
    for tag, (m, r) in nets.items():
        for i, hw in enumerate(SET5_SHAPES):
            hr = set5_like_hr(i, hw)
            h, w = hw
            hr = hr[:, :h - h % r, :w - w % r]
            lr = imresize(hr.permute(1, 2, 0).double().numpy(), scalar_scale=1.0 / r)
            lr_t = torch.from_numpy(lr).permute(2, 0, 1)[None].half().float()         # stored (and consumed) at fp16 precision
            with torch.no_grad():
                sr = m(lr_t)
            k = f"{tag}_{i}"
            d["lr_" + k] = lr_t.half().numpy()
            d["sr_sample_" + k] = _np(sr[..., ::4, ::4])
            d["sr_mean_" + k] = np.float64(sr.double().mean().item())
            d["sr_abs_" + k] = np.float64(sr.double().abs().mean().item())
            d["psnr_" + k] = np.float64(M.psnr(sr, hr[None], shave=r + 6).item())         # utils/estimate.py:123: shave = scale + 6
            d["psnr_y_" + k] = np.float64(M.psnr_y(sr, hr[None], shave=r).item())         # :125: shave = scale
            if tag == "x2":
                d[f"hr_sum_{i}"] = np.float64(hr.double().sum().item())
            print(f"G13 {tag} image {i} {tuple(hr.shape)}: psnr {d['psnr_' + k]:.4f} psnr_y {d['psnr_y_' + k]:.4f}")
    np.savez_compressed(os.path.join(OUT, "g13_set5_shaped.npz"), **d)


def main():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, REF)
    torch.set_num_threads(4)
    import models  # noqa: the reference package
    from models.basic_wdsr_b import BASIC_MODEL, Block
    import models.ops as ops
    import models.wdsr_b as wdsr_b
    gens = {"g1": lambda: g1_model(BASIC_MODEL), "g2": lambda: g2_block(Block), "g3": lambda: g3_pretrained(BASIC_MODEL),
            "g4": g4_pixel_shuffle, "g5": lambda: g5_rounding(ops), "g6": lambda: g6_split_block(wdsr_b), "g7": g7_g8_vsr,
            "g10": g10_nas_model, "g11": g11_g12_video_models, "g13": lambda: g13_set5_shaped(BASIC_MODEL), "g9": g9_metrics,
            "g14": g14_patches, "g15": g15_spynet}
    only = [a for a in sys.argv[1:] if not a.startswith("-")]          # e.g. `make_golden.py g9 g14`: just those fixtures
    for name, fn in gens.items():
        if not only or name in only:
            fn()


if __name__ == "__main__":
    main()
