"""Parity of the NAS supernet pieces on the HIP path: BinaryConv2d/rounding (G5), Split_Block.forward_body
with all gradients (G6, golden vectors from the reference), and the NAS_MODEL composition against the
oracle's restated glue (whole-model composition: parity unpinned, see oracle.nas_model_forward)."""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import wdsr_oracle as O

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def test_g5_rounding_and_effective_mask(golden_dir):
    from mobilesuperresolution_amd.models.ops import BinaryConv2d, rounding
    d = _load(golden_dir, "g5_binary_mask.npz")
    for name in ("all_keep", "straddle", "fallback", "ties"):
        w = d[f"{name}/w"]
        for lc in (8, 0):
            assert torch.equal(rounding(w, lc), d[f"{name}/mask_lc{lc}"])
        m = BinaryConv2d(24, 24, groups=24, least_channel=8)
        with torch.no_grad():
            m.weight.copy_(w)
        eff = m.effective()
        assert torch.equal(eff.detach(), d[f"{name}/mask_lc8"].reshape(-1))
        # forward value x * mask and straight-through gradient, vs the reference's conv
        x = d[f"{name}/x"]
        y = x * eff.view(1, -1, 1, 1)
        assert torch.equal(y.detach(), d[f"{name}/y"])
        y.backward(d[f"{name}/dy"])
        torch.testing.assert_close(m.weight.grad, d[f"{name}/dw"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("f", [24, 32])
def test_g6_split_block_fp32_matches_reference(golden_dir, f):
    from mobilesuperresolution_amd.models.wdsr_b import Split_Block
    d = _load(golden_dir, f"g6_split_block_f{f}.npz")
    blk = Split_Block(num_residual_units=f, kernel_size=3)
    sd = {k[2:]: v for k, v in d.items() if k.startswith("p/")}
    assert set(blk.state_dict().keys()) == set(sd.keys())
    blk.load_state_dict(sd, strict=True)
    blk = blk.cuda()
    x = d["x"].cuda().requires_grad_(True)
    y = blk.forward_body(x)
    err = (y.detach().cpu() - d["y"]).abs().max().item() / d["y"].abs().max().item()
    print(f"\nG6 F={f} fwd rel err {err:.2e}")
    assert err <= 1e-5
    y.backward(d["dy"].cuda())
    e = (x.grad.cpu() - d["dx"]).abs().max().item() / d["dx"].abs().max().item()
    print(f"G6 F={f} dx rel err {e:.2e}")
    assert e <= 1e-5
    worst = 0.0
    for k, p in blk.named_parameters():
        if "g/" + k not in d:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k      # beta: unused
            continue
        exp = d["g/" + k]
        ge = (p.grad.cpu() - exp).abs().max().item() / max(exp.abs().max().item(), 1e-12)
        worst = max(worst, ge)
        assert ge <= 2e-4, (k, ge)
    print(f"G6 F={f} worst param-grad rel err {worst:.2e}")


def _nas_ns(**kw):
    ns = argparse.Namespace(model_type="NAS_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=3,
                            num_residual_units=24, width_search=True, pretrained=False, hot_dtype="fp32")
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


@pytest.mark.parametrize("training", [True, False])
def test_nas_model_matches_oracle_glue(training):
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(3)
    m = get_model(_nas_ns())
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        m.mask.weight.copy_(torch.rand(24, 1, 1, 1, generator=g) * 0.7 + 0.25)      # some global channels off
        for i, blk in enumerate(m.body):
            blk.split.weight.copy_(torch.rand(24, 1, 1, 1, generator=g) * 0.7 + 0.2)
        m.body[1].alpha1.fill_(1.5)                                                 # block 1 is skipped (alpha1 >= alpha2)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    m = m.cuda().train(training)
    x = torch.rand(2, 3, 20, 28, generator=g)
    hr = torch.rand(2, 3, 80, 112, generator=g)
    out, speed = m(x.cuda())
    ref, rspeed = O.nas_model_forward(x, sd, 4, 0.5, training)
    err = (out.detach().cpu() - ref.detach()).abs().max().item() / ref.abs().max().item()
    print(f"\nNAS model ({'train' if training else 'eval'}) fwd rel err {err:.2e}; speed {speed.item():.4f} vs {rspeed.item():.4f}")
    assert err <= 2e-5
    assert abs(speed.item() - rspeed.item()) <= 1e-4 * abs(rspeed.item())
    assert m.get_current_blocks() == 2 and m.get_block_status() == [0, 2]
    if not training:
        return
    (torch.nn.functional.l1_loss(out, hr.cuda()) + 0.1 * speed.sum()).backward()
    (torch.nn.functional.l1_loss(ref, hr) + 0.1 * rspeed.sum()).backward()
    worst = 0.0
    for k, pg in m.named_reference_tensors(grads=True):
        rg = sd[k].grad
        if rg is None or float(rg.abs().max()) == 0.0:
            assert pg is None or float(pg.abs().max()) <= 1e-7, k
            continue
        ge = (pg.cpu() - rg).abs().max().item() / rg.abs().max().item()
        worst = max(worst, ge)
        assert ge <= 5e-4, (k, ge)
    print(f"NAS model worst param-grad rel err {worst:.2e}")


@pytest.mark.parametrize("units", [24, 32])
def test_nas_bf16_mode_tracks_fp32_mode(units):
    """throughput mode of the supernet (bf16 activations, depthwise stencils on packed bf16 dot products with lane = channel,
    pointwise convs on the matrix cores) against the exact fp32 mode on the same parameters: output and every gradient
    within bf16 tolerance, ragged image, masks partly off, one skipped block"""
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(9)
    a = get_model(_nas_ns(num_residual_units=units, hot_dtype="fp32"))
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        a.mask.weight.copy_(torch.rand(units, 1, 1, 1, generator=g) * 0.7 + 0.25)
        for blk in a.body:
            blk.split.weight.copy_(torch.rand(units, 1, 1, 1, generator=g) * 0.7 + 0.2)
    b = get_model(_nas_ns(num_residual_units=units, hot_dtype="bf16"))
    b.load_state_dict(a.state_dict())
    a, b = a.cuda().train(), b.cuda().train()
    x = torch.rand(2, 3, 29, 50, generator=g).cuda()
    hr = torch.rand(2, 3, 116, 200, generator=g).cuda()
    outs = []
    for m in (a, b):
        out, speed = m(x)
        (torch.nn.functional.l1_loss(out, hr) + 0.1 * speed.sum()).backward()
        outs.append((out.detach(), speed.detach()))
    rel = ((outs[1][0] - outs[0][0]).norm() / outs[0][0].norm()).item()
    assert rel <= 2e-2, rel
    assert abs(outs[1][1].item() - outs[0][1].item()) <= 1e-5 * abs(outs[0][1].item())
    ga, gb = a.flat.grad, b.flat.grad
    grel = ((gb - ga).norm() / ga.norm()).item()
    print(f"\nNAS F={units} bf16 vs fp32 mode: out rel L2 {rel:.2e}, flat grad rel L2 {grel:.2e}")
    assert grel <= 6e-2, grel
    worst = 0.0
    for (k, pa), (_, pb) in zip(a.named_reference_tensors(grads=True), b.named_reference_tensors(grads=True)):
        if pa is None or float(pa.abs().max()) == 0.0:
            continue
        e = ((pb - pa).norm() / pa.norm()).item()
        worst = max(worst, e)
        assert e <= 0.06, (k, e)
    print(f"worst per-tensor grad rel L2 {worst:.2e}")


@pytest.mark.gpu
@pytest.mark.parametrize("units", [24, 32])
def test_nas_scalars_kernel_matches_torch_formulas(units):
    """sr_nas_scalars (one launch) against the reference's formulas as torch ops: rounding() with its top-8 fallback and ties
    (models/ops.py:33-43), the gates (wdsr_b.py:517-534) and the latency head (speed_estimator.py:57-76); bit-exact masks."""
    from mobilesuperresolution_amd import _lib as L
    from mobilesuperresolution_amd.models.ops import rounding
    torch.manual_seed(5)
    nb, f = 16, units
    dev = torch.device("cuda", 0)
    for case in range(6):
        scale = (1.0, 0.55, 0.4, 1.0, 0.3, 0.7)[case]
        mask_w = (torch.rand(f, 1, 1, 1, device=dev) * scale).contiguous()
        split = (torch.rand(nb, f, device=dev) * scale).contiguous()
        if case >= 3:                                   # ties at the 8th value, exact 0.5s
            split = torch.round(split * 8) / 8
            mask_w = torch.round(mask_w * 8) / 8
        alpha = torch.rand(nb, 3, device=dev)
        a1, a2 = torch.rand(nb, device=dev), torch.rand(nb, device=dev)
        a2[::3] = a1[::3]                               # alpha1 == alpha2 -> gate (1, 0)
        out = torch.empty(f + 1 + nb * (f + 4), device=dev)
        src = torch.full((nb, 3 * f + 7), -7.0, device=dev)                # mask columns at offset 3 of rows of 3 F + 7
        scal = torch.empty(nb, 4, device=dev)
        L.check(L.lib().sr_nas_scalars(mask_w.data_ptr(), split.data_ptr(), alpha.data_ptr(), a1.data_ptr(), a2.data_ptr(), nb, f,
                                       out.data_ptr(), src.data_ptr(), src.stride(0), 3, scal.data_ptr(), L.stream_ptr()), "scalars")
        mh = rounding(mask_w)
        assert torch.equal(out[:f], mh.reshape(-1))
        assert float(out[f]) == float(mh.sum())
        o = f + 1
        assert torch.equal(out[o:o + nb * f].view(nb, f), (split >= 0.5).float())
        cs = torch.stack([rounding(split[b].view(f, 1, 1, 1)).sum() for b in range(nb)])
        assert torch.equal(out[o + nb * f:o + nb * (f + 1)], cs)
        sp = ((cs + 0.2 * mh.sum()).view(-1, 1) * torch.tensor([9.0, 25.0, 49.0], device=dev).view(1, 3) * alpha / 40).sum(1)
        torch.testing.assert_close(out[o + nb * (f + 1):o + nb * (f + 2)], sp, rtol=1e-6, atol=1e-6)
        g1 = (a1 >= a2).float()
        assert torch.equal(out[o + nb * (f + 2):].view(nb, 2), torch.stack([g1, 1 - g1], dim=1))
        mg, ms = mh.reshape(1, f).expand(nb, f), (split >= 0.5).float()
        expect = torch.cat([mg, ms, mg * ms, torch.zeros(nb, 1, device=dev), torch.ones(nb, 1, device=dev)], dim=1)
        assert torch.equal(src[:, 3:3 + 3 * f + 2], expect) and bool((src[:, :3] == -7).all()) and bool((src[:, 3 * f + 5:] == -7).all())
        torch.testing.assert_close(scal[:, :3], torch.softmax(alpha, dim=1), rtol=1e-6, atol=1e-7)
        assert torch.equal(scal[:, 3], 1 - g1)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_nas_native_plumbing_equals_torch_route(monkeypatch, dtype):
    """sr_param_pack / sr_param_grads / sr_nas_scalars (the default route of a training step) against the same step with the
    weight-norm, gathers, slab sums, masks and gates as torch ops (SR_NAS_TORCH_PREP=1): outputs and gradients agree to
    rounding / summation order (fp32) or to one bf16 rounding of a weight (bf16)."""
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(21)
    m = get_model(_nas_ns(num_blocks=4, num_residual_units=32, hot_dtype=dtype))
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        m.mask.weight.copy_(torch.rand(32, 1, 1, 1, generator=g) * 0.7 + 0.25)
        for blk in m.body:
            blk.split.weight.copy_(torch.rand(32, 1, 1, 1, generator=g) * 0.55 + 0.1)   # rows with fewer than 8 channels >= 0.5 too
        m.body[2].alpha1.fill_(1.5)
    m = m.cuda().train()
    x = torch.rand(2, 3, 24, 36, generator=g).cuda()
    hr = torch.rand(2, 3, 96, 144, generator=g).cuda()
    res = []
    for torch_route in (False, True):
        if torch_route:
            monkeypatch.setenv("SR_NAS_TORCH_PREP", "1")
        else:
            monkeypatch.delenv("SR_NAS_TORCH_PREP", raising=False)
        m.zero_grad(set_to_none=True)
        out, speed = m(x)
        (torch.nn.functional.l1_loss(out, hr) + 0.1 * speed.sum()).backward()
        res.append((out.detach().clone(), speed.detach().clone(), {k: v.clone() for k, v in m.named_reference_tensors(grads=True)
                                                                   if v is not None},
                    m.mask.weight.grad.clone()))
    (o0, s0, g0, mg0), (o1, s1, g1, mg1) = res
    # (the two weight-norm kernels round differently in the last bit, so not torch.equal)
    tol = 2e-6 if dtype == "fp32" else 1e-2
    assert float((o0 - o1).abs().max()) <= tol * float(o1.abs().max())
    torch.testing.assert_close(s0, s1, rtol=1e-6, atol=0)
    assert set(g0) == set(g1)
    for k in g0:
        scale = max(float(g1[k].abs().max()), 1e-12)
        assert float((g0[k] - g1[k]).abs().max()) <= (2e-5 if dtype == "fp32" else 2e-2) * scale, k
    torch.testing.assert_close(mg0, mg1, rtol=1e-4 if dtype == "fp32" else 2e-2, atol=1e-7 if dtype == "fp32" else 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("units", [24, 32])
def test_nas_fused_block_forward_is_bit_identical_to_two_kernels(monkeypatch, units):
    """nas_block_fwd_kernel (depthwise + pointwise of a block in one launch, V through an LDS tile) against the two-kernel
    route (SR_NAS_FWD_SPLIT=1): output, saved V (through the gradients) -- ragged image, masks partly off"""
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(31)
    m = get_model(_nas_ns(num_blocks=3, num_residual_units=units, hot_dtype="bf16"))
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        m.mask.weight.copy_(torch.rand(units, 1, 1, 1, generator=g) * 0.7 + 0.25)
        for blk in m.body:
            blk.split.weight.copy_(torch.rand(units, 1, 1, 1, generator=g) * 0.7 + 0.2)
    m = m.cuda().train()
    x = torch.rand(3, 3, 29, 50, generator=g).cuda()
    hr = torch.rand(3, 3, 116, 200, generator=g).cuda()
    res = []
    for split in (False, True):
        if split:
            monkeypatch.setenv("SR_NAS_FWD_SPLIT", "1")
        else:
            monkeypatch.delenv("SR_NAS_FWD_SPLIT", raising=False)
        m.zero_grad(set_to_none=True)
        out, speed = m(x)
        (torch.nn.functional.l1_loss(out, hr) + 0.1 * speed.sum()).backward()
        res.append((out.detach().clone(), m.flat.grad.clone(), m.mask.weight.grad.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


@pytest.mark.gpu
@pytest.mark.parametrize("units", [24, 32])
def test_nas_fused_block_backward_matches_separate_kernels(monkeypatch, units):
    """nas_block_bwd_a_kernel (pointwise backward + depthwise weight gradients in one launch, GZ handed over through the LDS
    tile) against the separate kernels (SR_NAS_BWD_SPLIT=1): the data gradient (through GZ) and the depthwise weight
    gradients are bit-identical, the pointwise sums agree to summation order -- ragged image, masks partly off"""
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(33)
    m = get_model(_nas_ns(num_blocks=3, num_residual_units=units, hot_dtype="bf16"))
    g = torch.Generator().manual_seed(10)
    with torch.no_grad():
        m.mask.weight.copy_(torch.rand(units, 1, 1, 1, generator=g) * 0.7 + 0.25)
        for blk in m.body:
            blk.split.weight.copy_(torch.rand(units, 1, 1, 1, generator=g) * 0.7 + 0.2)
    m = m.cuda().train()
    x = torch.rand(3, 3, 29, 50, generator=g).cuda().requires_grad_(True)
    hr = torch.rand(3, 3, 116, 200, generator=g).cuda()
    res = []
    for split in (False, True):
        if split:
            monkeypatch.setenv("SR_NAS_BWD_SPLIT", "1")
        else:
            monkeypatch.delenv("SR_NAS_BWD_SPLIT", raising=False)
        m.zero_grad(set_to_none=True)
        x.grad = None
        out, speed = m(x)
        (torch.nn.functional.l1_loss(out, hr) + 0.1 * speed.sum()).backward()
        res.append((dict((k, v.clone()) for k, v in m.named_reference_tensors(grads=True) if v is not None), m.head.weight_v.grad.clone()))
    (ga, ha), (gb, hb) = res
    assert torch.equal(ha, hb)                          # the head's gradient has passed through every block's data gradient
    assert set(ga) == set(gb)
    for k in ga:
        if ".body.0.weight_v" in k or ".body.0.weight_g" in k:       # depthwise convs: same MFMA sequence
            scale = max(float(gb[k].abs().max()), 1e-12)
            assert float((ga[k] - gb[k]).abs().max()) <= 1e-6 * scale, k
        else:
            scale = max(float(gb[k].abs().max()), 1e-12)
            assert float((ga[k] - gb[k]).abs().max()) <= 2e-5 * scale, k


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,weight", [("bf16", 1.0), ("bf16", 0.37), ("fp32", 0.37)])
def test_nas_search_loop_with_the_drop_in_criterion(dtype, weight):
    """search.py:72-89 unchanged -- `loss = w * criterions['l1'](sr, hr) + speed term; loss.backward()` -- with
    mobilesuperresolution_amd.training.L1Loss in place of nn.L1Loss: the loss is folded into the tail-backward kernel (no
    d(loss)/d(sr) tensor).  Same loss value and the same gradients as with nn.L1Loss on the same model: exactly for w = 1 in the
    fp32 parameters' gradients of the tail (same fp32 products), within bf16 rounding elsewhere (the data gradient is
    rounded to bf16 before the weight is applied instead of after)"""
    from mobilesuperresolution_amd.models import get_model
    from mobilesuperresolution_amd.training import L1Loss
    torch.manual_seed(4)
    m = get_model(_nas_ns(num_residual_units=24, hot_dtype=dtype)).cuda().train()
    g = torch.Generator().manual_seed(8)
    x = torch.rand(2, 3, 24, 40, generator=g).cuda()
    hr = torch.rand(2, 3, 96, 160, generator=g).cuda()
    res = []
    for crit in (torch.nn.L1Loss(), L1Loss()):
        m.zero_grad(set_to_none=True)
        sr, speed = m(x)
        loss = weight * crit(sr, hr) + 0.1 * speed.sum()
        loss.backward()
        res.append((loss.detach().clone(), {k: (None if t is None else t.clone()) for k, t in m.named_reference_tensors(grads=True)}))
    (l0, g0), (l1, g1) = res
    assert abs(l1.item() - l0.item()) <= 1e-6 * abs(l0.item()) + 1e-7, (l0.item(), l1.item())
    tol = 2e-6 if dtype == "fp32" else (1e-6 if weight == 1.0 else 1e-2)
    worst = 0.0
    for k in g0:
        a, b = g0[k], g1[k]
        if a is None or float(a.abs().max()) == 0.0:
            assert b is None or float(b.abs().max()) <= 1e-7, k
            continue
        e = ((b - a).norm() / a.norm()).item()
        worst = max(worst, e)
        assert e <= tol, (k, e)
    print(f"\nNAS drop-in criterion ({dtype}, w = {weight}): worst per-tensor grad rel L2 {worst:.2e}")
    # a criterion on something else than the model's own output takes torch's ops
    sr, speed = m(x)
    other = L1Loss()(sr[:, :, ::2], hr[:, :, ::2])
    assert torch.allclose(other, torch.nn.functional.l1_loss(sr[:, :, ::2], hr[:, :, ::2]))
