"""Product (HIP) path against fixtures taken from the reference's own whole-model classes -- what round 1 left "parity
unpinned": G10 NAS_MODEL (train + eval, one gated block, speed loss), G11 MotionVectorVSR through the module, G12
BasicVSR_origin incl. the PixelShuffle(2) x 2 upsampler, G13 Set5-shaped images (fp32 parity and the bf16 mode's PSNR
cost in dB), the seeded-init breadcrumb, and a bf16 whole-model gradient check at C2 size against the oracle."""
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


def _sd(d, prefix="p/"):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def _rel(got, exp):
    return (got.detach().cpu().float() - exp).abs().max().item() / max(exp.abs().max().item(), 1e-30)


# ------------------------------------------------------------------------------------------------------------------
def test_g10_nas_model_train_and_eval_match_reference(golden_dir):
    from mobilesuperresolution_amd.models import get_model
    d = _load(golden_dir, "g10_nas_model.npz")
    ns = argparse.Namespace(model_type="NAS_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=4,
                            num_residual_units=24, width_search=True, pretrained=False, hot_dtype="fp32")
    m = get_model(ns)
    missing, unexpected = m.load_state_dict(_sd(d), strict=False)
    assert not unexpected and all("speed_estimator" in k for k in missing), (missing, unexpected)
    m = m.cuda().train()
    out, speed = m(d["x"].cuda())
    assert _rel(out, d["out_train"]) <= 2e-5
    assert abs(speed.item() - d["speed_train"].item()) <= 1e-5 * abs(d["speed_train"].item())
    ori, tgt = float(d["ori_speed"]), float(d["speed_target"])
    l1 = torch.nn.functional.l1_loss(out, d["hr"].cuda())
    ls = O.speed_loss(speed, tgt, ori - tgt, 0.1)
    assert abs(l1.item() - d["loss_l1"].item()) <= 2e-6 and abs(ls.item() - d["loss_speed"].item()) <= 2e-6
    (l1 + ls).backward()
    worst, n = 0.0, 0
    for k, pg in m.named_reference_tensors(grads=True):    # the reference's tensors, as views of the flat parameter's gradient
        if "speed_estimator" in k:
            continue
        if "g/" + k in d:
            assert pg is not None, k
            e = _rel(pg, d["g/" + k])
            worst = max(worst, e)
            assert e <= 5e-4, (k, e)
            n += 1
        else:                                                # beta, beta1, beta2: no gradient in the reference either
            assert pg is None or float(pg.abs().max()) == 0.0, k
    assert n == 98
    print(f"\nG10 train: worst param-grad rel err {worst:.2e} over {n} tensors")
    for i in range(4):                                       # forward() rewrote the gates (wdsr_b.py:534)
        assert float(m.body[i].beta1) == float(d[f"after/body.{i}.beta1"]) and float(m.body[i].beta2) == float(d[f"after/body.{i}.beta2"])
    m.eval()
    with torch.no_grad():
        oe, se = m(d["x"].cuda())
    assert _rel(oe, d["out_eval"]) <= 2e-5
    assert abs(se.item() - d["speed_eval"].item()) <= 1e-5 * abs(d["speed_eval"].item())
    assert m.get_current_blocks() == int(d["current_blocks"]) and list(m.get_block_status()) == list(d["block_status"].numpy())


# ------------------------------------------------------------------------------------------------------------------
def _grads_match(model, d, tol):
    worst = 0.0
    flat_grads = {}
    for name in ("backward_trunk", "forward_trunk"):
        tr = getattr(model, name)
        for k, v in tr.named_tensors(tr.flat.grad):
            flat_grads[f"{name}.{k}"] = v
    for k, p in model.named_parameters():
        if not k.endswith(".flat"):
            flat_grads[k] = p.grad
    for k in [k[2:] for k in d if k.startswith("g/")]:
        assert flat_grads.get(k) is not None, k
        e = _rel(flat_grads[k], d["g/" + k])
        worst = max(worst, e)
        assert e <= tol, (k, e)
    return worst


def test_g11_motion_vector_vsr_module_matches_reference(golden_dir):
    """the trainer's 'basic_mv' model (num_feat = 20: embedded in the 24-wide kernels) THROUGH THE MODULE"""
    from mobilesuperresolution_amd.models import MotionVectorVSR
    d = _load(golden_dir, "g11_mvvsr.npz")
    m = MotionVectorVSR(num_feat=20, num_block=2, spynet_path=None, hot_dtype="fp32")
    m.load_state_dict(_sd(d), strict=True)
    m = m.cuda().train()
    x = d["x"].cuda().requires_grad_(True)
    b, n, _, h, w = x.shape
    feats = {"backward_trunk": [], "forward_trunk": []}
    hooks = [getattr(m, k).register_forward_hook(lambda mod, i, o, k=k: feats[k].append((o[0] if isinstance(o, tuple) else o).detach())) for k in feats]
    out = m(x, 4 * h, 4 * w)
    for hk in hooks:
        hk.remove()
    assert _rel(out, d["out"]) <= 5e-5
    assert _rel(torch.stack(feats["backward_trunk"], 1), d["feat_backward"]) <= 5e-5      # call order: frames n-1 .. 0
    assert _rel(torch.stack(feats["forward_trunk"], 1), d["feat_forward"]) <= 5e-5
    loss = O.charbonnier(out, d["target"].cuda())
    assert abs(loss.item() - d["loss"].item()) <= 2e-6
    loss.backward()
    assert _rel(x.grad, d["dx"]) <= 5e-4
    worst = _grads_match(m, d, 5e-4)
    print(f"\nG11 MotionVectorVSR: worst param-grad rel err {worst:.2e}")


def test_g12_basicvsr_origin_module_matches_reference(golden_dir):
    from mobilesuperresolution_amd.models import BasicVSR_origin
    d = _load(golden_dir, "g12_basicvsr_origin.npz")
    m = BasicVSR_origin(num_feat=24, num_block=2, spynet_path=None, hot_dtype="fp32")
    m.load_state_dict(_sd(d), strict=True)
    m = m.cuda().train()
    x = d["x"].cuda().requires_grad_(True)
    b, n, _, h, w = x.shape
    with pytest.raises(NotImplementedError, match="SPyNet"):
        m(x, 4 * h, 4 * w)                                   # no flows, 3-channel input: the out-of-scope prior is not faked
    out = m(x, 4 * h, 4 * w, flows=(d["flows_forward"].cuda(), d["flows_backward"].cuda()))
    assert _rel(out, d["out"]) <= 5e-5
    loss = O.charbonnier(out, d["target"].cuda())
    assert abs(loss.item() - d["loss"].item()) <= 2e-6
    loss.backward()
    assert _rel(x.grad, d["dx"]) <= 5e-4
    worst = _grads_match(m, d, 5e-4)
    print(f"\nG12 BasicVSR_origin: worst param-grad rel err {worst:.2e}")


@pytest.mark.parametrize("r", [2, 3, 4])
def test_standalone_pixel_shuffle_bit_exact(r, golden_dir):
    from mobilesuperresolution_amd.models import pixel_shuffle
    g = torch.Generator().manual_seed(40 + r)
    for shape in [(2, 3 * r * r, 5, 7), (1, 64 * r * r if r == 2 else 2 * r * r, 12, 16), (1, r * r, 1, 1)]:
        x = torch.randn(shape, generator=g).cuda().requires_grad_(True)
        y = pixel_shuffle(x, r)
        ref = torch.nn.functional.pixel_shuffle(x.detach(), r)
        assert torch.equal(y, ref)
        gy = torch.randn(ref.shape, generator=g).cuda()
        y.backward(gy)
        assert torch.equal(x.grad, torch.nn.functional.pixel_unshuffle(gy, r))
    d = _load(golden_dir, "g4_pixel_shuffle.npz")            # the reference's own nn.PixelShuffle on arange tensors
    assert torch.equal(pixel_shuffle(d[f"x_r{r}"].float().cuda(), r).cpu(), d[f"y_r{r}"].float())


# ------------------------------------------------------------------------------------------------------------------
def _basic(ns_kw, sd=None):
    from mobilesuperresolution_amd.models import get_model
    ns = argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, **ns_kw)
    m = get_model(ns)
    if sd is not None:
        m.load_state_dict(sd, strict=True)
    return m.cuda()


def test_seeded_init_breadcrumb(golden_dir):
    """torch.manual_seed(0); BASIC_MODEL(C1).eval()(torch.rand(1,3,48,48)).mean() == 0.80466092 -- the
    product draws the reference's random numbers in the reference's order"""
    d = _load(golden_dir, "g1_basic_model_c1.npz")
    torch.manual_seed(0)
    m = _basic(dict(scale=4, num_blocks=4, num_residual_units=24, hot_dtype="fp32")).eval()
    x = torch.rand(1, 3, 48, 48)
    with torch.no_grad():
        y = m(x.cuda())
    assert abs(y.mean().item() - float(d["breadcrumb_mean"])) <= 2e-6
    assert abs(float(d["breadcrumb_mean"]) - 0.80466092) <= 1e-7


@pytest.mark.parametrize("tag", ["x2", "x4"])
def test_g13_set5_shaped_fp32_parity_and_bf16_psnr_cost(golden_dir, tag):
    """fp32 mode: outputs equal the reference's on five Set5-shaped images (LR up to 256 x 256), |delta psnr| and
    |delta psnr_y| <= 1e-3 dB.  bf16 mode: its PSNR cost is STATED -- on the trained x2 checkpoint (outputs in [0, 1],
    ~32.6 dB) the bf16 path stays within 0.02 dB of the reference's psnr_y and above 50 dB against the reference's own
    output."""
    from oracle.set5_like import SET5_SHAPES, set5_like_hr
    d = _load(golden_dir, "g13_set5_shaped.npz")
    if tag == "x2":
        sd = _sd(_load(golden_dir, "g3_pretrained_x2_8_24.npz"))
        kw, r = dict(scale=2, num_blocks=8, num_residual_units=24), 2
    else:
        torch.manual_seed(130)
        sd, kw, r = None, dict(scale=4, num_blocks=16, num_residual_units=24), 4
    m32 = _basic(dict(kw, hot_dtype="fp32"), sd).eval()
    m16 = _basic(dict(kw, hot_dtype="bf16"), m32.state_dict()).eval()
    worst = [0.0, 0.0, 1e9]
    for i, hw in enumerate(SET5_SHAPES):
        hr = set5_like_hr(i, hw)
        hr = hr[:, :hw[0] - hw[0] % r, :hw[1] - hw[1] % r][None]
        k = f"{tag}_{i}"
        lr = d["lr_" + k].float().cuda()
        with torch.no_grad():
            sr32 = m32(lr).cpu()
            sr16 = m16(lr).cpu()
        assert _rel(sr32[..., ::4, ::4], d["sr_sample_" + k]) <= 5e-5, k
        assert abs(sr32.double().mean().item() - float(d["sr_mean_" + k])) <= 5e-6
        dp = abs(O.psnr(sr32, hr, shave=r + 6).item() - float(d["psnr_" + k]))
        dpy = abs(O.psnr_y(sr32, hr, shave=r).item() - float(d["psnr_y_" + k]))
        assert dp <= 1e-3 and dpy <= 1e-3, (k, dp, dpy)
        if tag == "x2":
            d16 = abs(O.psnr_y(sr16, hr, shave=r).item() - float(d["psnr_y_" + k]))
            self_psnr = O.psnr_y(sr16, sr32.clamp(0, 1), shave=r).item()
            worst = [max(worst[0], d16), max(worst[1], dpy), min(worst[2], self_psnr)]
            assert d16 <= 0.02 and self_psnr >= 50.0, (k, d16, self_psnr)
    if tag == "x2":
        print(f"\nG13 x2: fp32 worst |d psnr_y| {worst[1]:.2e} dB; bf16 worst |d psnr_y| {worst[0]:.4f} dB, "
              f"bf16-vs-fp32 psnr_y >= {worst[2]:.1f} dB")


def test_bf16_whole_model_gradient_against_oracle_at_c2_size():
    """C2's shape (16 blocks / 24 units / 48x48, batch 8 to keep the CPU oracle short): bf16-mode parameter gradient
    against the fp32 oracle's, relative L2 error <= 3 % over the whole flat gradient and <= 6 % per tensor"""
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(7)
    ns = argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=16,
                            num_residual_units=24, hot_dtype="bf16")
    m = get_model(ns)
    ref = O.OracleBasicModel(ns)
    ref.load_state_dict(m.state_dict(), strict=True)
    m = m.cuda().train()
    g = torch.Generator().manual_seed(8)
    x = torch.rand(8, 3, 48, 48, generator=g)
    hr = torch.rand(8, 3, 192, 192, generator=g)
    torch.nn.functional.l1_loss(m(x.cuda()), hr.cuda()).backward()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    torch.nn.functional.l1_loss(ref(x), hr).backward()
    gf = m.flat.grad.cpu()
    rg = dict(ref.named_parameters())
    num = den = 0.0
    worst = ("", 0.0)
    for k, (off, shape) in m.layout.entries.items():
        a, b = gf[off:off + rg[k].numel()].view(shape), rg[k].grad
        num += (a - b).pow(2).sum().item()
        den += b.pow(2).sum().item()
        e = ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
        if e > worst[1]:
            worst = (k, e)
    tot = (num / den) ** 0.5
    print(f"\nbf16 whole-model gradient vs oracle: L2 rel {tot:.3e}; worst tensor {worst[0]} {worst[1]:.3e}")
    assert tot <= 3e-2 and worst[1] <= 6e-2
