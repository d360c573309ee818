"""Pin oracle/wdsr_oracle.py (the CPU restatement) against golden vectors that
oracle/make_golden.py captured from the reference itself.
CPU only."""
import argparse
import math
import os

import numpy as np
import pytest
import torch

from oracle import wdsr_oracle as O


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _close(got, exp, rel=3e-6, msg=None):
    """fp32 restatement vs reference: agree to a few ulps of the tensor's scale
    (the only arithmetic difference is the rounding order inside weight-norm)."""
    scale = max(float(exp.abs().max()), 1e-30)
    err = float((got.detach() - exp).abs().max())
    assert err <= rel * scale, f"{msg or ''} max|diff| {err:.3e} > {rel:g} * {scale:.3e}"


def _sd(d, prefix="p/"):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def test_g1_model_forward_backward(golden_dir):
    d = _load(golden_dir, "g1_basic_model_c1.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in _sd(d).items()}
    x = d["x"].clone().requires_grad_(True)
    y = O.basic_model_forward(x, sd, scale=4, image_mean=0.5)
    assert y.shape == (1, 3, 192, 192)
    _close(y, d["y"])
    loss = torch.nn.functional.l1_loss(y, d["hr"])
    torch.testing.assert_close(loss, d["loss"], rtol=1e-6, atol=1e-7)
    loss.backward()
    _close(x.grad, d["dx"], rel=1e-4)
    for k, p in sd.items():
        _close(p.grad, d["g/" + k], rel=1e-4, msg=k)


def test_g1_module_breadcrumb_and_keys(golden_dir):
    d = _load(golden_dir, "g1_basic_model_c1.npz")
    ns = argparse.Namespace(image_mean=0.5, num_channels=3, scale=4, num_blocks=4, num_residual_units=24)
    m = O.OracleBasicModel(ns)
    assert set(m.state_dict().keys()) == set(_sd(d).keys())
    assert sum(p.numel() for p in m.parameters()) == 58984
    m.load_state_dict(_sd(d), strict=True)
    y = m(d["x"])
    _close(y, d["y"])
    # init values restated from basic_wdsr_b.py:40,62,75,115,126,136
    m0 = O.OracleBasicModel(ns)
    assert float(m0.body[0].body[0].weight_g[0]) == 2.0
    assert float(m0.body[0].body[3].weight_g[0]) == pytest.approx(1 / math.sqrt(4))
    assert float(m0.head.weight_g[0]) == 1.0 and float(m0.tail.bias.abs().sum()) == 0.0


@pytest.mark.parametrize("f", [24, 32])
def test_g2_block(golden_dir, f):
    d = _load(golden_dir, f"g2_block_f{f}.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in _sd(d).items()}
    x = d["x"].clone().requires_grad_(True)
    y, h, t, r = O.block_forward(x, sd, "body", return_intermediates=True)
    for got, key in ((y, "y"), (h, "h"), (t, "t"), (r, "r")):
        _close(got, d[key], msg=key)
    y.backward(d["dy"])
    _close(x.grad, d["dx"], rel=1e-5)
    for k, p in sd.items():
        _close(p.grad, d["g/" + k], rel=1e-4, msg=k)


def test_g3_pretrained_x2(golden_dir):
    d = _load(golden_dir, "g3_pretrained_x2_8_24.npz")
    sd = _sd(d)
    assert len(sd) == 81                                             # SURVEY section 2 "pretrained weights"
    y = O.basic_model_forward(d["x"], sd, scale=2)
    _close(y, d["y"])
    ns = argparse.Namespace(image_mean=0.5, num_channels=3, scale=2, num_blocks=8, num_residual_units=24)
    m = O.OracleBasicModel(ns)
    m.load_state_dict(sd, strict=True)


def test_g4_pixel_shuffle_bit_exact(golden_dir):
    d = _load(golden_dir, "g4_pixel_shuffle.npz")
    for r in (2, 3, 4):
        assert torch.equal(O.pixel_shuffle(d[f"x_r{r}"], r), d[f"y_r{r}"])


def test_g5_rounding_and_binary_mask(golden_dir):
    d = _load(golden_dir, "g5_binary_mask.npz")
    for name in ("all_keep", "straddle", "fallback", "ties"):
        w = d[f"{name}/w"]
        for lc in (8, 0):
            assert torch.equal(O.rounding(w, lc), d[f"{name}/mask_lc{lc}"]), (name, lc)
        wp = w.clone().requires_grad_(True)
        x = d[f"{name}/x"].clone().requires_grad_(True)
        y = O.binary_mask_forward(x, wp, 8)
        assert torch.equal(y, d[f"{name}/y"])
        y.backward(d[f"{name}/dy"])
        torch.testing.assert_close(x.grad, d[f"{name}/dx"], rtol=0, atol=0)
        torch.testing.assert_close(wp.grad, d[f"{name}/dw"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("f", [24, 32])
def test_g6_split_block(golden_dir, f):
    d = _load(golden_dir, f"g6_split_block_f{f}.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in _sd(d).items()}
    x = d["x"].clone().requires_grad_(True)
    y = O.split_block_forward_body(x, sd)
    _close(y, d["y"])
    y.backward(d["dy"])
    _close(x.grad, d["dx"], rel=1e-5)
    for k in d:
        if k.startswith("g/"):
            _close(sd[k[2:]].grad, d[k], rel=1e-4, msg=k)
    # beta is never used in forward: the reference leaves its grad None
    assert "g/beta" not in d and sd["beta"].grad is None


def test_g7_vsr_trunk(golden_dir):
    d = _load(golden_dir, "g7_vsr_trunk.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in _sd(d).items()}
    x = d["x"].clone().requires_grad_(True)
    y = O.conv_residual_blocks_forward(x, sd, "main")
    _close(y, d["y"])
    y.backward(d["dy"])
    _close(x.grad, d["dx"], rel=1e-5)
    for k, p in sd.items():
        _close(p.grad, d["g/" + k], rel=1e-4, msg=k)


def test_g8_flow_warp(golden_dir):
    d = _load(golden_dir, "g8_flow_warp.npz")
    x = d["x"].clone().requires_grad_(True)
    fl = d["flow"].clone().requires_grad_(True)
    y = O.flow_warp(x, fl)
    torch.testing.assert_close(y, d["y"], rtol=0, atol=1e-6)
    y.backward(d["dy"])
    torch.testing.assert_close(x.grad, d["dx"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(fl.grad, d["dflow"], rtol=1e-4, atol=1e-5)


def test_g9_psnr_hand_cases():
    """hand-computed cases; the reference's own values: test_g9_psnr_matches_reference_metrics"""
    hr = torch.full((1, 3, 16, 16), 0.5)
    sr = hr + 1.0 / 255.0
    # quantise: (0.5+1/255)*255 = 128.5 -> round-half-even 128 -> 128/255; diff = 128/255-0.5 = 0.5/255
    exp = -10 * math.log10((0.5 / 255) ** 2)
    assert float(O.psnr(sr, hr, shave=4)) == pytest.approx(exp, abs=1e-3)
    sr2 = hr + 2.0 / 255.0      # 129.5 -> 130 (half-even) -> diff 2.5/255
    assert float(O.psnr(sr2, hr, shave=4)) == pytest.approx(-10 * math.log10((2.5 / 255) ** 2), abs=1e-3)
    # psnr_y: no quantisation (the reference drops `r`), luma weights on the difference
    e = 1.0 / 255.0
    exp_y = -10 * math.log10(((0.257 + 0.504 + 0.098) * e) ** 2)
    assert float(O.psnr_y(hr + e, hr, shave=4)) == pytest.approx(exp_y, abs=1e-3)
    # batch is summed, not averaged (metrics.py:19)
    assert float(O.psnr_y(torch.cat([hr + e] * 3), torch.cat([hr] * 3), shave=2)) == pytest.approx(3 * exp_y, abs=3e-3)
    # clamp to [0,1] before the difference
    assert float(O.psnr_y(torch.full_like(hr, 1.5), torch.ones_like(hr) - e, shave=0)) == pytest.approx(exp_y, abs=1e-3)


def test_g9_psnr_matches_reference_metrics(golden_dir):
    """G9 (round 3): psnr / psnr_y values written by the reference's OWN common/metrics.py (imported by
    oracle/make_golden.py with skimage / mmedit stubbed) pin the oracle's restatement: clamps, batch sum, 1-channel input,
    shave 0 / 4 / 10."""
    d = _load(golden_dir, "g9_metrics.npz")
    for k in range(int(d["n_cases"])):
        sr, hr, shave = d[f"sr_{k}"], d[f"hr_{k}"], int(d[f"shave_{k}"])
        assert float(O.psnr(sr, hr, shave=shave)) == pytest.approx(float(d[f"psnr_{k}"]), abs=1e-4), k
        assert float(O.psnr_y(sr, hr, shave=shave)) == pytest.approx(float(d[f"psnr_y_{k}"]), abs=1e-4), k


def test_g14_patch_oracle_matches_reference_dataset_items(golden_dir):
    """G14: items of the reference's own ImageSuperResolutionDataset.__getitem__ (TRAIN; datasets/_isr.py:56-121) under a
    seeded `random` pin oracle/patch_oracle.py: crop origins, flip / swap order, HR crop at scale, number of RNG draws."""
    import random
    from oracle import patch_oracle as PO
    z = np.load(os.path.join(golden_dir, "g14_patches.npz"))
    for ci, (scale, P, ignored, num_patches) in enumerate(z["cfgs"].tolist()):
        n_img = int(z[f"c{ci}_n_img"])
        lrs = [z[f"c{ci}_lr{k}"] for k in range(n_img)]
        hrs = [z[f"c{ci}_hr{k}"] for k in range(n_img)]
        rng = random.Random(int(z[f"c{ci}_seed"]))
        for b, i in enumerate(z[f"c{ci}_idx"].tolist()):
            lr, hr = PO.train_item(lrs, hrs, i, P, scale, ignored, num_patches, rng)
            assert np.array_equal(lr, z[f"c{ci}_lr_items"][b].astype(np.float32) / np.float32(255)), (ci, b)
            assert np.array_equal(hr, z[f"c{ci}_hr_items"][b].astype(np.float32) / np.float32(255)), (ci, b)
        assert rng.random() == float(z[f"c{ci}_next_random"])


# ---- G10: the whole reference NAS_MODEL pins the oracle's model-level glue ----
def test_g10_nas_model_glue_matches_reference(golden_dir):
    d = _load(golden_dir, "g10_nas_model.npz")
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in _sd(d).items()}
    out, speed = O.nas_model_forward(d["x"], sd, 4, 0.5, training=True)
    _close(out, d["out_train"])
    _close(speed, d["speed_train"], rel=1e-6)
    ori, tgt = float(d["ori_speed"]), float(d["speed_target"])
    l1 = torch.nn.functional.l1_loss(out, d["hr"])
    ls = O.speed_loss(speed, tgt, ori - tgt, 0.1)
    torch.testing.assert_close(l1, d["loss_l1"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(ls, d["loss_speed"], rtol=1e-6, atol=1e-7)
    (l1 + ls).backward()
    n_checked = 0
    for k, p in sd.items():
        if "g/" + k in d:
            assert p.grad is not None, k
            _close(p.grad, d["g/" + k], rel=2e-4, msg=k)
            n_checked += 1
        elif p.is_floating_point() and p.grad is not None:
            # parameters the reference leaves without a gradient: beta (unused), beta1 / beta2 (ConditionFunction returns
            # None for them) -- SURVEY section 9's DDP hazard
            assert float(p.grad.abs().max()) == 0.0 or k.endswith(("beta", "beta1", "beta2")), k
    assert n_checked == sum(1 for k in d if k.startswith("g/")) == 98
    sd_eval = {k: v.detach().clone() for k, v in sd.items()}
    for k in d:                                              # forward() rewrote beta1 / beta2 (wdsr_b.py:534)
        if k.startswith("after/"):
            sd_eval[k[len("after/"):]] = d[k]
    oe, se = O.nas_model_forward(d["x"], sd_eval, 4, 0.5, training=False)
    _close(oe, d["out_eval"])
    _close(se, d["speed_eval"], rel=1e-6)
    assert list(d["block_status"].numpy()) == [0, 2, 3] and int(d["current_blocks"]) == 3


# ---- G11 / G12: the reference video models pin the propagation loops and the upsamplers ----
def test_g11_motion_vector_vsr_matches_reference(golden_dir):
    d = _load(golden_dir, "g11_mvvsr.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in _sd(d).items()}
    x = d["x"].clone().requires_grad_(True)
    b, n, _, h, w = x.shape
    out, fb, ff = O.mvvsr_forward(x, sd, 4 * h, 4 * w, num_feat=20, return_feats=True)
    _close(out, d["out"], rel=2e-5)
    # hooks recorded the trunk outputs in CALL order: backward direction runs frames n-1 .. 0
    _close(torch.stack(fb[::-1], 1), d["feat_backward"], rel=2e-5)
    _close(torch.stack(ff, 1), d["feat_forward"], rel=2e-5)
    loss = O.charbonnier(out, d["target"])
    torch.testing.assert_close(loss, d["loss"], rtol=1e-6, atol=1e-7)
    loss.backward()
    _close(x.grad, d["dx"], rel=2e-4)
    for k, p in sd.items():
        if "g/" + k in d:
            _close(p.grad, d["g/" + k], rel=2e-4, msg=k)
        else:                                                # upconv1/2, conv_hr: constructed, never used (mvvsr_arch.py:99-100)
            assert p.grad is None, k


def test_g12_basicvsr_origin_matches_reference(golden_dir):
    d = _load(golden_dir, "g12_basicvsr_origin.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in _sd(d).items()}
    x = d["x"].clone().requires_grad_(True)
    b, n, _, h, w = x.shape
    out = O.basicvsr_origin_forward(x, d["flows_forward"], d["flows_backward"], sd, 4 * h, 4 * w, num_feat=24)
    _close(out, d["out"], rel=2e-5)
    loss = O.charbonnier(out, d["target"])
    torch.testing.assert_close(loss, d["loss"], rtol=1e-6, atol=1e-7)
    loss.backward()
    _close(x.grad, d["dx"], rel=2e-4)
    for k, p in sd.items():
        _close(p.grad, d["g/" + k], rel=2e-4, msg=k)


# ---- G13: Set5-shaped synthetic images: the oracle reproduces the reference's outputs and PSNRs ----
def _g13_net(tag, golden_dir):
    if tag == "x2":
        g3 = _load(golden_dir, "g3_pretrained_x2_8_24.npz")
        ns = argparse.Namespace(image_mean=0.5, num_channels=3, scale=2, num_blocks=8, num_residual_units=24)
        m = O.OracleBasicModel(ns).eval()
        m.load_state_dict(_sd(g3), strict=True)
        return m, 2
    torch.manual_seed(130)
    ns = argparse.Namespace(image_mean=0.5, num_channels=3, scale=4, num_blocks=16, num_residual_units=24)
    return O.OracleBasicModel(ns).eval(), 4


@pytest.mark.parametrize("tag", ["x2", "x4"])
def test_g13_set5_shaped_outputs_and_psnr(golden_dir, tag):
    from oracle.set5_like import SET5_SHAPES, set5_like_hr
    d = _load(golden_dir, "g13_set5_shaped.npz")
    m, r = _g13_net(tag, golden_dir)
    if tag == "x4":                                          # the seeded init IS the reference's (same draws, same order)
        ws = sum(v.double().sum().item() for v in m.state_dict().values())
        assert abs(ws - float(d["x4_weight_sum"])) <= 1e-9 * float(d["x4_weight_abs"])
    for i, hw in enumerate(SET5_SHAPES if tag == "x2" else SET5_SHAPES[1:3]):       # (x4 at 16 blocks: two images keep the CPU suite short)
        if tag == "x4":
            i += 1
        hr = set5_like_hr(i, hw)
        hr = hr[:, :hw[0] - hw[0] % r, :hw[1] - hw[1] % r]
        if tag == "x2":
            assert abs(hr.double().sum().item() - float(d[f"hr_sum_{i}"])) <= 1e-6 * float(d[f"hr_sum_{i}"])
        k = f"{tag}_{i}"
        with torch.no_grad():
            sr = m(d["lr_" + k].float())
        _close(sr[..., ::4, ::4], d["sr_sample_" + k], rel=2e-5)
        assert abs(sr.double().mean().item() - float(d["sr_mean_" + k])) <= 2e-6
        assert abs(O.psnr(sr, hr[None], shave=r + 6).item() - float(d["psnr_" + k])) <= 1e-3
        assert abs(O.psnr_y(sr, hr[None], shave=r).item() - float(d["psnr_y_" + k])) <= 1e-3
