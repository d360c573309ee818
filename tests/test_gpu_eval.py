"""f4: the evaluation path on device -- full-image inference in tiles (bit-identical to the untiled run) and psnr / psnr_y
(common/metrics.py:10-38) against the hand cases G9, the oracle, and the reference's own values on the Set5-shaped
images of fixture G13."""
import argparse
import math
import os

import numpy as np
import pytest
import torch

from oracle import wdsr_oracle as O

pytestmark = pytest.mark.gpu


def _model(dtype, nb=4, scale=4, seed=3):
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(seed)
    ns = argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=scale, num_blocks=nb,
                            num_residual_units=24, hot_dtype=dtype)
    return get_model(ns).cuda().eval()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("shape,tile", [((1, 70, 93), 32), ((2, 48, 200), (48, 64)), ((1, 31, 37), 16), ((1, 40, 40), 64)])
def test_tiled_inference_is_bit_identical_to_untiled(dtype, shape, tile):
    from mobilesuperresolution_amd.inference import tiled_forward, window_plan
    m = _model(dtype)
    n, h, w = shape
    x = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(h * w)).cuda()
    with torch.no_grad():
        whole = m(x)
    tiled = tiled_forward(m, x, tile, max_windows=5)
    assert tiled.shape == whole.shape
    assert torch.equal(tiled, whole)
    nwin = len(window_plan(h, w, tile, m.receptive_halo())[2])
    print(f"\n{dtype} {shape} tile {tile}: {nwin} windows, bit-identical")


def test_device_psnr_hand_cases_g9():
    """the hand-computed cases that pin the oracle (tests/test_oracle_golden.py::test_g9_psnr_hand_cases), on device"""
    from mobilesuperresolution_amd.metrics import psnr, psnr_y
    hr = torch.full((1, 3, 16, 16), 0.5).cuda()
    e = 1.0 / 255.0
    assert float(psnr(hr + e, hr, shave=4)) == pytest.approx(-10 * math.log10((0.5 / 255) ** 2), abs=1e-3)   # 128.5 -> 128
    assert float(psnr(hr + 2 * e, hr, shave=4)) == pytest.approx(-10 * math.log10((2.5 / 255) ** 2), abs=1e-3)  # 129.5 -> 130
    exp_y = -10 * math.log10(((0.257 + 0.504 + 0.098) * e) ** 2)
    assert float(psnr_y(hr + e, hr, shave=4)) == pytest.approx(exp_y, abs=1e-3)          # no quantisation: `r` is unused
    assert float(psnr_y(torch.cat([hr + e] * 3), torch.cat([hr] * 3), shave=2)) == pytest.approx(3 * exp_y, abs=3e-3)
    assert float(psnr_y(torch.full_like(hr, 1.5), torch.ones_like(hr) - e, shave=0)) == pytest.approx(exp_y, abs=1e-3)
    assert math.isnan(float(psnr(hr + e, hr, shave=8)))                                  # everything shaved: mean of nothing


def test_device_psnr_matches_reference_metrics_g9(golden_dir):
    """sr_psnr against values written by the reference's own common/metrics.py (fixture G9, oracle/make_golden.py)"""
    from mobilesuperresolution_amd.metrics import psnr, psnr_y
    z = np.load(os.path.join(golden_dir, "g9_metrics.npz"))
    for k in range(int(z["n_cases"])):
        sr, hr, shave = torch.from_numpy(z[f"sr_{k}"]).cuda(), torch.from_numpy(z[f"hr_{k}"]).cuda(), int(z[f"shave_{k}"])
        assert float(psnr(sr, hr, shave=shave)) == pytest.approx(float(z[f"psnr_{k}"]), abs=1e-3), k
        assert float(psnr_y(sr, hr, shave=shave)) == pytest.approx(float(z[f"psnr_y_{k}"]), abs=1e-3), k


@pytest.mark.parametrize("shape,shave", [((3, 3, 37, 53), 4), ((1, 3, 192, 192), 10), ((2, 1, 20, 24), 2), ((2, 5, 3, 24, 28), 4),
                                         ((1, 3, 9, 9), 0)])
def test_device_psnr_matches_oracle(shape, shave):
    from mobilesuperresolution_amd.metrics import psnr, psnr_y
    g = torch.Generator().manual_seed(sum(shape))
    hr = torch.rand(shape, generator=g)
    sr = hr + 0.05 * torch.randn(shape, generator=g) + 0.3 * (torch.rand(shape, generator=g) > 0.97)    # some values leave [0, 1]
    for fn, ofn in ((psnr, O.psnr), (psnr_y, O.psnr_y)):
        if fn is psnr_y and len(shape) == 5 and shape[1] == 3:
            continue                                  # the reference's conv2d raises on 5-D input there
        got, exp = float(fn(sr.cuda(), hr.cuda(), shave=shave)), float(ofn(sr, hr, shave=shave))
        assert got == pytest.approx(exp, abs=2e-4 * max(1, shape[0])), (fn.__name__, got, exp)


def test_device_psnr_on_set5_shaped_images_matches_reference_values(golden_dir):
    """G13: psnr / psnr_y the REFERENCE computed on its own outputs; here model output, tiling and both metrics stay on
    the GPU (utils/estimate.py:123-124 shaves scale + 6 and scale)"""
    from oracle.set5_like import SET5_SHAPES, set5_like_hr
    from mobilesuperresolution_amd.inference import tiled_forward
    from mobilesuperresolution_amd.metrics import psnr, psnr_y
    from mobilesuperresolution_amd.models import get_model
    z = np.load(os.path.join(golden_dir, "g13_set5_shaped.npz"))
    g3 = np.load(os.path.join(golden_dir, "g3_pretrained_x2_8_24.npz"))
    ns = argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=2, num_blocks=8,
                            num_residual_units=24, hot_dtype="fp32")
    m = get_model(ns)
    m.load_state_dict({k[2:]: torch.from_numpy(g3[k]) for k in g3.files if k.startswith("p/")})
    m = m.cuda().eval()
    for i, hw in enumerate(SET5_SHAPES):
        hr = set5_like_hr(i, hw)
        hr = hr[:, :hw[0] - hw[0] % 2, :hw[1] - hw[1] % 2][None].cuda()
        lr = torch.from_numpy(z[f"lr_x2_{i}"]).float().cuda()
        sr = tiled_forward(m, lr, tile=64)
        dp = abs(float(psnr(sr, hr, shave=2 + 6)) - float(z[f"psnr_x2_{i}"]))
        dpy = abs(float(psnr_y(sr, hr, shave=2)) - float(z[f"psnr_y_x2_{i}"]))
        print(f"G13 image {i}: |d psnr| {dp:.2e} dB, |d psnr_y| {dpy:.2e} dB (tiled, on device)")
        assert dp <= 1e-3 and dpy <= 1e-3
