"""f3: training patches cut on device (csrc/patches.h) against items of the reference's own dataset class (fixture G14, written by
oracle/make_golden.py from datasets/_isr.py:56-121) and against the CPU restatement, draw for draw."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import patch_oracle as PO

pytestmark = pytest.mark.gpu


def _images(scale, n=5, seed=0):
    g = np.random.default_rng(seed)
    lrs, hrs = [], []
    for k in range(n):
        h, w = int(g.integers(40, 90)), int(g.integers(40, 90))
        lrs.append(g.integers(0, 256, (h, w, 3), dtype=np.uint8))
        hrs.append(g.integers(0, 256, (h * scale + (k % 2), w * scale + (k % 3), 3), dtype=np.uint8))   # HR may be a few pixels larger
    return lrs, hrs


@pytest.mark.parametrize("scale,P,ignored,num_patches", [(4, 24, 0, 3), (2, 32, 4, 1), (3, 17, 2, 2)])
def test_device_patches_equal_reference_items_draw_for_draw(scale, P, ignored, num_patches):
    from mobilesuperresolution_amd.datasets import DevicePatchCache
    lrs, hrs = _images(scale)
    ds = DevicePatchCache(lrs, hrs, P, scale, ignored, num_patches)
    assert len(ds) == len(lrs) * num_patches
    idx = list(range(len(ds))) * 12                      # 60-180 items: all eight flip / transpose combinations occur
    lr, hr = ds.batch(idx, random.Random(1234))
    rng = random.Random(1234)
    seen = set()
    for b, i in enumerate(idx):
        state = rng.getstate()
        el, eh = PO.train_item(lrs, hrs, i, P, scale, ignored, num_patches, rng)
        probe = random.Random()
        probe.setstate(state)
        seen.add(ds.draw(i, probe)[6])
        assert torch.equal(lr[b].cpu(), torch.from_numpy(el)), (b, i)
        assert torch.equal(hr[b].cpu(), torch.from_numpy(eh)), (b, i)
    assert seen == set(range(8))


def test_device_patches_equal_reference_dataset_items_g14(golden_dir):
    """G14: every item the reference's ImageSuperResolutionDataset.__getitem__ produced under a seeded `random` comes out of
    DevicePatchCache.batch bit for bit, and the RNG has made the same number of draws afterwards"""
    from mobilesuperresolution_amd.datasets import DevicePatchCache
    z = np.load(os.path.join(golden_dir, "g14_patches.npz"))
    for ci, (scale, P, ignored, num_patches) in enumerate(z["cfgs"].tolist()):
        n_img = int(z[f"c{ci}_n_img"])
        lrs = [z[f"c{ci}_lr{k}"] for k in range(n_img)]
        hrs = [z[f"c{ci}_hr{k}"] for k in range(n_img)]
        ds = DevicePatchCache(lrs, hrs, P, scale, ignored, num_patches)
        rng = random.Random(int(z[f"c{ci}_seed"]))
        lr, hr = ds.batch(z[f"c{ci}_idx"].tolist(), rng)
        exp_lr = torch.from_numpy(z[f"c{ci}_lr_items"]).float().div(255)
        exp_hr = torch.from_numpy(z[f"c{ci}_hr_items"]).float().div(255)
        assert torch.equal(lr.cpu(), exp_lr), ci
        assert torch.equal(hr.cpu(), exp_hr), ci
        assert rng.random() == float(z[f"c{ci}_next_random"])


def test_device_cache_consumes_the_rng_like_the_reference():
    from mobilesuperresolution_amd.datasets import DevicePatchCache
    lrs, hrs = _images(4, n=3, seed=2)
    ds = DevicePatchCache(lrs, hrs, 24, 4)
    a, b = random.Random(9), random.Random(9)
    ds.batch([0, 1, 2, 1], a)
    for i in (0, 1, 2, 1):
        PO.train_item(lrs, hrs, i, 24, 4, 0, 1, b)
    assert a.getstate() == b.getstate()


def test_feeds_the_training_step():
    """batch -> BASIC_MODEL.train_step without leaving the device"""
    import argparse
    from mobilesuperresolution_amd.datasets import DevicePatchCache
    from mobilesuperresolution_amd.models import get_model
    lrs, hrs = _images(4, n=4, seed=5)
    ds = DevicePatchCache(lrs, hrs, 24, 4)
    torch.manual_seed(0)
    ns = argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=2, num_residual_units=24,
                            hot_dtype="bf16")
    m = get_model(ns).cuda().train()
    st = m.make_train_state(1e-3)
    lr, hr = ds.batch(range(len(ds)), random.Random(0))
    l0 = m.train_step(lr, hr, st).item()
    for _ in range(20):
        l1 = m.train_step(lr, hr, st).item()
    assert l1 < l0
