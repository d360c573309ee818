"""Host-side behaviour of the model wrappers that needs no GPU: pickling / deep copies (the reference trainers pickle
whole modules, train_video_superresolution.py:306), device checks, the argparse surface."""
import argparse
import copy
import pickle

import pytest
import torch


def _ns(**kw):
    d = dict(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=2, num_residual_units=24,
             hot_dtype="bf16")
    d.update(kw)
    return argparse.Namespace(**d)


def test_deepcopy_and_pickle_drop_device_state():
    from mobilesuperresolution_amd.models import get_model
    m = get_model(_ns())
    m._dev[("cuda", 0)] = object()                           # stands for a populated _DeviceState (ctypes pointers)
    m2 = copy.deepcopy(m)
    assert m2._dev == {} and m2.flat is not m.flat and torch.equal(m2.flat, m.flat)
    m3 = pickle.loads(pickle.dumps(m))
    assert m3._dev == {} and torch.equal(m3.flat, m.flat)
    assert list(m3.state_dict().keys()) == list(m.state_dict().keys())


def test_cpu_tensors_and_device_mismatch_raise():
    from mobilesuperresolution_amd import _lib as L
    from mobilesuperresolution_amd.models import get_model
    m = get_model(_ns())
    with pytest.raises(L.HotpathError):
        m(torch.rand(1, 3, 8, 8))                            # CPU tensor: no fallback

    class FakeCuda:                                          # device bookkeeping only: no GPU in this container
        def __init__(self, idx):
            self.is_cuda, self.device = True, torch.device("cuda", idx)
    x, flat = FakeCuda(0), FakeCuda(1)
    # exercise the check directly: x on cuda:0, parameters on cuda:1
    with pytest.raises(L.HotpathError, match="different devices"):
        type(m)._check_input(type("M", (), {"flat": flat})(), x)


def test_window_plan_covers_every_pixel_once_with_halo():
    """inference.window_plan: cores partition the image; every window edge that is not the image border is at least
    `halo` away from the core it serves"""
    from mobilesuperresolution_amd.inference import window_plan
    for (h, w, tile, halo) in [(70, 93, 32, 6), (48, 200, (48, 64), 18), (31, 37, 16, 6), (40, 40, 64, 6), (512, 340, 96, 18),
                               (37, 37, 1, 3)]:
        wh, ww, plan = window_plan(h, w, tile, halo)
        cover = [[0] * w for _ in range(h)]
        for (oy, ox, y0, y1, x0, x1) in plan:
            assert 0 <= oy and oy + wh <= h and 0 <= ox and ox + ww <= w
            assert oy <= y0 < y1 <= oy + wh and ox <= x0 < x1 <= ox + ww
            assert oy == 0 or y0 - oy >= halo
            assert oy + wh == h or oy + wh - y1 >= halo
            assert ox == 0 or x0 - ox >= halo
            assert ox + ww == w or ox + ww - x1 >= halo
            for y in range(y0, y1):
                for x in range(x0, x1):
                    cover[y][x] += 1
        assert all(v == 1 for row in cover for v in row)


def test_metrics_refuse_cpu_tensors():
    import pytest
    import torch
    from mobilesuperresolution_amd import _lib as L
    from mobilesuperresolution_amd.metrics import psnr, psnr_y
    a = torch.rand(1, 3, 16, 16)
    for fn in (psnr, psnr_y):
        with pytest.raises(L.HotpathError):
            fn(a, a)


def test_patch_oracle_shapes_and_flip_order():
    """oracle/patch_oracle.py restates datasets/_isr.py:87-121: crop, then row flip, column flip, axis swap, then to_tensor"""
    import random
    import numpy as np
    from oracle import patch_oracle as PO

    class Fixed:
        def __init__(self, ints, floats):
            self.ints, self.floats = list(ints), list(floats)

        def randrange(self, a, b):
            v = self.ints.pop(0)
            assert a <= v < b
            return v

        def random(self):
            return self.floats.pop(0)
    lr = np.arange(10 * 12 * 3, dtype=np.uint8).reshape(10, 12, 3)
    hr = np.arange(20 * 24 * 3, dtype=np.uint8).reshape(20, 24, 3)
    l, h = PO.train_item([lr], [hr], 0, 4, 2, 1, 1, Fixed([3, 5], [0.1, 0.9, 0.2]))       # row flip, no column flip, swap
    assert l.shape == (3, 4, 4) and h.shape == (3, 8, 8) and l.dtype == np.float32
    crop = lr[3:7, 5:9][::-1]
    assert np.array_equal(l, np.swapaxes(crop, 0, 1).transpose(2, 0, 1).astype(np.float32) / np.float32(255))
    assert h[0, 0, 0] == np.float32(hr[13, 10, 0]) / np.float32(255)                        # HR crop rows 6..13 flipped -> first is row 13


def test_device_mismatch_is_a_python_error_not_a_gpu_fault():
    """tensors on different devices must raise before any kernel is enqueued (checked with meta / cpu stand-ins:
    the comparison happens before any device call)"""
    import pytest
    import torch
    from mobilesuperresolution_amd import _lib as L
    from mobilesuperresolution_amd.models import ConvResidualBlocks, flow_warp
    t = ConvResidualBlocks(27, 24, 1)
    with pytest.raises(L.HotpathError):
        t(torch.zeros(1, 27, 8, 8))                           # CPU tensor: refused, no fallback
    with pytest.raises(L.HotpathError):
        flow_warp(torch.zeros(1, 4, 8, 8), torch.zeros(1, 8, 8, 2))

    class FakeCuda(torch.Tensor):                             # claims to live on cuda:1 while the parameters are elsewhere
        @property
        def is_cuda(self):
            return True

        @property
        def device(self):
            return torch.device("cuda", 1)
    x = torch.zeros(1, 27, 8, 8).as_subclass(FakeCuda)
    with pytest.raises(L.HotpathError, match="parameters on"):
        t(x)
    with pytest.raises(L.HotpathError, match="parameters on"):
        t.forward_warped(torch.zeros(1, 3, 8, 8).as_subclass(FakeCuda))
