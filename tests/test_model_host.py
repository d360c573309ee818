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
