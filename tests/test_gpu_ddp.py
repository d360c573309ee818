"""Data-parallel path of the HIP BASIC_MODEL (pretrain.py:216,239): the two-segment parameter mode gives the
one-parameter gradient, and two DistributedDataParallel ranks (half batch each) give the full-batch gradient with the
late segment's all-reduce issued BEFORE the early half of the backward runs."""
import argparse
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ns(**kw):
    d = dict(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=16, num_residual_units=24,
             hot_dtype="bf16")
    d.update(kw)
    return argparse.Namespace(**d)


@pytest.mark.parametrize("dtype,nb", [("bf16", 16), ("bf16", 5), ("fp32", 4)])
def test_two_segment_backward_equals_one_parameter_backward(dtype, nb):
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(0)
    a = get_model(_ns(hot_dtype=dtype, num_blocks=nb)).cuda().train()
    torch.manual_seed(0)
    b = get_model(_ns(hot_dtype=dtype, num_blocks=nb, hot_grad_segments=2)).cuda().train()
    assert torch.equal(a.flat.detach(), b.flat.detach())
    x = torch.rand(3, 3, 20, 28, device="cuda")
    hr = torch.rand(3, 3, 80, 112, device="cuda")
    ya, yb = a(x), b(x)
    assert torch.equal(ya, yb)
    torch.nn.functional.l1_loss(ya, hr).backward()
    torch.nn.functional.l1_loss(yb, hr).backward()
    gb = torch.cat([b.flat_lo.grad, b.flat_hi.grad])
    if dtype == "bf16":
        assert torch.equal(a.flat.grad, gb)
    else:                                                     # fp32 mode sums db2 through LDS float atomics (arrival order)
        assert (a.flat.grad - gb).abs().max().item() <= 1e-6 * a.flat.grad.abs().max().item()
    # the optimizer route: two tensors, same update
    oa = torch.optim.Adam(a.parameters(), lr=1e-3)
    ob = torch.optim.Adam(b.parameters(), lr=1e-3)
    oa.step(), ob.step()
    if dtype == "bf16":
        assert torch.equal(a.flat.detach(), b.flat.detach())
        with torch.no_grad():
            assert torch.equal(a(x), b(x))                    # both parameters are still views of the one buffer the kernels read


def _run_two_ranks(tmp_path, mode):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / f"ddp_{mode}.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_worker.py"), "--out", out, "--mode", mode]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    return json.load(open(out))


@pytest.mark.parametrize("mode", ["fused", "fused_overlap"])
def test_fused_data_parallel_train_step_two_ranks(tmp_path, mode):
    """model.train_step(..., process_group) on two ranks (half batch each; gradient all-reduced once after the backward, or
    in two halves with the first under the early half of the backward) == one process stepping on the whole batch: replicas
    stay bit-equal, parameters match the full-batch run"""
    res = _run_two_ranks(tmp_path, mode)
    assert res["replicas_equal"], res
    # Adam's first steps move every parameter by ~lr whatever the gradient's size, so a gradient that is zero up to fp32 summation
    # order (the two ranks sum their weight gradients in another grouping) may step the other way: the bulk must agree closely, a
    # stray parameter may differ by a fraction of the three steps' 3 lr
    assert res["param_bad_frac"] <= 1e-3 and res["param_err"] <= 0.1 * 3 * res["lr"], res
    assert all(abs(a) > 0 for a in res["loss_rank0"])


def test_ddp_two_ranks_full_batch_gradient_and_overlap(tmp_path):
    res = _run_two_ranks(tmp_path, "wrapper")
    assert res["rel_err"] <= 2e-5, res                        # per-sample values identical; only the fp32 summation order differs
    ev = res["events"]
    kinds = [e[0] for e in ev]
    assert kinds.count("bucket") == 2 and "lo_backward" in kinds, ev
    first_bucket = kinds.index("bucket")
    assert first_bucket < kinds.index("lo_backward"), f"late segment's all-reduce must start before the early half runs: {ev}"
    sizes = sorted(e[2] for e in ev if e[0] == "bucket")
    assert sizes == sorted(res["n_params"]), (sizes, res["n_params"])


def test_nas_search_phases_rewrap_under_ddp(tmp_path):
    """SURVEY 8 C5 hazard: the reference registers beta / beta1 / beta2 as Parameters that never receive gradients, which
    breaks stock DDP on the second iteration.  NAS_MODEL keeps every block tensor as a slice of one flat parameter (zero
    gradient there), so stock wrapping runs, and the three phases of search.py:290-405 (unwrap, toggle length_grad /
    mask_grad, wrap again) train with the replicas staying bit-equal."""
    res = _run_two_ranks(tmp_path, "nas_phases")
    assert res["stock_ddp_error"] is None, res["stock_ddp_error"]
    for ph in (1, 2, 3):
        assert res[f"phase{ph}_equal"], ph
    gate = lambda ks: [k for k in ks if k.endswith((".alpha1", ".alpha2"))]
    masks = lambda ks: [k for k in ks if k.endswith("split.weight") or k == "mask.weight"]
    convs = lambda ks: [k for k in ks if "weight_v" in k]
    never = lambda ks: [k for k in ks if k.endswith(".beta")]
    # phase 1: gates frozen, masks (trainable since construction) and kernels train
    assert not gate(res["phase1_changed"]) and masks(res["phase1_changed"]) and convs(res["phase1_changed"])
    # phase 2: gates and masks train; nothing is frozen
    assert res["phase2_frozen_kinds"] == []
    assert gate(res["phase2_changed"]) and masks(res["phase2_changed"])
    # phase 3: only kernels (and alpha) train
    assert not gate(res["phase3_changed"]) and not masks(res["phase3_changed"]) and convs(res["phase3_changed"])
    for ph in (1, 2, 3):                                      # `beta` is never used and never moves (zero gradient, fresh Adam)
        assert not never(res[f"phase{ph}_changed"])
