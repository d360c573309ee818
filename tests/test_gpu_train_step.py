"""Loss / optimizer epilogue on the hot path: the loss folded into the tail backward kernel, the fused Adam
kernel and the one-call training step, against the unfused route (torch L1 / Charbonnier + torch.optim.Adam), which is
what pretrain.py:73-80,137 and train_video_superresolution.py:43-53 run."""
import argparse
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ns(dtype, nb=3, f=24, scale=4):
    return argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=scale, num_blocks=nb,
                              num_residual_units=f, hot_dtype=dtype)


def _charbonnier(sr, hr):
    return torch.sqrt((sr - hr) ** 2 + 1e-12).mean()          # train_video_superresolution.py:43-53


def test_adam_kernel_bit_identical_to_torch_adam():
    from mobilesuperresolution_amd import _lib as L
    from mobilesuperresolution_amd.models.basic_wdsr_b import AdamState
    g = torch.Generator().manual_seed(5)
    n = 191_368 + 3                                            # C2's parameter count, plus a ragged tail
    p0 = torch.randn(n, generator=g).cuda()
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p_ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)     # pretrain.py:137
    p = p0.clone()
    st = AdamState(p, 1e-3, (0.9, 0.999), 1e-8)
    for step in range(6):
        grad = (torch.randn(n, generator=g) * (10.0 ** (step - 3))).cuda()
        p_ref.grad = grad.clone()
        opt.step()
        sc = st.next_scalars()
        L.check(L.lib().sr_adam_step(p.data_ptr(), grad.data_ptr(), st.exp_avg.data_ptr(), st.exp_avg_sq.data_ptr(), n,
                                     ctypes.byref(sc), None, 0, 0.0, None, L.stream_ptr()), "adam")
        torch.cuda.synchronize()
        s = opt.state[p_ref]
        assert torch.equal(st.exp_avg, s["exp_avg"]), f"exp_avg differs at step {step}"
        assert torch.equal(st.exp_avg_sq, s["exp_avg_sq"]), f"exp_avg_sq differs at step {step}"
        assert torch.equal(p, p_ref.detach()), f"parameters differ at step {step}"


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("kind", ["l1", "charbonnier"])
def test_folded_loss_matches_unfused_route(dtype, kind):
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(0)
    m = get_model(_ns(dtype)).cuda().train()
    x = torch.rand(3, 3, 20, 28, device="cuda")
    hr = torch.rand(3, 3, 80, 112, device="cuda")
    hr[0, :, :4, :4] = m(x).detach()[0, :, :4, :4]            # exact zeros of sr - hr: sign(0) = 0 must hold
    m.flat.grad = None
    sr = m(x)
    ref_loss = torch.nn.functional.l1_loss(sr, hr) if kind == "l1" else _charbonnier(sr, hr)
    (0.7 * ref_loss).backward()
    g_ref = m.flat.grad.clone()
    m.flat.grad = None
    loss = m.loss(x, hr, kind=kind, weight=0.7)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - 0.7 * ref_loss.item()) <= 2e-6 * abs(ref_loss.item()) + 1e-7
    if kind == "l1" and dtype == "bf16":
        assert torch.equal(m.flat.grad, g_ref)               # same gradient values into the same kernels
    elif kind == "l1":
        # fp32 parity mode sums db2 through LDS float atomics (arrival order): the conv2 bias gradients move by an ulp
        # from run to run, everything else is bit-identical
        d = (m.flat.grad - g_ref).abs()
        assert d.max().item() <= 1e-6 * g_ref.abs().max().item() and (d > 0).sum().item() <= 4 * m.layout.NB * 20
    else:
        rel = ((m.flat.grad - g_ref).norm() / g_ref.norm()).item()
        assert rel <= (2e-6 if dtype == "fp32" else 2e-3), rel   # torch forms d / sqrt(d^2 + eps) in a different order


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_train_step_equals_unfused_steps(dtype):
    """model.train_step == zero_grad / forward / F.l1_loss / backward / Adam.step, parameter for parameter"""
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(1)
    a = get_model(_ns(dtype, nb=4)).cuda().train()
    b = get_model(_ns(dtype, nb=4)).cuda().train()
    b.load_state_dict(a.state_dict(), strict=True)
    opt = torch.optim.Adam(a.parameters(), lr=1e-3)
    st = b.make_train_state(lr=1e-3)
    g = torch.Generator().manual_seed(2)
    for step in range(4):
        x = torch.rand(4, 3, 48, 48, generator=g).cuda()
        hr = torch.rand(4, 3, 192, 192, generator=g).cuda()
        opt.zero_grad()
        la = torch.nn.functional.l1_loss(a(x), hr)
        la.backward()
        opt.step()
        lb = b.train_step(x, hr, st)
        torch.cuda.synchronize()
        assert abs(la.item() - lb.item()) <= 2e-6 * abs(la.item()), (step, la.item(), lb.item())
        if dtype == "bf16":
            assert torch.equal(a.flat.detach(), b.flat.detach()), f"parameters differ after step {step}"
        else:                                                 # (db2 through LDS float atomics in fp32 mode, see above)
            assert (a.flat.detach() - b.flat.detach()).abs().max().item() <= 2e-5, step
    # the updated weights are the ones the next forward uses (packed blobs are re-made)
    with torch.no_grad():
        if dtype == "bf16":
            assert torch.equal(a(x), b(x))
        else:
            assert (a(x) - b(x)).abs().max().item() <= 2e-3      # Adam turns an ulp of db2 into a full-size first step


def test_backward_after_interleaved_forward_repacks():
    """two forwards with DIFFERENT models' weights share nothing; two forwards of one model before one backward: the
    saved packed-blob key differs only if the weights changed, which autograd's version check refuses"""
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(3)
    m = get_model(_ns("fp32", nb=2)).cuda().train()
    x = torch.rand(1, 3, 12, 24, device="cuda")
    y = m(x)
    with torch.no_grad():
        m.flat.add_(0.01)                                      # "optimizer step" between forward and backward
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        y.sum().backward()


# ---- the reference trainers' own loop on the fast route: training.L1Loss / L1_Charbonnier_loss + training.Adam ----
def _reference_loop(model, opt, crit, x, hr, steps, weight=1.0):
    """pretrain.py:69-82, verbatim in structure"""
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        sr = model(x)
        loss = 0
        loss_sr_l1 = weight * crit(sr, hr)
        loss += loss_sr_l1
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(loss.item())
    return losses


@pytest.mark.parametrize("kind", ["l1", "charbonnier"])
def test_reference_training_loop_with_drop_in_loss_and_adam_is_bit_identical(kind):
    """the reference's loop with the two swapped imports (mobilesuperresolution_amd.training) against the same loop with
    torch's nn.L1Loss / the reference's Charbonnier module and torch.optim.Adam: bf16 mode, same kernels underneath, so the
    parameters agree bit for bit after every step; the loss values to fp32 rounding"""
    from mobilesuperresolution_amd import training as T
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(3)
    a = get_model(_ns("bf16", nb=4)).cuda().train()
    b = get_model(_ns("bf16", nb=4)).cuda().train()
    b.load_state_dict(a.state_dict(), strict=True)
    x = torch.rand(4, 3, 24, 24, device="cuda")
    hr = torch.rand(4, 3, 96, 96, device="cuda")
    crit_t = torch.nn.L1Loss() if kind == "l1" else (lambda s, h: _charbonnier(s, h))
    crit_h = T.L1Loss() if kind == "l1" else T.L1_Charbonnier_loss()
    opt_t = torch.optim.Adam(filter(lambda p: p.requires_grad, a.parameters()), 1e-3)
    opt_h = T.Adam(filter(lambda p: p.requires_grad, b.parameters()), 1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt_h, milestones=[2], gamma=0.3)       # pretrain.py:139-142
    sched_t = torch.optim.lr_scheduler.MultiStepLR(opt_t, milestones=[2], gamma=0.3)
    for epoch in range(4):
        lt = _reference_loop(a, opt_t, crit_t, x, hr, 2, weight=0.5)
        lh = _reference_loop(b, opt_h, crit_h, x, hr, 2, weight=0.5)
        sched.step()
        sched_t.step()
        for u, v in zip(lt, lh):
            assert abs(u - v) <= 2e-6 * abs(u) + 1e-7
        if kind == "l1":
            assert torch.equal(a.flat.detach(), b.flat.detach()), f"parameters differ after epoch {epoch}"
        else:
            assert float((a.flat.detach() - b.flat.detach()).abs().max()) <= 2e-5
    sd = opt_h.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 8.0
    opt_h2 = T.Adam(b.parameters(), 1e-3)
    opt_h2.load_state_dict(sd)                                  # checkpoint round trip (pretrain.py:262-267)
    assert torch.equal(opt_h2.state[b.flat]["exp_avg"], opt_h.state[b.flat]["exp_avg"])


def test_drop_in_loss_other_uses_of_the_output_still_work():
    """a second, ordinary use of `sr` next to the folded criterion (gradient = folded + usual backward), a criterion on a
    detached / foreign tensor (torch's own ops), and eval mode"""
    from mobilesuperresolution_amd import training as T
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(4)
    m = get_model(_ns("bf16", nb=2)).cuda().train()
    x = torch.rand(2, 3, 20, 28, device="cuda")
    hr = torch.rand(2, 3, 80, 112, device="cuda")
    crit = T.L1Loss()
    sr = m(x)
    (crit(sr, hr) + 0.1 * sr.mean()).backward()
    g = m.flat.grad.clone()
    m.flat.grad = None
    sr = m(x)
    (torch.nn.functional.l1_loss(sr, hr) + 0.1 * sr.mean()).backward()
    ref = m.flat.grad
    assert float((g - ref).abs().max()) <= 5e-3 * float(ref.abs().max())        # (bf16: the two gradients are rounded separately)
    assert float((g - ref).norm()) <= 5e-3 * float(ref.norm())
    assert float(crit(sr.detach(), hr)) == pytest.approx(float(torch.nn.functional.l1_loss(sr.detach(), hr)), rel=1e-6)
    with torch.no_grad():
        m.eval()
        assert float(crit(m(x), hr)) > 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_scale_by_device_scalar_forms_the_product_in_fp32(dtype):
    """sr_scale_by (what a folded loss's data gradient is multiplied with: the trainer's loss weight as autograd's device scalar):
    y = round_to_dtype(fp32(x) * scale), ragged length, 16-byte aligned buffers"""
    from mobilesuperresolution_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    for n in (8, 2048, 2048 * 3 + 5 * 8, 100003 * 8):
        x = torch.randn(n, generator=g).cuda().to(dtype)
        s = torch.tensor([0.3712345], device="cuda")
        y = torch.full_like(x, float("nan"))
        L.check(L.lib().sr_scale_by(y.data_ptr(), x.data_ptr(), n, s.data_ptr(), L.DTYPE_CODE[dtype], L.stream_ptr()), "sr_scale_by")
        torch.cuda.synchronize()
        assert torch.equal(y, (x.float() * s).to(dtype))
    assert L.lib().sr_scale_by(None, x.data_ptr(), 8, s.data_ptr(), 1, L.stream_ptr()) == -2
