"""Parity of the BasicVSR propagation trunk (ConvResidualBlocks on csrc/conv3x3.h) against the
reference's golden vector G7 and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import wdsr_oracle as O

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _trunk(sd, dtype, nin=27, nb=8):
    from mobilesuperresolution_amd.models import ConvResidualBlocks
    m = ConvResidualBlocks(nin, 24, nb, hot_dtype=dtype)
    m.load_state_dict(sd, strict=True)
    return m.cuda()


def test_g7_trunk_fp32_matches_reference(golden_dir):
    d = _load(golden_dir, "g7_vsr_trunk.npz")
    sd = {k[2:]: v for k, v in d.items() if k.startswith("p/")}
    m = _trunk(sd, "fp32")
    assert set(m.state_dict().keys()) == set(sd.keys())
    x = d["x"].cuda().requires_grad_(True)
    y = m(x)
    err = (y.detach().cpu() - d["y"]).abs().max().item() / d["y"].abs().max().item()
    print(f"\nG7 fwd rel err {err:.2e}")
    assert err <= 1e-5
    y.backward(d["dy"].cuda())
    e = (x.grad.cpu() - d["dx"]).abs().max().item() / d["dx"].abs().max().item()
    print(f"G7 dx rel err {e:.2e}")
    assert e <= 1e-5
    worst = 0.0
    assert [k for k, _ in m.named_parameters()] == ["flat"]          # one flat parameter, reference keys as views
    for k, gview in m.named_tensors(m.flat.grad):
        exp = d["g/" + k]
        ge = (gview.cpu() - exp).abs().max().item() / exp.abs().max().item()
        worst = max(worst, ge)
        assert ge <= 1e-4, (k, ge)
    print(f"G7 worst param-grad rel err {worst:.2e}")


def test_trunk_bf16_tolerance(golden_dir):
    d = _load(golden_dir, "g7_vsr_trunk.npz")
    sd = {k[2:]: v for k, v in d.items() if k.startswith("p/")}
    m = _trunk(sd, "bf16")
    with torch.no_grad():
        y = m(d["x"].cuda()).cpu()
    rel = ((y - d["y"]).norm() / d["y"].norm()).item()
    print(f"\nbf16 trunk L2 rel err {rel:.2e}")
    assert rel <= 2e-2           # 17 chained bf16-stored layers, fp32 accumulation


def test_recurrent_propagation_matches_oracle():
    """C4 shape (64x64, 5 frames, F=24, 8 blocks) with given flows: the two propagation loops around the
    HIP trunk vs the oracle trunk (flow_warp = the oracle's restatement of the vendored one on both sides)."""
    from mobilesuperresolution_amd.models import ConvResidualBlocks
    from mobilesuperresolution_amd.models.basicvsr_arch import propagate
    torch.manual_seed(0)
    fb, ff = ConvResidualBlocks(27, 24, 8, "fp32"), ConvResidualBlocks(27, 24, 8, "fp32")
    sdb, sdf = fb.state_dict(), ff.state_dict()
    g = torch.Generator().manual_seed(4)
    x = torch.rand(1, 5, 3, 64, 64, generator=g)
    fl = torch.rand(1, 4, 2, 64, 64, generator=g) * 4 - 2
    ob, of = propagate(x.cuda(), fl.cuda(), -fl.cuda(), fb.cuda(), ff.cuda(), O.flow_warp)
    # reference side: the ORACLE's own restatement of the loops (oracle/wdsr_oracle.py:_propagate, pinned against the
    # reference MotionVectorVSR / BasicVSR_origin by G11 / G12), not the product helper
    sd = {f"backward_trunk.{k}": v for k, v in sdb.items()}
    sd.update({f"forward_trunk.{k}": v for k, v in sdf.items()})
    rb, rf = O._propagate(x, fl, -fl, sd, 24)
    for a, b in zip(ob + of, rb + rf):
        assert (a.cpu() - b).abs().max().item() <= 2e-5 * b.abs().max().item()


def test_g8_flow_warp_matches_reference(golden_dir):
    from mobilesuperresolution_amd.models.spynet_arch import flow_warp
    d = _load(golden_dir, "g8_flow_warp.npz")
    x = d["x"].cuda().requires_grad_(True)
    fl = d["flow"].cuda().requires_grad_(True)
    y = flow_warp(x, fl)
    assert (y.detach().cpu() - d["y"]).abs().max().item() <= 1e-5 * d["y"].abs().max().item()
    y.backward(d["dy"].cuda())
    assert (x.grad.cpu() - d["dx"]).abs().max().item() <= 1e-5 * d["dx"].abs().max().item()
    assert (fl.grad.cpu() - d["dflow"]).abs().max().item() <= 2e-4 * d["dflow"].abs().max().item()
    # integer flows are pure shifts with zero fill: bit-exact
    xs = torch.rand(1, 3, 9, 11, device="cuda")
    f = torch.zeros(1, 9, 11, 2, device="cuda")
    f[..., 0], f[..., 1] = 2.0, -1.0
    out = flow_warp(xs, f)
    exp = torch.zeros_like(xs)
    exp[:, :, 1:, :9] = xs[:, :, :8, 2:]
    assert torch.allclose(out, exp, atol=1e-6)


def test_recurrent_propagation_all_hip():
    """propagate() with the HIP trunk AND the HIP flow_warp, vs the oracle on both."""
    from mobilesuperresolution_amd.models import ConvResidualBlocks
    from mobilesuperresolution_amd.models.basicvsr_arch import propagate
    from mobilesuperresolution_amd.models.spynet_arch import flow_warp
    torch.manual_seed(1)
    fb, ff = ConvResidualBlocks(27, 24, 2, "fp32"), ConvResidualBlocks(27, 24, 2, "fp32")
    sdb, sdf = fb.state_dict(), ff.state_dict()
    g = torch.Generator().manual_seed(6)
    x = torch.rand(2, 3, 3, 40, 36, generator=g)
    fl = torch.rand(2, 2, 2, 40, 36, generator=g) * 4 - 2
    ob, of = propagate(x.cuda(), fl.cuda(), -fl.cuda(), fb.cuda(), ff.cuda(), flow_warp)
    sd = {f"backward_trunk.{k}": v for k, v in sdb.items()}
    sd.update({f"forward_trunk.{k}": v for k, v in sdf.items()})
    rb, rf = O._propagate(x, fl, -fl, sd, 24)                # the oracle's loops (pinned by G11 / G12), not the product's
    for a, b in zip(ob + of, rb + rf):
        assert (a.cpu() - b).abs().max().item() <= 5e-5 * b.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 20, 28), (1, 7, 9)])
def test_fused_residual_block_launches_bit_identical_to_per_layer_path(shape):
    """bf16: the trunk's one-launch-per-residual-block kernels (forward and backward-data) and the batched weight
    gradient launches give, bit for bit, what the per-layer entry points (sr_c3_fwd / _bwd_data / _wgrad) give."""
    import torch.nn as nn
    from mobilesuperresolution_amd import _lib as L
    from mobilesuperresolution_amd.models import ConvResidualBlocks
    from mobilesuperresolution_amd.models.basicvsr_arch import _pack, _tables
    n, h, w = shape
    nb, dt, code = 3, torch.bfloat16, 1
    torch.manual_seed(11)
    m = ConvResidualBlocks(27, 24, nb, "bf16").cuda()
    x = torch.randn(n, 27, h, w, device="cuda").requires_grad_(True)
    dy = torch.randn(n, 24, h, w, device="cuda")
    y = m(x)
    y.backward(dy)

    # per-layer restatement through the single-conv entry points
    sd = m.state_dict()
    convs = []
    for key in ["main.0"] + [f"main.2.{i}.conv{j}" for i in range(nb) for j in (1, 2)]:
        c = nn.Conv2d(sd[key + ".weight"].shape[1], 24, 3, 1, 1).cuda()
        with torch.no_grad():
            c.weight.copy_(sd[key + ".weight"]); c.bias.copy_(sd[key + ".bias"])
        convs.append(c)
    blobs = [_pack(c, dt) for c in convs]
    st = L.stream_ptr
    lib = L.lib()
    x0 = torch.zeros(n, h, w, 32, dtype=dt, device="cuda")
    x0[..., :27] = x.detach().permute(0, 2, 3, 1)
    acts = torch.empty(nb + 1, n, h, w, 24, dtype=dt, device="cuda")
    mids = torch.empty(nb, n, h, w, 24, dtype=dt, device="cuda")
    L.check(lib.sr_c3_fwd(x0.data_ptr(), None, acts[0].data_ptr(), blobs[0].data_ptr(), n, h, w, 32, 2, code, st()), "f0")
    for i in range(nb):
        L.check(lib.sr_c3_fwd(acts[i].data_ptr(), None, mids[i].data_ptr(), blobs[1 + 2 * i].data_ptr(), n, h, w, 24, 1, code, st()), "f1")
        L.check(lib.sr_c3_fwd(mids[i].data_ptr(), acts[i].data_ptr(), acts[i + 1].data_ptr(), blobs[2 + 2 * i].data_ptr(), n, h, w,
                              24, 0, code, st()), "f2")
    assert torch.equal(y, acts[nb].permute(0, 3, 1, 2).float())
    wgs = 64
    g = dy.permute(0, 2, 3, 1).to(dt).contiguous()
    parts = torch.empty(len(convs), wgs, 9 * 1024, dtype=torch.float32, device="cuda")
    dtmp = torch.empty_like(g)
    for i in range(nb - 1, -1, -1):
        L.check(lib.sr_c3_wgrad(mids[i].data_ptr(), g.data_ptr(), None, parts[2 + 2 * i].data_ptr(), wgs, n, h, w, 24, 0, code, st()), "w2")
        L.check(lib.sr_c3_bwd_data(g.data_ptr(), None, None, dtmp.data_ptr(), blobs[2 + 2 * i].data_ptr(), n, h, w, 24, 0, code, st()), "b2")
        L.check(lib.sr_c3_wgrad(acts[i].data_ptr(), dtmp.data_ptr(), mids[i].data_ptr(), parts[1 + 2 * i].data_ptr(), wgs, n, h, w,
                                24, 1, code, st()), "w1")
        gn = torch.empty_like(g)
        L.check(lib.sr_c3_bwd_data(dtmp.data_ptr(), mids[i].data_ptr(), g.data_ptr(), gn.data_ptr(), blobs[1 + 2 * i].data_ptr(),
                                   n, h, w, 24, 1, code, st()), "b1")
        g = gn
    L.check(lib.sr_c3_wgrad(x0.data_ptr(), g.data_ptr(), acts[0].data_ptr(), parts[0].data_ptr(), wgs, n, h, w, 32, 2, code, st()), "w0")
    dx0 = torch.empty_like(x0)
    L.check(lib.sr_c3_bwd_data(g.data_ptr(), acts[0].data_ptr(), None, dx0.data_ptr(), blobs[0].data_ptr(), n, h, w, 32, 2, code, st()), "b0")
    torch.cuda.synchronize()
    assert torch.equal(x.grad, dx0[..., :27].permute(0, 3, 1, 2).float())
    slabs = parts.sum(1)
    ref = []
    for k, c in enumerate(convs):
        gidx = _tables(c.weight.shape[1], 0)[1]
        ref.append(slabs[k].index_select(0, gidx))
    ref = torch.cat(ref)                               # the in-kernel slab reduction adds the 64 slabs in another order
    assert (m.flat.grad - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


def _unfused_step(trunk, frame, prev_feat, flow):
    """the reference's three statements (models/basicvsr_arch.py:74-76) through the separate kernels"""
    from mobilesuperresolution_amd.models import flow_warp
    if flow is not None:
        prev_feat = flow_warp(prev_feat, flow.permute(0, 2, 3, 1))
    return trunk(torch.cat([frame, prev_feat], dim=1))


@pytest.mark.parametrize("dtype,nf,amp", [("fp32", 24, 2.0), ("fp32", 20, 2.0), ("bf16", 24, 2.0), ("fp32", 24, 9.0)])
def test_warp_concat_first_conv_prologue_equals_separate_kernels(dtype, nf, amp):
    """f1: ConvResidualBlocks.forward_warped (warp + concat gathered into the first conv's LDS tile, flow_warp backward in
    gather form without atomics) against flow_warp -> torch.cat -> trunk on the same inputs: features, d flat, d frame,
    d flow, and the state gradient that reaches the previous frame's trunk.  amp 9: flows far beyond the usual window and
    well outside the image."""
    from mobilesuperresolution_amd.models import ConvResidualBlocks
    torch.manual_seed(5)
    n, h, w, nb = 2, 30, 41, 2
    mk = lambda: ConvResidualBlocks(nf + 3, nf, nb, hot_dtype=dtype).cuda()
    a = mk()
    b = mk()
    b.load_state_dict(a.state_dict())
    g = torch.Generator().manual_seed(11)
    f0, f1 = (torch.rand(n, 3, h, w, generator=g).cuda() for _ in range(2))
    flow = ((torch.rand(n, 2, h, w, generator=g) * 2 - 1) * amp).cuda()
    wy, ws = torch.randn(n, nf, h, w, generator=g).cuda(), torch.randn(n, nf, h, w, generator=g).cuda()

    def run(fused):
        m = a if fused else b
        fr1 = f1.clone().requires_grad_(True)
        fl = flow.clone().requires_grad_(True)
        if fused:
            y0, st = m.forward_warped(f0)
            y1, _ = m.forward_warped(fr1, st, fl)
        else:
            y0 = _unfused_step(m, f0, torch.zeros(n, nf, h, w, device="cuda"), None)
            y1 = _unfused_step(m, fr1, y0, fl)
        ((y1 * wy).sum() + (y0 * ws).sum()).backward()
        return y0.detach(), y1.detach(), m.flat.grad.clone(), fr1.grad.clone(), fl.grad.clone()
    got, ref = run(True), run(False)
    tol = 2e-5 if dtype == "fp32" else 3e-2
    for name, x, y in zip(("y0", "y1", "dflat", "dframe", "dflow"), got, ref):
        err = ((x - y).norm() / y.norm().clamp_min(1e-12)).item()
        print(f"{dtype} F={nf} amp={amp} {name}: rel L2 {err:.2e}")
        assert err <= tol, (name, err)
    if dtype == "bf16":                                          # the gathered input is the very tensor the separate kernels build
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])


def test_propagate_fused_equals_unfused_loops():
    """propagate() with the hot trunks takes the fused route; handing it a wrapped flow_warp forces the reference-shaped
    loop over the separate kernels: same features, same parameter gradients"""
    from mobilesuperresolution_amd.models import ConvResidualBlocks, flow_warp
    from mobilesuperresolution_amd.models.basicvsr_arch import propagate
    torch.manual_seed(2)
    bt, ft = (ConvResidualBlocks(27, 24, 3, hot_dtype="fp32").cuda() for _ in range(2))
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 4, 3, 32, 40, generator=g).cuda()
    ff, fb = ((torch.rand(2, 3, 2, 32, 40, generator=g) * 4 - 2).cuda() for _ in range(2))

    def run(fw):
        for m in (bt, ft):
            m.flat.grad = None
        ob, of = propagate(x, ff, fb, bt, ft, fw)
        sum((o * (i + 1)).sum() for i, o in enumerate(ob + of)).backward()
        return torch.stack(ob + of).detach(), bt.flat.grad.clone(), ft.flat.grad.clone()
    fused = run(flow_warp)
    plain = run(lambda a, b: flow_warp(a, b))
    for name, p, q in zip(("features", "d backward_trunk", "d forward_trunk"), fused, plain):
        err = ((p - q).norm() / q.norm()).item()
        print(f"propagate fused vs unfused {name}: rel L2 {err:.2e}")
        assert err <= 2e-5


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_both_directions_in_one_launch_equal_separate_directions(dtype, monkeypatch):
    """propagate() runs the backward-time and the forward-time loop's frame step k in ONE set of launches (round 3: the trunk kernels
    pick the weights by batch half, the weight-gradient slabs are split by half): features per frame and direction and every
    gradient equal the two separately run loops (SR_VSR_SEPARATE_DIRECTIONS=1) -- bit for bit in bf16, to fp32 summation
    order of the weight gradients otherwise (another number of slabs per conv)"""
    from mobilesuperresolution_amd.models import ConvResidualBlocks, flow_warp
    from mobilesuperresolution_amd.models.basicvsr_arch import propagate
    torch.manual_seed(21)
    bt = ConvResidualBlocks(27, 24, 3, hot_dtype=dtype).cuda()
    ft = ConvResidualBlocks(27, 24, 3, hot_dtype=dtype).cuda()
    b, n, h, w = 3, 4, 30, 52
    clip = torch.rand(b, n, 3, h, w, device="cuda")
    ff = (torch.rand(b, n - 1, 2, h, w, device="cuda") * 6 - 3).requires_grad_(True)
    fb = (torch.rand(b, n - 1, 2, h, w, device="cuda") * 6 - 3).requires_grad_(True)
    wts = [torch.randn(b, 24, h, w, device="cuda") for _ in range(2 * n)]

    def run():
        for p in (bt.flat, ft.flat, ff, fb):
            p.grad = None
        ob, of = propagate(clip, ff, fb, bt, ft, flow_warp)
        sum((o * wt).sum() for o, wt in zip(ob + of, wts)).backward()
        return [o.detach().clone() for o in ob + of], [t.grad.clone() for t in (bt.flat, ft.flat, ff, fb)]

    monkeypatch.delenv("SR_VSR_SEPARATE_DIRECTIONS", raising=False)
    feats, grads = run()
    monkeypatch.setenv("SR_VSR_SEPARATE_DIRECTIONS", "1")
    feats_s, grads_s = run()
    for a, c in zip(feats, feats_s):
        assert torch.equal(a, c)
    for a, c, name in zip(grads, grads_s, ("backward trunk", "forward trunk", "flows_forward", "flows_backward")):
        if name.startswith("flows") and dtype == "bf16":
            assert torch.equal(a, c), name
        else:
            assert float((a - c).abs().max()) <= 2e-5 * max(float(c.abs().max()), 1e-6), name
