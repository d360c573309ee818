"""Parity of the fused residual-block HIP kernels (through the C ABI) against the reference's
golden vectors (G2) and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from mobilesuperresolution_amd import packing as P
from oracle import wdsr_oracle as O

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def block_src(d):
    w = [O.weight_norm(d[f"p/body.{i}.weight_v"], d[f"p/body.{i}.weight_g"]).reshape(-1) for i in (0, 2, 3)]
    b = [d[f"p/body.{i}.bias"] for i in (0, 2, 3)]
    return torch.cat(w + b + [torch.tensor([0.0, 1.0])])


def run_block_fwd(x_nchw, src, f, dtype):
    from mobilesuperresolution_amd import _lib as L
    tab = P.block_fwd_tables(f, 6 * f, int(f * 0.84))
    dev = "cuda"
    srcd = src.to(dev)
    w = srcd[torch.from_numpy(tab["w"]).to(dev)].to(dtype).contiguous()
    ci = srcd[torch.from_numpy(tab["cinit"]).to(dev)].float().contiguous()
    x = x_nchw.to(dev).permute(0, 2, 3, 1).contiguous().to(dtype)
    y = torch.full_like(x, float("nan"))
    n, h, wd, _ = x.shape
    L.check(L.lib().sr_wdsr_block_fwd(L.ptr(x), L.ptr(y), L.ptr(w), L.ptr(ci), n, h, wd, f,
                                      L.DTYPE_CODE[dtype], L.stream_ptr()), "sr_wdsr_block_fwd")
    torch.cuda.synchronize()
    return y.float().permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("f", [24, 32])
def test_block_fwd_fp32_matches_reference_golden(golden_dir, f):
    d = _load(golden_dir, f"g2_block_f{f}.npz")
    y = run_block_fwd(d["x"], block_src(d), f, torch.float32)
    err = (y - d["y"]).abs().max().item()
    scale = d["y"].abs().max().item()
    print(f"\nF={f} fp32 max|diff|={err:.3e} scale={scale:.3f}")
    assert err <= 1e-5 * scale            # exact-fp32 MFMA path: fp32 rounding only


@pytest.mark.parametrize("f", [24, 32])
def test_block_fwd_bf16_tolerance(golden_dir, f):
    d = _load(golden_dir, f"g2_block_f{f}.npz")
    y = run_block_fwd(d["x"], block_src(d), f, torch.bfloat16)
    # compare against the oracle fed the same bf16-rounded input; tolerance: bf16 storage (2^-8 rel)
    xr = d["x"].bfloat16().float()
    sd = {k[2:]: v for k, v in d.items() if k.startswith("p/")}
    exp = O.block_forward(xr, sd)
    rel = (y - exp).abs().max().item() / exp.abs().max().item()
    print(f"\nF={f} bf16 rel err={rel:.3e}")
    assert rel < 2e-2


def _assert_same_up_to_summation_order(got, ref, what=""):
    """bf16 outputs of two kernels that form the SAME products in a different fp32 summation order (the dense-K 3x3 of
    csrc/wdsr_fwd_rs.h adds the residual first and walks the taps row by row): equal except where the fp32 sum sits on a bf16
    rounding boundary -- a one-ulp flip on a small fraction of the elements (and whatever a flip in the first block's output does
    to the second block)"""
    g, r = got.float(), ref.float()
    assert torch.isfinite(g).all(), what
    d = (g - r).abs()
    scale = float(r.abs().max())
    assert float(d.max()) <= 2.0 ** -6 * scale, (what, float(d.max()), scale)
    assert float((d > 0).float().mean()) <= 0.03, (what, float((d > 0).float().mean()))
    assert float(d.mean()) <= 1e-4 * scale, (what, float(d.mean()))


def run_block_fwd_rs(x_nchw, src, f):
    """the register-resident forward kernel (one block per launch) on the same golden block"""
    from mobilesuperresolution_amd import _lib as L, hotpath as HP
    blob, cinit = HP.pack_blocks(src.cuda()[None], f, torch.bfloat16)
    x = x_nchw.cuda().permute(0, 2, 3, 1).contiguous().bfloat16()
    y = torch.full_like(x, float("nan"))
    n, h, wd, _ = x.shape
    L.check(L.lib().sr_wdsr_fwd_rs(x.data_ptr(), None, y.data_ptr(), blob[0].data_ptr(), None, cinit[0].data_ptr(), None, None, None,
                                   1, n, h, wd, f, 1, L.stream_ptr()), "sr_wdsr_fwd_rs")
    torch.cuda.synchronize()
    return y.float().permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("f", [24, 32])
def test_block_fwd_rs_bf16_against_reference_golden(golden_dir, f):
    """the graded kernel (wdsr_fwd_rs_kernel: register-resident weights, dense-K 3x3, residual as accumulator init; 32 units:
    wdsr_fwd_rs16_kernel, sixteen waves, weights from LDS at use, 28-channel t rows) directly against G2: same tolerance as the
    round-1 bf16 kernel (bf16 storage, 2^-8 relative)"""
    d = _load(golden_dir, f"g2_block_f{f}.npz")
    y = run_block_fwd_rs(d["x"], block_src(d), f)
    xr = d["x"].bfloat16().float()
    sd = {k[2:]: v for k, v in d.items() if k.startswith("p/")}
    exp = O.block_forward(xr, sd)
    rel = (y - exp).abs().max().item() / exp.abs().max().item()
    l2 = ((y - exp).norm() / exp.norm()).item()
    print(f"\nrs kernel F={f} bf16: max rel err {rel:.3e}, L2 rel {l2:.3e}")
    assert rel < 2e-2 and l2 < 4e-3


def test_block_fwd_48x48_batch_vs_oracle():
    """BASELINE config shape (48x48 patches, F=24), random weights, batch 3."""
    f = 24
    g = torch.Generator().manual_seed(0)
    blk = O.OracleBlock(f, 0.25)
    with torch.no_grad():
        for n_, p in blk.named_parameters():
            if n_.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach() for k, v in blk.named_parameters()}
    d = {"p/" + k: v for k, v in sd.items()}
    x = torch.randn(3, f, 48, 48, generator=g)
    exp = O.block_forward(x, sd)
    y = run_block_fwd(x, block_src(d), f, torch.float32)
    assert not torch.isnan(y).any()
    assert (y - exp).abs().max().item() <= 1e-5 * exp.abs().max().item()


# ------------------------------------------------------------------------------------------
# backward
# ------------------------------------------------------------------------------------------
def _src_with_grad(d):
    ps = {k[2:]: v.clone().requires_grad_(True) for k, v in d.items() if k.startswith("p/")}
    w = [O.weight_norm(ps[f"body.{i}.weight_v"], ps[f"body.{i}.weight_g"]).reshape(-1) for i in (0, 2, 3)]
    b = [ps[f"body.{i}.bias"] for i in (0, 2, 3)]
    return torch.cat(w + b + [torch.tensor([0.0, 1.0])]), ps


def run_block_bwd(d, f, dtype, wgs=3):
    from mobilesuperresolution_amd import hotpath as HP
    src, ps = _src_with_grad(d)
    srcd = src.detach().cuda()[None]
    blob, cinit = HP.pack_blocks(srcd, f, dtype)
    x = d["x"].cuda().permute(0, 2, 3, 1).contiguous().to(dtype)
    dy = d["dy"].cuda().permute(0, 2, 3, 1).contiguous().to(dtype)
    dx = torch.full_like(x, float("nan"))
    HP.block_bwd_data(x, dy, dx, blob[0], cinit[0])
    dsrc = HP.block_wgrad(x[None], dy[None], blob, cinit, wgs_per_layer=wgs)
    torch.cuda.synchronize()
    src.backward(dsrc[0].cpu())
    return dx.float().permute(0, 3, 1, 2).cpu(), {k: p.grad for k, p in ps.items()}


@pytest.mark.parametrize("f", [24, 32])
def test_block_bwd_fp32_matches_reference_golden(golden_dir, f):
    d = _load(golden_dir, f"g2_block_f{f}.npz")
    dx, grads = run_block_bwd(d, f, torch.float32)
    err = (dx - d["dx"]).abs().max().item() / d["dx"].abs().max().item()
    print(f"\nF={f} fp32 dx rel err {err:.2e}")
    assert err <= 1e-5
    for k, g in grads.items():
        exp = d["g/" + k]
        e = (g - exp).abs().max().item() / exp.abs().max().item()
        print(f"  {k}: rel err {e:.2e}")
        assert e <= 1e-4, k


def _effective(d, f, rnd):
    """effective (weight-normalised) weights/biases of the golden block, optionally bf16-rounded"""
    q = (lambda t: t.bfloat16().float()) if rnd else (lambda t: t)
    w = [q(O.weight_norm(d[f"p/body.{i}.weight_v"], d[f"p/body.{i}.weight_g"])) for i in (0, 2, 3)]
    b = [q(d[f"p/body.{i}.bias"]) for i in (0, 2, 3)]
    return [t.clone().requires_grad_(True) for t in w + b]


def _block_eff(x, p):
    import torch.nn.functional as Fn
    h = Fn.relu(Fn.conv2d(x, p[0], p[3]))
    t = Fn.conv2d(h, p[1], p[4])
    return Fn.conv2d(t, p[2], p[5], padding=1) + x


@pytest.mark.parametrize("f", [24, 32])
def test_block_bwd_bf16_tolerance(golden_dir, f):
    """bf16 storage / bf16 MFMA operands, fp32 accumulate.  The oracle gets the same bf16-rounded
    x, dy and effective weights, so the ReLU masks agree and what remains is the rounding of the
    intermediates (h, t, dt, dpre to bf16): tolerance 2% of each tensor's max, relative L2 <= 1%."""
    from mobilesuperresolution_amd import hotpath as HP
    d = _load(golden_dir, f"g2_block_f{f}.npz")
    p = _effective(d, f, True)
    x = d["x"].bfloat16().float().requires_grad_(True)
    dy = d["dy"].bfloat16().float()
    _block_eff(x, p).backward(dy)
    src = torch.cat([t.detach().reshape(-1) for t in p] + [torch.tensor([0.0, 1.0])]).cuda()[None]
    blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
    xd = x.detach().cuda().permute(0, 2, 3, 1).contiguous().bfloat16()
    dyd = dy.cuda().permute(0, 2, 3, 1).contiguous().bfloat16()
    dx = torch.full_like(xd, float("nan"))
    HP.block_bwd_data(xd, dyd, dx, blob[0], cinit[0])
    dsrc = HP.block_wgrad(xd[None], dyd[None], blob, cinit, wgs_per_layer=5)[0].cpu()
    got_dx = dx.float().permute(0, 3, 1, 2).cpu()

    def cmp(name, got, exp):
        mx = (got - exp).abs().max().item() / exp.abs().max().item()
        l2 = ((got - exp).norm() / exp.norm()).item()
        print(f"  {name}: max rel {mx:.2e}  L2 rel {l2:.2e}")
        assert mx <= 2e-2 and l2 <= 1e-2, name

    print(f"\nF={f} bf16 backward")
    cmp("dx", got_dx, x.grad)
    o = 0
    for name, t in zip(("w1", "w2", "w3", "b1", "b2", "b3"), p):
        n = t.numel()
        cmp(name, dsrc[o:o + n], t.grad.reshape(-1))
        o += n


@pytest.mark.parametrize("shape", [(3, 48, 48), (2, 20, 28), (1, 7, 9)])
def test_block_pair_kernel_bit_identical_to_two_launches(shape):
    """sr_wdsr_block2_fwd (two blocks per launch, halo-2 recompute) against two sr_wdsr_block_fwd launches (round-1 kernel):
    the same products, summed in the dense-K order"""
    from mobilesuperresolution_amd import _lib as L, hotpath as HP
    n, h, w = shape
    f = 24
    g = torch.Generator().manual_seed(21)
    src = (torch.randn(2, HP.tables(f, torch.device("cuda", 0))["src_size"], generator=g) * 0.08).cuda()
    src[:, -2], src[:, -1] = 0.0, 1.0
    blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
    x = torch.randn(n, h, w, f, generator=g).cuda().bfloat16()
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    HP.block_fwd(x, y1, blob[0], cinit[0])
    HP.block_fwd(y1, y2, blob[1], cinit[1])
    p1, p2 = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
    L.check(L.lib().sr_wdsr_block2_fwd(x.data_ptr(), p1.data_ptr(), p2.data_ptr(), blob[0].data_ptr(), blob[1].data_ptr(),
                                       cinit[0].data_ptr(), cinit[1].data_ptr(), None, None, n, h, w, f, 1, L.stream_ptr()), "pair")
    torch.cuda.synchronize()
    _assert_same_up_to_summation_order(p1, y1, "block 0")
    _assert_same_up_to_summation_order(p2, y2, "block 1")


@pytest.mark.parametrize("f,nblk", [(24, 1), (24, 2), (32, 1), (32, 2)])
@pytest.mark.parametrize("shape", [(3, 48, 48), (2, 20, 28), (1, 7, 9), (2, 37, 91)])
def test_role_specialised_forward_matches_round1_kernels_and_itself(shape, f, nblk):
    """sr_wdsr_fwd_rs (register-resident weights, LDS-DMA staging, dense-K 3x3) against sr_wdsr_block_fwd launches (the same
    products in another summation order), and -- bit for bit -- the two-block launch against two one-block launches: block
    outputs and the saved t images, ragged tiles and images smaller than one tile included"""
    from mobilesuperresolution_amd import _lib as L, hotpath as HP
    n, h, w = shape
    g = torch.Generator().manual_seed(23)
    src = (torch.randn(2, HP.tables(f, torch.device("cuda", 0))["src_size"], generator=g) * 0.08).cuda()
    src[:, -2], src[:, -1] = 0.0, 1.0
    blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
    x = torch.randn(n, h, w, f, generator=g).cuda().bfloat16()
    tiles = ((h + 11) // 12) * ((w + 23) // 24)
    lp = 24 if f == 24 else 32
    lib = L.lib()
    # reference: the single-block kernel through the whole-net "save" path's building block (keeps t as well)
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    HP.block_fwd(x, y1, blob[0], cinit[0])
    HP.block_fwd(y1, y2, blob[1], cinit[1])
    p1, p2 = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
    ts = torch.full((2, n, tiles, 288, lp), float("nan"), device="cuda", dtype=torch.bfloat16)
    if nblk == 2:
        L.check(lib.sr_wdsr_fwd_rs(x.data_ptr(), p1.data_ptr(), p2.data_ptr(), blob[0].data_ptr(), blob[1].data_ptr(),
                                   cinit[0].data_ptr(), cinit[1].data_ptr(), ts[0].data_ptr(), ts[1].data_ptr(), 2, n, h, w, f, 1,
                                   L.stream_ptr()), "rs2")
    else:
        L.check(lib.sr_wdsr_fwd_rs(x.data_ptr(), None, p1.data_ptr(), blob[0].data_ptr(), None, cinit[0].data_ptr(), None,
                                   ts[0].data_ptr(), None, 1, n, h, w, f, 1, L.stream_ptr()), "rs1a")
        L.check(lib.sr_wdsr_fwd_rs(p1.data_ptr(), None, p2.data_ptr(), blob[1].data_ptr(), None, cinit[1].data_ptr(), None,
                                   ts[1].data_ptr(), None, 1, n, h, w, f, 1, L.stream_ptr()), "rs1b")
    torch.cuda.synchronize()
    _assert_same_up_to_summation_order(p1, y1, "block 0")
    _assert_same_up_to_summation_order(p2, y2, "block 1")
    # saved t images: one-block and two-block launches must keep the same t (the weight-gradient tests check t itself
    # against the recompute kernels)
    if nblk == 2:
        ts1 = torch.full_like(ts, float("nan"))
        q1, q2 = torch.empty_like(x), torch.empty_like(x)
        L.check(lib.sr_wdsr_fwd_rs(x.data_ptr(), None, q1.data_ptr(), blob[0].data_ptr(), None, cinit[0].data_ptr(), None,
                                   ts1[0].data_ptr(), None, 1, n, h, w, f, 1, L.stream_ptr()), "rs1a")
        L.check(lib.sr_wdsr_fwd_rs(q1.data_ptr(), None, q2.data_ptr(), blob[1].data_ptr(), None, cinit[1].data_ptr(), None,
                                   ts1[1].data_ptr(), None, 1, n, h, w, f, 1, L.stream_ptr()), "rs1b")
        torch.cuda.synchronize()
        assert torch.equal(p1, q1) and torch.equal(p2, q2)
        assert torch.equal(torch.nan_to_num(ts, nan=-7.0), torch.nan_to_num(ts1, nan=-7.0))


@pytest.mark.parametrize("f", [24, 32])
@pytest.mark.parametrize("shape", [(3, 48, 48), (2, 20, 28), (1, 7, 9)])
def test_block_pair_bwd_data_matches_two_launches_and_torch(shape, f):
    """sr_wdsr_block2_bwd_data (round 3: csrc/wdsr_bwd_rs.h, register-resident weights, the skip term as the accumulator's initial
    value) against two sr_wdsr_block_bwd_data launches of the round-1 kernel (G2-checked; same products, another summation
    order), incl. ragged tiles; the saved dt images against torch's conv_transpose2d of the bf16 gradients"""
    from mobilesuperresolution_amd import _lib as L, hotpath as HP
    n, h, w = shape
    lp, ll = (24, 20) if f == 24 else (32, 26)
    g = torch.Generator().manual_seed(22)
    src = (torch.randn(2, HP.tables(f, torch.device("cuda", 0))["src_size"], generator=g) * 0.08).cuda()
    src[:, -2], src[:, -1] = 0.0, 1.0
    blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
    xa = torch.randn(n, h, w, f, generator=g).cuda().bfloat16()
    xb = torch.randn(n, h, w, f, generator=g).cuda().bfloat16()
    dyb = torch.randn(n, h, w, f, generator=g).cuda().bfloat16()
    d1, d0 = torch.empty_like(xa), torch.empty_like(xa)
    HP.block_bwd_data(xb, dyb, d1, blob[1], cinit[1])
    HP.block_bwd_data(xa, d1, d0, blob[0], cinit[0])
    p1, p0 = torch.full_like(xa, float("nan")), torch.full_like(xa, float("nan"))
    tiles = ((h + 11) // 12) * ((w + 23) // 24)
    dts = torch.full((2, n, tiles, 288, lp), float("nan"), device="cuda", dtype=torch.bfloat16)
    L.check(L.lib().sr_wdsr_block2_bwd_data(xa.data_ptr(), xb.data_ptr(), dyb.data_ptr(), p1.data_ptr(), p0.data_ptr(),
                                            blob[0].data_ptr(), blob[1].data_ptr(), cinit[0].data_ptr(), cinit[1].data_ptr(),
                                            dts[0].data_ptr(), dts[1].data_ptr(), n, h, w, f, 1, L.stream_ptr()), "pair bwd")
    q1, q0 = torch.full_like(xa, float("nan")), torch.full_like(xa, float("nan"))
    L.check(L.lib().sr_wdsr_block2_bwd_data(xa.data_ptr(), xb.data_ptr(), dyb.data_ptr(), q1.data_ptr(), q0.data_ptr(),
                                            blob[0].data_ptr(), blob[1].data_ptr(), cinit[0].data_ptr(), cinit[1].data_ptr(),
                                            None, None, n, h, w, f, 1, L.stream_ptr()), "pair bwd, no dt")
    torch.cuda.synchronize()
    assert torch.equal(p1, q1) and torch.equal(p0, q0)                 # (with and without the saved images)
    if n > 1:                                                          # an image's result does not depend on its batch
        s1, s0 = torch.full_like(xa[-1:], float("nan")), torch.full_like(xa[-1:], float("nan"))
        L.check(L.lib().sr_wdsr_block2_bwd_data(xa[-1:].data_ptr(), xb[-1:].data_ptr(), dyb[-1:].data_ptr(), s1.data_ptr(), s0.data_ptr(),
                                                blob[0].data_ptr(), blob[1].data_ptr(), cinit[0].data_ptr(), cinit[1].data_ptr(),
                                                None, None, 1, h, w, f, 1, L.stream_ptr()), "pair bwd, last image alone")
        torch.cuda.synchronize()
        assert torch.equal(s1, p1[-1:]) and torch.equal(s0, p0[-1:])
    if f == 32:            # csrc/wdsr_bwd_pair_lds.h: the single-block kernel's own chain per pixel tile, dx_b rounded in between as the tensor is
        assert torch.equal(p1, d1) and torch.equal(p0, d0)
    else:
        _assert_same_up_to_summation_order(p1, d1, "dx of block b")
        _assert_same_up_to_summation_order(p0, d0, "dx of block a")
    # the saved dt images: dt = conv_transpose(dy, W3) on bf16 operands, tile-local [tile 12 x 24][288][24] with zeros in channels 20..23
    # and outside the image
    import torch.nn.functional as Fn
    off = P.BlockGeom(f, 6 * f, ll).off
    for blk, dy in ((1, dyb), (0, p1)):
        w3 = src[blk, off["w3"]:off["w3"] + f * ll * 9].view(f, ll, 3, 3).bfloat16().float()
        ref = Fn.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w3, padding=1)          # (n, L, h, w)
        got = dts[blk].float()
        th, tw = (h + 11) // 12, (w + 23) // 24
        full = torch.zeros(n, th * 12, tw * 24, lp, device="cuda")
        full[:, :h, :w, :ll] = ref.permute(0, 2, 3, 1)
        exp = full.view(n, th, 12, tw, 24, lp).permute(0, 1, 3, 2, 4, 5).reshape(n, tiles, 288, lp)
        assert torch.isfinite(got).all()
        err = float((got - exp).abs().max()) / float(exp.abs().max())
        assert err <= 1e-2, (blk, err)


@pytest.mark.parametrize("f,r,shape", [(24, 4, (3, 48, 48)), (24, 2, (2, 20, 28)), (32, 3, (1, 7, 9)), (32, 4, (2, 13, 50))])
def test_fused_tail_backward_bit_identical_to_two_launches(f, r, shape):
    """sr_tail_bwd (data + weight gradients of the tail, one launch, bf16) == sr_tail_bwd_data + sr_tail_wgrad"""
    from mobilesuperresolution_amd import _lib as L, hotpath as HP
    n, h, w = shape
    dev = torch.device("cuda", 0)
    tb = HP.ends_tables(f, r, dev)
    g = torch.Generator().manual_seed(31)
    src_tail = (torch.randn(tb["tail_size"], generator=g) * 0.05).cuda()
    src_tail[-2], src_tail[-1] = 0.0, 1.0
    blob = HP.pack_tail(src_tail, f, r, torch.bfloat16)
    feat = torch.randn(n, h, w, f, generator=g).cuda().bfloat16()
    x = torch.rand(n, 3, h, w, generator=g).cuda()
    dout = torch.randn(n, 3, r * h, r * w, generator=g).cuda()
    wgs = 16
    d_ref = torch.empty_like(feat)
    p_ref = torch.zeros(wgs, tb["tail_slab"], device=dev)
    HP.tail_bwd_data(dout, d_ref, blob, r)
    L.check(L.lib().sr_tail_wgrad(dout.data_ptr(), feat.data_ptr(), x.data_ptr(), 0.5, p_ref.data_ptr(), wgs, n, h, w, f, r, 1,
                                  L.stream_ptr()), "wgrad")
    d_new = torch.full_like(feat, float("nan"))
    p_new = torch.zeros(wgs, tb["tail_slab"], device=dev)
    L.check(L.lib().sr_tail_bwd(dout.data_ptr(), feat.data_ptr(), x.data_ptr(), 0.5, blob.data_ptr(), d_new.data_ptr(),
                                p_new.data_ptr(), wgs, n, h, w, f, r, 1, L.stream_ptr()), "fused")
    torch.cuda.synchronize()
    # slab columns outside the gather table come from LDS bytes past the 4-channel image rows (never used): compare
    # what the network consumes, slab by slab
    used = tb["tail_grad"]
    assert torch.equal(d_new, d_ref) and torch.equal(p_new.index_select(1, used), p_ref.index_select(1, used))


@pytest.mark.parametrize("shape", [(150, 48, 48), (80, 50, 70)])
def test_persistent_forward_for_many_tiles_bit_identical(shape):
    """>= 1024 tiles per launch: sr_wdsr_fwd_rs takes its persistent form (one workgroup per CU walking the tiles, weights
    staged once, the next tile's x landing under the current tile) -- same bits as the per-tile launches of the same kernel
    family on a small batch, for one block with and without saved t images and for two blocks per launch; tile counts that do
    not divide by the workgroup count; against the round-1 single-block kernels up to the summation order"""
    from mobilesuperresolution_amd import _lib as L, hotpath as HP
    n, h, w = shape
    tiles = ((h + 11) // 12) * ((w + 23) // 24)
    assert n * tiles >= 1024 and (n * tiles) % 256 != 0
    f = 24
    g = torch.Generator().manual_seed(29)
    src = (torch.randn(2, HP.tables(f, torch.device("cuda", 0))["src_size"], generator=g) * 0.08).cuda()
    src[:, -2], src[:, -1] = 0.0, 1.0
    blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
    x = torch.randn(n, h, w, f, generator=g).cuda().bfloat16()
    lib = L.lib()
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    HP.block_fwd(x, y1, blob[0], cinit[0])
    HP.block_fwd(y1, y2, blob[1], cinit[1])
    # two blocks per launch, no saved images (inference)
    p1, p2 = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
    L.check(lib.sr_wdsr_fwd_rs(x.data_ptr(), p1.data_ptr(), p2.data_ptr(), blob[0].data_ptr(), blob[1].data_ptr(), cinit[0].data_ptr(),
                               cinit[1].data_ptr(), None, None, 2, n, h, w, f, 1, L.stream_ptr()), "rs2 persistent")
    m = 5                                                 # 5 images: far below the persistent threshold
    s1, s2 = torch.empty_like(x[:m]), torch.empty_like(x[:m])
    L.check(lib.sr_wdsr_fwd_rs(x[:m].contiguous().data_ptr(), s1.data_ptr(), s2.data_ptr(), blob[0].data_ptr(), blob[1].data_ptr(),
                               cinit[0].data_ptr(), cinit[1].data_ptr(), None, None, 2, m, h, w, f, 1, L.stream_ptr()), "rs2 per tile")
    torch.cuda.synchronize()
    assert torch.equal(p1[:m], s1) and torch.equal(p2[:m], s2)
    _assert_same_up_to_summation_order(p1, y1, "block 0")
    _assert_same_up_to_summation_order(p2, y2, "block 1")
    # one block per launch with the saved t image, against the per-tile launches of a smaller batch (same per-tile layout)
    ts = torch.full((n, tiles, 288, 24), float("nan"), device="cuda", dtype=torch.bfloat16)
    q1 = torch.full_like(x, float("nan"))
    L.check(lib.sr_wdsr_fwd_rs(x.data_ptr(), None, q1.data_ptr(), blob[0].data_ptr(), None, cinit[0].data_ptr(), None, ts.data_ptr(), None,
                               1, n, h, w, f, 1, L.stream_ptr()), "rs1 persistent")
    ts_small = torch.full((m, tiles, 288, 24), float("nan"), device="cuda", dtype=torch.bfloat16)
    q_small = torch.empty_like(x[:m])
    L.check(lib.sr_wdsr_fwd_rs(x[:m].contiguous().data_ptr(), None, q_small.data_ptr(), blob[0].data_ptr(), None, cinit[0].data_ptr(), None,
                               ts_small.data_ptr(), None, 1, m, h, w, f, 1, L.stream_ptr()), "rs1 per tile")
    torch.cuda.synchronize()
    assert torch.equal(q1[:m], q_small)
    _assert_same_up_to_summation_order(q1, y1, "one block")
    assert torch.equal(torch.nan_to_num(ts[:m], nan=-7.0), torch.nan_to_num(ts_small, nan=-7.0))


@pytest.mark.parametrize("shape", [(256, 48, 48), (384, 20, 48), (512, 8, 48)])
def test_streaming_forward_for_whole_images_bit_identical(shape):
    """>= 256 images of width 48 (H % 4 == 0): sr_wdsr_fwd_rs with two blocks takes the streaming kernel (csrc/wdsr_fwd_stream.h:
    one image per workgroup, bands of four rows, row rings in LDS, no halo recompute).  Same per-pixel arithmetic as the tile
    kernels: block outputs and both saved t images equal the per-tile launches of sub-batches bit for bit, with and without
    the saved images, with and without block 0's output"""
    from mobilesuperresolution_amd import _lib as L, hotpath as HP
    n, h, w = shape
    f = 24
    g = torch.Generator().manual_seed(41)
    src = (torch.randn(2, HP.tables(f, torch.device("cuda", 0))["src_size"], generator=g) * 0.08).cuda()
    src[:, -2], src[:, -1] = 0.0, 1.0
    blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
    x = torch.randn(n, h, w, f, generator=g).cuda().bfloat16()
    tiles = ((h + 11) // 12) * ((w + 23) // 24)
    lib = L.lib()

    def run(xs, save, want_a):
        m = xs.shape[0]
        a = torch.full_like(xs, float("nan")) if want_a else None
        b = torch.full_like(xs, float("nan"))
        ts = torch.full((2, m, tiles, 288, 24), float("nan"), device="cuda", dtype=torch.bfloat16) if save else None
        L.check(lib.sr_wdsr_fwd_rs(xs.data_ptr(), a.data_ptr() if want_a else None, b.data_ptr(), blob[0].data_ptr(), blob[1].data_ptr(),
                                   cinit[0].data_ptr(), cinit[1].data_ptr(), ts[0].data_ptr() if save else None,
                                   ts[1].data_ptr() if save else None, 2, m, h, w, f, 1, L.stream_ptr()), "rs2")
        torch.cuda.synchronize()
        return a, b, ts

    m = 7                                                 # sub-batch: per-tile launches
    ra, rb, rts = run(x[:m].contiguous(), True, True)
    ra2, rb2, rts2 = run(x[n - m:].contiguous(), True, True)
    for save, want_a in ((True, True), (False, True), (False, False)):
        a, b, ts = run(x, save, want_a)
        assert torch.isfinite(b.float()).all()
        assert torch.equal(b[:m], rb) and torch.equal(b[n - m:], rb2)
        if want_a:
            assert torch.equal(a[:m], ra) and torch.equal(a[n - m:], ra2)
        if save:
            nn = lambda t: torch.nan_to_num(t.float(), nan=-7.0)
            assert torch.equal(nn(ts[:, :m]), nn(rts)) and torch.equal(nn(ts[:, n - m:]), nn(rts2))
