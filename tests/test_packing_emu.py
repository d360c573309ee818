"""CPU checks of the host packing tables + the wave-level algorithm (numpy MFMA
emulation, tests/mfma_emu.py) against the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

from mobilesuperresolution_amd import packing as P
from oracle import wdsr_oracle as O
from tests import mfma_emu as M


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def block_src(d, geom):
    w = [O.weight_norm(d[f"p/body.{i}.weight_v"], d[f"p/body.{i}.weight_g"]).reshape(-1) for i in (0, 2, 3)]
    b = [d[f"p/body.{i}.bias"] for i in (0, 2, 3)]
    src = torch.cat(w + b + [torch.tensor([0.0, 1.0])]).double().numpy()
    assert src.size == geom.off["size"]
    return src


@pytest.mark.parametrize("f", [24, 32])
def test_block_fwd_tables_match_golden(golden_dir, f):
    d = _load(golden_dir, f"g2_block_f{f}.npz")
    tab = P.block_fwd_tables(f, 6 * f, int(f * 0.84))
    g = tab["geom"]
    src = block_src(d, g)
    pw, ci = src[tab["w"]], src[tab["cinit"]]
    assert pw.size == (tab["n_w1"] + tab["n_w2"] + tab["n_w3"]) * 512
    x = d["x"][0].permute(1, 2, 0).double().numpy()          # (H, W, F) = 20 x 28: partial tiles
    y = M.emu_block_fwd(x, pw, ci, g, dtype="f32")
    exp = d["y"][0].permute(1, 2, 0).double().numpy()
    err = np.abs(y - exp).max()
    assert err <= 3e-6 * np.abs(exp).max(), err


def test_block_fwd_bf16_emulation_tolerance(golden_dir):
    d = _load(golden_dir, "g2_block_f24.npz")
    tab = P.block_fwd_tables(24, 144, 20)
    g = tab["geom"]
    src = block_src(d, g)
    x = d["x"][0].permute(1, 2, 0).double().numpy()[:12, :24]
    exp = O.block_forward(torch.from_numpy(x).float().permute(2, 0, 1)[None], {k[2:]: v for k, v in d.items() if k.startswith("p/")})
    y = M.emu_block_fwd(x, src[tab["w"]], src[tab["cinit"]], g, dtype="bf16")
    exp = exp[0].permute(1, 2, 0).double().numpy()
    rel = np.abs(y - exp).max() / np.abs(exp).max()
    assert rel < 2e-2, rel        # bf16 storage/operands, fp32 accumulate


def test_geometry_constants():
    g = P.BlockGeom(24, 144, 20)
    assert (g.KX, g.KS1, g.NET, g.KS2, g.LP, g.CPT, g.KS3) == (32, 2, 5, 9, 24, 3, 15)
    g = P.BlockGeom(32, 192, 26)
    assert (g.KX, g.KS1, g.NET, g.KS2, g.LP, g.CPT, g.KS3) == (32, 2, 6, 12, 32, 4, 20)


def _oracle_src_with_grad(d):
    """canonical src vector as a differentiable function of (weight_g, weight_v, bias)"""
    ps = {k[2:]: v.clone().requires_grad_(True) for k, v in d.items() if k.startswith("p/")}
    w = [O.weight_norm(ps[f"body.{i}.weight_v"], ps[f"body.{i}.weight_g"]).reshape(-1) for i in (0, 2, 3)]
    b = [ps[f"body.{i}.bias"] for i in (0, 2, 3)]
    return torch.cat(w + b + [torch.tensor([0.0, 1.0])]), ps


@pytest.mark.parametrize("f", [24, 32])
def test_block_bwd_tables_match_golden(golden_dir, f):
    d = _load(golden_dir, f"g2_block_f{f}.npz")
    tab = P.block_tables(f, 6 * f, int(f * 0.84))
    gt = P.block_grad_tables(f, 6 * f, int(f * 0.84))
    g = tab["geom"]
    src_t, ps = _oracle_src_with_grad(d)
    src = src_t.detach().double().numpy()
    pw, ci = src[tab["w"]], src[tab["cinit"]]
    slab_a, slab_b = None, None
    for n in range(d["x"].shape[0]):
        x = d["x"][n].permute(1, 2, 0).double().numpy()
        dy = d["dy"][n].permute(1, 2, 0).double().numpy()
        dx = M.emu_block_bwd_data(x, dy, pw, ci, tab)
        exp = d["dx"][n].permute(1, 2, 0).double().numpy()
        assert np.abs(dx - exp).max() <= 1e-5 * np.abs(exp).max()
        slab_a = M.emu_block_wgrad12(x, dy, pw, ci, tab, slab=slab_a)
        slab_b = M.emu_block_wgrad3(x, dy, pw, ci, tab, slab=slab_b)
    assert slab_a.size == gt["a_size"] and slab_b.size == gt["b_size"]
    ga, gb = slab_a[gt["a"]], slab_b[gt["b"]]
    o = g.off
    dsrc = np.zeros(o["size"])
    E, L = g.E, g.L
    dsrc[o["w1"]:o["w1"] + E * f] = ga[:E * f]
    dsrc[o["w2"]:o["w2"] + L * E] = ga[E * f:E * f + L * E]
    dsrc[o["b1"]:o["b1"] + E] = ga[E * f + L * E:E * f + L * E + E]
    dsrc[o["b2"]:o["b2"] + L] = ga[E * f + L * E + E:]
    dsrc[o["w3"]:o["w3"] + f * L * 9] = gb[:f * L * 9]
    dsrc[o["b3"]:o["b3"] + f] = gb[f * L * 9:]
    src_t.backward(torch.from_numpy(dsrc).float())
    for k, p in ps.items():
        exp = d["g/" + k]
        err = (p.grad - exp).abs().max().item()
        assert err <= 1e-4 * exp.abs().max().item(), (k, err)


@pytest.mark.parametrize("units", [24, 32])
@pytest.mark.parametrize("dtype,tol", [("f32", 3e-6), ("bf16", 6e-3)])
def test_dense_k_conv3_tables_match_golden(golden_dir, dtype, tol, units):
    """W3D (packing.block_tables) + the dense-K addressing of csrc/wdsr_fwd_rs.h (RwBAddrD: 16 contiguous bytes per lane half
    and k-step, the ones chunk in its slot of the last window row, the residual as accumulator init) against the reference block's
    own t -> y (G2: y = conv3x3(t) + b3 + x); 24 units: 20-channel t rows, 12 k-steps; 32 units: 26 channels in 28-channel rows, 18"""
    d = _load(golden_dir, f"g2_block_f{units}.npz")
    E, Lc = units * 6, int(units * 6 * 0.84 // 6) if False else {24: 20, 32: 26}[units]
    tab = P.block_tables(units, E, Lc)
    g = tab["geom"]
    src = block_src(d, g)
    sec = tab["sec"]
    ks3d = {24: 12, 32: 18}[units]
    assert tab["KS3D"] == ks3d and tab["nfrag"] == sec["W3D"] + ks3d
    w3d = src[tab["w"]][sec["W3D"] * 512:(sec["W3D"] + ks3d) * 512]
    t = d["t2"][0].permute(1, 2, 0).double().numpy() if "t2" in d else None
    x = d["x"][0].permute(1, 2, 0).double().numpy()
    if t is None:                                                    # the fixture names its intermediates differently: recompute t
        import torch.nn.functional as Fn
        sd = {k[2:]: v for k, v in d.items() if k.startswith("p/")}
        w1 = O.weight_norm(sd["body.0.weight_v"], sd["body.0.weight_g"])
        w2 = O.weight_norm(sd["body.2.weight_v"], sd["body.2.weight_g"])
        tt = Fn.conv2d(Fn.relu(Fn.conv2d(d["x"][:1], w1, sd["body.0.bias"])), w2, sd["body.2.bias"])
        t = tt[0].permute(1, 2, 0).double().numpy()
    y = M.emu_conv3_dense(t, x, w3d, Lc, units, dtype=dtype)
    import torch.nn.functional as Fn
    sd = {k[2:]: v for k, v in d.items() if k.startswith("p/")}
    w3 = O.weight_norm(sd["body.3.weight_v"], sd["body.3.weight_g"]).double()
    exp = Fn.conv2d(torch.from_numpy(t).permute(2, 0, 1)[None], w3, sd["body.3.bias"].double(), padding=1)[0].permute(1, 2, 0).numpy() + x
    err = np.abs(y - exp).max() / np.abs(exp).max()
    assert err <= tol, err
