"""CPU checks of the NAS path's host logic that replaced the reference's host-side `if`s with tensor ops
(no GPU sync in the training path): rounding, the batched latency terms and the batched hard gate."""
import torch

from oracle import wdsr_oracle as O
from mobilesuperresolution_amd.models.ops import rounding
from mobilesuperresolution_amd.models.wdsr_b import ConditionFunction, _GateFunction


def test_sync_free_rounding_equals_reference_semantics():
    g = torch.Generator().manual_seed(0)
    for trial in range(200):
        c = [8, 12, 24, 32][trial % 4]
        w = torch.rand(c, 1, 1, 1, generator=g) * (0.6 if trial % 3 else 1.0)       # often fewer than 8 above 0.5
        if trial % 7 == 0:
            w[: c // 2] = w[0]                                                      # ties at the k-th largest value
        for least in (8, 0, 3):
            if least > c:
                continue
            assert torch.equal(rounding(w, least), O.rounding(w, least)), (trial, least)


def test_batched_rows_rounding_equals_per_row():
    """the (NB, F) form used by NAS_MODEL._body for the latency terms"""
    g = torch.Generator().manual_seed(1)
    W = torch.rand(16, 32, generator=g) * 0.7
    W[3, :12] = W[3, 0]                                                             # ties around the 8th value
    hard = (W >= 0.5).float()
    top8 = ((W.unsqueeze(1) > W.unsqueeze(2)).sum(2) < 8).float()                   # topk-free: fewer than 8 strictly larger
    batched = torch.where(hard.sum(1, keepdim=True) >= 8, hard, top8)
    for i in range(W.shape[0]):
        assert torch.equal(batched[i], O.rounding(W[i].view(-1, 1, 1, 1), 8).view(-1))


def test_batched_gate_matches_condition_function():
    g = torch.Generator().manual_seed(2)
    a1 = torch.rand(16, generator=g).requires_grad_(True)
    a2 = torch.rand(16, generator=g).requires_grad_(True)
    a2.data[3] = a1.data[3]                                                         # equality -> (1, 0), as `>=` in the reference
    gates = _GateFunction.apply(a1, a2)
    up = torch.randn(16, 2, generator=g)
    (gates * up).sum().backward()
    for i in range(16):
        x1 = a1.detach()[i:i + 1].clone().requires_grad_(True)
        x2 = a2.detach()[i:i + 1].clone().requires_grad_(True)
        b1, b2 = ConditionFunction.apply(x1, x2, torch.zeros(1), torch.ones(1))
        assert float(b1) == float(gates[i, 0]) and float(b2) == float(gates[i, 1]) and float(b1 + b2) == 1.0
        (b1 * up[i, 0] + b2 * up[i, 1]).sum().backward()
        assert torch.equal(x1.grad, a1.grad[i:i + 1]) and torch.equal(x2.grad, a2.grad[i:i + 1])   # straight-through


def test_load_pretrained_is_positional_like_the_reference(tmp_path):
    """reference wdsr_b.py:235-250: walk parameters(), take the checkpoint's next tensor when shapes agree.  A BASIC_MODEL
    checkpoint therefore fills the head (bias, weight_g, weight_v) and then stalls on body.0's 144-wide bias."""
    import argparse
    import numpy as np
    import torch
    from mobilesuperresolution_amd.models import get_model
    torch.manual_seed(3)
    sd = {"head.0.bias": torch.randn(24), "head.0.weight_g": torch.randn(24, 1, 1, 1), "head.0.weight_v": torch.randn(24, 3, 3, 3),
          "body.0.body.0.bias": torch.randn(144), "body.0.body.0.weight_g": torch.randn(144, 1, 1, 1)}
    path = str(tmp_path / "wdsr_b_x2_4_24.pt")
    torch.save(sd, path)
    ns = argparse.Namespace(model_type="NAS_MODEL", image_mean=0.5, num_channels=3, scale=2, num_blocks=4, num_residual_units=24,
                            width_search=True, length_search=False, pretrained=False)
    torch.manual_seed(0)
    m = get_model(ns)
    before = {k: v.detach().clone() for k, v in m.named_reference_tensors()}
    assert m.load_pretrained(path) == 3
    for k, v in m.named_reference_tensors():
        if k in ("head.bias", "head.weight_g", "head.weight_v"):
            assert torch.equal(v, sd["head.0." + k.split(".")[1]])
        else:
            assert torch.equal(v, before[k]), k
    ns.pretrained, ns.pretrained_path = True, path
    m2 = get_model(ns)
    assert torch.equal(m2.head.weight_v, sd["head.0.weight_v"])
    ns.pretrained_path = str(tmp_path / "missing.pt")
    import pytest
    with pytest.raises(FileNotFoundError):
        get_model(ns)
    # parameter registration order is the reference's (fixture G10 lists the reference's own named_parameters order)
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g10_nas_model.npz"))
    ref_order = [k[2:] for k in z.keys() if k.startswith("p/")]
    ns16 = argparse.Namespace(**{**vars(ns), "pretrained": False, "num_blocks": len({k.split(".")[1] for k in ref_order if k.startswith("body.")})})
    mine = [k for k, _ in get_model(ns16).named_reference_tensors() if not k.startswith("speed_estimator")]
    assert mine == ref_order                                 # the positional walk is over these, in this order


def test_flat_body_parameter_views_and_freezing():
    """NAS_MODEL keeps the 25 tensors per block as views of ONE flat parameter: reference keys round-trip, body[i].x views
    write through, and length_grad / mask_grad / kernel_grad freeze kinds by zeroing their gradient slices"""
    import argparse
    import torch
    from mobilesuperresolution_amd.models import get_model
    from mobilesuperresolution_amd.models.wdsr_b import _SplitFlat
    ns = argparse.Namespace(model_type="NAS_MODEL", image_mean=0.5, num_channels=3, scale=2, num_blocks=3, num_residual_units=24,
                            width_search=True, length_search=False, pretrained=False)
    torch.manual_seed(1)
    m = get_model(ns)
    names = [k for k, _ in m.named_parameters()]
    assert names[0] == "flat" and not any(k.startswith("body.") for k in names)
    assert m.gradless_parameters() == [] and m.freeze_gradless_parameters() == []
    sd = m.state_dict()
    assert sd["body.2.body.7.0.body.0.weight_v"].shape == (24, 1, 7, 7) and sd["body.1.alpha1"].shape == (1,)
    with torch.no_grad():
        m.body[1].alpha1.fill_(1.5)
        m.body[2].split.weight.mul_(0.0)
    assert float(m.state_dict()["body.1.alpha1"]) == 1.5 and float(m.state_dict()["body.2.split.weight"].abs().sum()) == 0.0
    assert m.get_current_blocks() == 2 and m.get_block_status() == [0, 2]
    m2 = get_model(ns)
    m2.load_state_dict(m.state_dict())
    assert torch.equal(m2.flat, m.flat)
    # freezing: gradient slices of the frozen kinds are exactly zero, the others flow
    m.length_grad(False)
    m.mask_grad(False)
    outs = _SplitFlat.apply(m.flat, m._layout, m._frozen)
    sum((o.float() ** 2).sum() for o in outs if o.requires_grad).backward()
    for kind in ("alpha1", "alpha2", "beta1", "beta2", "split.weight"):
        assert float(m.kind(kind, m.flat.grad).abs().sum()) == 0.0, kind
    assert float(m.kind("alpha", m.flat.grad).abs().sum()) > 0 and not m.mask.weight.requires_grad
    m.length_grad(True)
    m.mask_grad(True)
    assert m._frozen == frozenset() and m.mask.weight.requires_grad
    m.kernel_grad(False)
    a = m.kind("alpha")
    assert torch.equal(a.sum(1), torch.ones(3)) and set(a.unique().tolist()) == {0.0, 1.0} and "alpha" in m._frozen


def test_nas_prep_tables_drive_weight_norm_pack_and_gradient_scatter():
    """packing.nas_prep_tables (the tables behind sr_param_pack / sr_param_grads on the supernet body), with the four
    kernels of csrc/wdsr_prep.h restated in numpy: the source rows equal torch's weight-norm of every conv + the biases,
    the operand gathers equal the index_select route, and slab sums scattered and pushed through the weight-norm
    backward equal autograd's gradient of the same weights."""
    import numpy as np
    from mobilesuperresolution_amd import packing as P
    from mobilesuperresolution_amd.models.wdsr_b import _body_kinds
    f, nb = 24, 3
    rng = np.random.default_rng(0)
    layout, off = [], 0
    for name, shape in _body_kinds(f):
        n = nb * int(np.prod(shape))
        layout.append((name, off, n, (nb,) + tuple(shape)))
        off += n
    flat = rng.standard_normal(off).astype(np.float32)
    t = P.nas_prep_tables(f, nb, tuple(layout))
    o, size, ds = t["off"], t["size"], t["ds"]
    lay = {name: (o_, n_, shp) for name, o_, n_, shp in layout}
    kind = lambda name, src=flat: src[lay[name][0]:lay[name][0] + lay[name][1]].reshape(lay[name][2])
    # --- wn_src_kernel restated
    src = np.zeros(nb * size, np.float32)
    for v_off, g_off, K, dst in t["chan_tab"]:
        v = flat[v_off:v_off + K]
        src[dst:dst + K] = v * (flat[g_off] / np.sqrt((v.astype(np.float64) ** 2).sum()))
    for a, b, dst in t["bias_tab"]:
        assert b == -1
        src[dst] = flat[a]
    src = src.reshape(nb, size)
    for ki, k in enumerate((3, 5, 7)):
        for j, col, K in ((0, o[f"wdw{k}"], k * k), (2, o["wpw"] + ki * f * f, f)):
            v = torch.from_numpy(kind(f"body.{k}.0.body.{j}.weight_v").reshape(nb * f, -1))
            g = torch.from_numpy(kind(f"body.{k}.0.body.{j}.weight_g").reshape(nb * f, 1))
            w = torch._weight_norm(v, g, 0).reshape(nb, f * K).numpy()
            np.testing.assert_allclose(src[:, col:col + f * K], w, rtol=2e-6, atol=1e-7)
            bcol = o["bdw" if j == 0 else "bpw"] + ki * f
            np.testing.assert_array_equal(src[:, bcol:bcol + f], kind(f"body.{k}.0.body.{j}.bias"))
    # --- unpack_all_kernel + wn_bwd_kernel restated, against autograd
    base = P.nas_tables(f)
    wgs = 3
    part_pw = rng.standard_normal((nb, wgs, base["pw_slab"])).astype(np.float32)
    part_dw = rng.standard_normal((nb, wgs, base["dw_slab"])).astype(np.float32)
    dsrc = np.zeros((nb, ds), np.float32)
    dsrc[:, t["pw_dst"]] = part_pw.sum(1)[:, t["pw_sidx"]]
    dsrc[:, t["dw_dst"]] = part_dw.sum(1)[:, t["dw_sidx"]]
    spw, sdw = part_pw.sum(1), part_dw.sum(1)
    np.testing.assert_array_equal(dsrc[:, o["wpw"]:o["wpw"] + 3 * f * f], spw[:, base["g_wpw"]])
    np.testing.assert_array_equal(dsrc[:, t["extra"]["r"]:t["extra"]["r"] + 3 * f], spw[:, base["g_r"]])
    np.testing.assert_array_equal(dsrc[:, t["extra"]["sxy"]], spw[:, base["sxy"]])
    np.testing.assert_array_equal(dsrc[:, o["wdw7"]:o["wdw7"] + 49 * f], sdw[:, base["g_wdw"][2]])
    np.testing.assert_array_equal(dsrc[:, t["extra"]["sB"]:t["extra"]["sB"] + f], sdw[:, base["g_sB"]])
    gflat = np.zeros_like(flat)
    dflat = dsrc.reshape(-1)
    for v_off, g_off, K, dst in t["chan_bwd"]:
        v, dw = flat[v_off:v_off + K].astype(np.float64), dflat[dst:dst + K].astype(np.float64)
        ss, dot = (v * v).sum(), (v * dw).sum()
        nrm = np.sqrt(ss)
        gflat[v_off:v_off + K] = (flat[g_off] / nrm) * (dw - v * dot / ss)
        gflat[g_off] = dot / nrm
    for a, b, dst in t["bias_bwd"]:
        gflat[a] = dflat[dst]
    k, j = 5, 2
    v = torch.from_numpy(kind(f"body.{k}.0.body.{j}.weight_v").reshape(nb * f, -1)).requires_grad_(True)
    g = torch.from_numpy(kind(f"body.{k}.0.body.{j}.weight_g").reshape(nb * f, 1)).requires_grad_(True)
    up = torch.from_numpy(dsrc[:, o["wpw"] + f * f:o["wpw"] + 2 * f * f].reshape(nb * f, f).copy())
    torch._weight_norm(v, g, 0).backward(up)
    np.testing.assert_allclose(kind(f"body.{k}.0.body.{j}.weight_v", gflat).reshape(nb * f, -1), v.grad.numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(kind(f"body.{k}.0.body.{j}.weight_g", gflat).reshape(nb * f, 1), g.grad.numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_array_equal(kind("body.3.0.body.0.bias", gflat), dsrc[:, o["bdw"]:o["bdw"] + f])
    # kinds no table names keep a zero gradient from this route
    for name in ("alpha", "beta", "alpha1", "alpha2", "split.weight"):
        assert not kind(name, gflat).any()
    # eval with a skipped block: rows 0, 1 of the buffers belong to blocks 0 and 2
    sub = P.nas_prep_tables(f, nb, tuple(layout), blocks=(0, 2))
    full = t["chan_tab"].reshape(nb, 6 * f, 4)
    got = sub["chan_tab"].reshape(2, 6 * f, 4)
    np.testing.assert_array_equal(got[0], full[0])
    np.testing.assert_array_equal(got[1][:, :3], full[2][:, :3])                    # block 2's parameters ...
    np.testing.assert_array_equal(got[1][:, 3], full[2][:, 3] - size)               # ... into row 1
    np.testing.assert_array_equal(sub["bias_tab"].reshape(2, 6 * f, 3)[1][:, 0], t["bias_tab"].reshape(nb, 6 * f, 3)[2][:, 0])

