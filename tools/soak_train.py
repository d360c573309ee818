"""Soak: 500 Adam steps on a fixed synthetic batch, bf16 throughput mode vs fp32 parity mode of the hot path
(same init): the losses must fall and track each other."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd.models import get_model
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
hr = torch.rand(16, 3, 96, 96, generator=g)
x = torch.nn.functional.avg_pool2d(hr, 4).to(dev)        # a learnable x4 problem
hr = hr.to(dev)
curves = {}
for mode in ("fp32", "bf16"):
    torch.manual_seed(1)
    ns = argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=8,
                            num_residual_units=24, hot_dtype=mode)
    m = get_model(ns).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    c = []
    for it in range(500):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.l1_loss(m(x), hr)
        loss.backward()
        opt.step()
        if it % 50 == 0 or it == 499:
            c.append(float(loss.detach()))
    curves[mode] = c
    assert all(torch.isfinite(p).all() for p in m.parameters())
print("fp32:", " ".join(f"{v:.4f}" for v in curves["fp32"]))
print("bf16:", " ".join(f"{v:.4f}" for v in curves["bf16"]))
rel = max(abs(a - b) / a for a, b in zip(curves["fp32"], curves["bf16"]))
print(f"max relative gap between the curves: {rel:.3f}; final {curves['fp32'][-1]:.4f} vs {curves['bf16'][-1]:.4f}")
