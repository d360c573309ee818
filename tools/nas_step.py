"""Workload for profiling: NAS supernet training steps (C5 shape: 16 blocks / 32 units / batch 32, bf16)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd.models import get_model
dev = torch.device("cuda", 0)
ns = argparse.Namespace(model_type="NAS_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=16,
                        num_residual_units=32, width_search=True, pretrained=False, hot_dtype="bf16")
torch.manual_seed(0)
m = get_model(ns).to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
x = torch.rand(32, 3, 48, 48, device=dev)
hr = torch.rand(32, 3, 192, 192, device=dev)
from mobilesuperresolution_amd.training import L1Loss
crit = torch.nn.L1Loss() if os.environ.get("NAS_TORCH_LOSS") else L1Loss()      # search.py:261 criterions['l1']
def step():
    opt.zero_grad(set_to_none=True)
    out, speed = m(x)
    (1.0 * crit(out, hr) + 0.1 * speed.sum()).backward()                       # search.py:74-89
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get("NAS_STEPS", 10))
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"NAS step: host issue {(t1 - t0) / n * 1e3:.2f} ms, wall {(t2 - t0) / n * 1e3:.2f} ms")
