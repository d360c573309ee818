"""Workload for profiling: C3 shape per GPU (16 blocks / 32 units / batch 32, bf16), fused training step."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd.models import get_model
dev = torch.device("cuda", 0)
ns = argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=16, num_residual_units=32, hot_dtype="bf16")
torch.manual_seed(0)
m = get_model(ns).to(dev).train()
st = m.make_train_state(1e-3)
x = torch.rand(32, 3, 48, 48, device=dev); hr = torch.rand(32, 3, 192, 192, device=dev)
for _ in range(5): m.train_step(x, hr, st)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get("C3_STEPS", 30))
for _ in range(n): m.train_step(x, hr, st)
torch.cuda.synchronize()
print(f"C3 step: {(time.perf_counter() - t0) / n * 1e3:.3f} ms")
