"""Host vs GPU time of the reference's training loop on the drop-in route (training.L1Loss / training.Adam)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import training as T
from mobilesuperresolution_amd.models import get_model
ns = argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=16, num_residual_units=24, hot_dtype="bf16")
m = get_model(ns).cuda().train()
x = torch.rand(32, 3, 48, 48, device="cuda"); hr = torch.rand(32, 3, 192, 192, device="cuda")
crit, opt = T.L1Loss(), T.Adam(m.parameters(), 1e-3)
def step(item):
    opt.zero_grad(); sr = m(x); loss = 0; l1 = 1.0 * crit(sr, hr); loss += l1; loss.backward(); opt.step(); opt.zero_grad()
    return loss.item() if item else loss
for _ in range(10): step(True)
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n): step(False)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"no item sync: host issue {1e3*(t1-t0)/n:.4f} ms/step, with final sync {1e3*(t2-t0)/n:.4f} ms/step")
t0 = time.perf_counter()
for _ in range(n): step(True)
t1 = time.perf_counter()
print(f"item sync every step: {1e3*(t1-t0)/n:.4f} ms/step")
st = m.make_train_state(1e-3)
for _ in range(5): m.train_step(x, hr, st).item()
t0 = time.perf_counter()
for _ in range(n): m.train_step(x, hr, st).item()
print(f"train_step + item: {1e3*(time.perf_counter()-t0)/n:.4f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step(False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
