"""Host time of each statement of the reference's training loop on the drop-in route (GPU idle before each: pure issue cost)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import training as T
from mobilesuperresolution_amd.models import get_model
ns = argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=16, num_residual_units=24, hot_dtype="bf16")
m = get_model(ns).cuda().train()
x = torch.rand(32, 3, 48, 48, device="cuda"); hr = torch.rand(32, 3, 192, 192, device="cuda")
crit, opt = T.L1Loss(), T.Adam(m.parameters(), 1e-3)
acc = {}
def t(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return r
n = 200
for it in range(n + 20):
    if it == 20:
        acc.clear()
    t("zero_grad", opt.zero_grad)
    sr = t("model(x)", lambda: m(x))
    loss = t("1.0 * crit(sr, hr)", lambda: 1.0 * crit(sr, hr))
    t("loss.backward()", loss.backward)
    t("opt.step()", opt.step)
    t("loss.item()", loss.item)
for k, v in acc.items():
    print(f"{k:22s} {v / n * 1e6:7.1f} us host")
