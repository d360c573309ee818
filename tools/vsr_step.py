"""Workload for profiling: BasicVSR propagation forward+backward (C4 shape: 8 clips x 5 frames x 64x64, bf16)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd.models import ConvResidualBlocks, flow_warp
from mobilesuperresolution_amd.models.basicvsr_arch import propagate
dev = torch.device("cuda", 0)
b, n, h, w = 8, 5, 64, 64
torch.manual_seed(0)
bt = ConvResidualBlocks(27, 24, 8, hot_dtype="bf16").to(dev)
ft = ConvResidualBlocks(27, 24, 8, hot_dtype="bf16").to(dev)
clip = torch.rand(b, n, 3, h, w, device=dev)
fl_f = torch.rand(b, n - 1, 2, h, w, device=dev) * 4 - 2
fl_b = torch.rand(b, n - 1, 2, h, w, device=dev) * 4 - 2
params = list(bt.parameters()) + list(ft.parameters())
gones = torch.ones(b, 24, h, w, device=dev)
def step():
    for p in params:
        p.grad = None
    ob, of = propagate(clip, fl_f, fl_b, bt, ft, flow_warp)
    if os.environ.get("VSR_SUM_LOSS"):
        (sum(o.sum() for o in ob) + sum(o.sum() for o in of)).backward()
    else:                                                # the same gradient without the harness's ten reductions and nine adds
        torch.autograd.backward(ob + of, [gones] * (len(ob) + len(of)))
for _ in range(3):
    step()
torch.cuda.synchronize()
k = int(os.environ.get("VSR_STEPS", 10))
t0 = time.perf_counter()
for _ in range(k):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"VSR fwd+bwd: host issue {(t1 - t0) / k * 1e3:.2f} ms, wall {(t2 - t0) / k * 1e3:.2f} ms")
