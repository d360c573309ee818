"""Diagnostic: per-phase timeline of sr_tail_wgrad from in-kernel stamps (C2 shape)."""
import os, sys
os.environ["SR_HOTPATH_DEBUG_LIB"] = "1"          # the diagnostic build (python -m mobilesuperresolution_amd.build --debug)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from mobilesuperresolution_amd import _lib as L
from mobilesuperresolution_amd.models import get_model
dev = torch.device("cuda", 0)
m = get_model(B.model_ns("bf16")).to(dev).train()
x = torch.rand(B.BATCH, 3, B.LR, B.LR, device=dev)
st = torch.zeros(256 * 32, dtype=torch.int64, device=dev)
for _ in range(3):
    m.flat.grad = None
    m(x).sum().backward()
torch.cuda.synchronize()
L.check(L.lib().sr_debug_set_stamps(st.data_ptr()), "set")
m.flat.grad = None
m(x).sum().backward()
torch.cuda.synchronize()
L.check(L.lib().sr_debug_set_stamps(None), "unset")
a = st.cpu().numpy().reshape(256, 32).astype(np.float64) * 10.0
wgs = int((a[:, 0] > 0).sum())
a = a[:wgs]
n = int((a[0] > 0).sum())
print(f"{wgs} workgroups, {n} stamps; kernel span {a[:, :n].max() - a[:, 0].min():.0f} ns")
names = ["top barrier (+ compute of the previous tile)", "stage dconv", "stage feat+img", "barrier"]
d = np.mean(a[:, 1:n] - a[:, :n - 1], axis=0)
for k in range(n - 3):
    print(f"  tile {k // 4} {names[k % 4]:50s} {d[k]:7.0f} ns")
print(f"  compute of the last tile {d[n - 3]:7.0f} ns; epilogue store {d[n - 2]:7.0f} ns")
