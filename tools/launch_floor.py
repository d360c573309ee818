"""Diagnostic: fixed cost of a dependent kernel launch (empty kernel, same grid / workgroup shape as ours)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import _lib as L
out = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")
reps = 400
for (gx, gy, thr, lds) in [(8, 32, 64, 4), (8, 32, 768, 65536), (8, 32, 896, 147456), (8, 32, 768, 160000), (8, 64, 768, 65536),
                           (16, 16, 640, 150000), (8, 256, 768, 65536)]:
    def run():
        L.check(L.lib().sr_probe_launch_floor(out.data_ptr(), gx, gy, thr, lds, reps, L.stream_ptr()), "floor")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    print(f"grid {gx}x{gy} threads {thr} lds {lds}: {e0.elapsed_time(e1) * 1e3 / reps:.2f} us per launch")
import ctypes
res, host = ctypes.c_float(0), ctypes.c_float(0)
for (gx, gy, thr, lds, reps) in [(8, 32, 64, 4, 100), (8, 32, 896, 147456, 100), (8, 32, 896, 147456, 12)]:
    L.check(L.lib().sr_probe_launch_floor_graph(out.data_ptr(), gx, gy, thr, lds, reps, 20, ctypes.addressof(res),
                                                ctypes.addressof(host)), "graph")
    print(f"hipGraph of {reps} kernels: grid {gx}x{gy} threads {thr} lds {lds}: {res.value:.2f} us per kernel on the GPU, "
          f"{host.value:.1f} us of host time per graph launch")
