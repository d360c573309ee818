"""Diagnostic: per-phase, per-wave timeline of nas_pw_bwd_kernel from in-kernel stamps (C5 shape: batch 32, 48x48, 32 units).
Needs the diagnostic library: python -m mobilesuperresolution_amd.build --debug.
    python tools/stamp_nas.py [units]"""
import os, sys
os.environ["SR_HOTPATH_DEBUG_LIB"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import _lib as L, packing as P
f = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n, wgs = 32, 256
t = P.nas_tables(f)
yin = torch.randn(n, 48, 48, f, device="cuda").bfloat16()
V = torch.randn(3, n, 48, 48, f, device="cuda").bfloat16()
gy = torch.randn_like(yin); GZ = torch.empty_like(V)
frags = (torch.randn(12 * 512, device="cuda") * 0.1).bfloat16()
tabs = torch.rand(160, device="cuda"); scal = torch.rand(4, device="cuda")
part = torch.empty(wgs, t["pw_slab"], device="cuda")
st = torch.zeros(wgs * 16 * 16 * 2, dtype=torch.int64, device="cuda")
lib = L.lib()
def run():
    L.check(lib.sr_nas_pw_bwd(yin.data_ptr(), V.data_ptr(), gy.data_ptr(), GZ.data_ptr(), frags.data_ptr(), tabs.data_ptr(),
                              scal.data_ptr(), part.data_ptr(), wgs, n, 48, 48, f, 1, L.stream_ptr()), "pw_bwd")
for it in range(5):
    run()
torch.cuda.synchronize()
L.check(lib.sr_debug_set_stamps(st.data_ptr(), wgs), "set")
for it in range(3):
    run()
torch.cuda.synchronize()
L.check(lib.sr_debug_set_stamps(None, 0), "unset")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(20):
    run()
e1.record(); torch.cuda.synchronize()
print("pw_bwd: %.1f us per launch (20 back to back)" % (e0.elapsed_time(e1) * 50))
raw = st.cpu().numpy().reshape(wgs, 16, 16, 2).astype(np.float64)
s = raw[..., 0] * 10.0
nst = int((s[0, 0] > 0).sum()); nw = int((s[0, :, 0] > 0).sum()); print("stamps", nst, "waves", nw, raw[0, 0, :8, 0])
t0 = s[:, :nw, 0].min()
print(f"{wgs} workgroups, {nw} waves, {nst} stamps; first start -> last end {s[:, :nw, nst - 1].max() - t0:.0f} ns; start spread {s[:, 0, 0].max() - t0:.0f} ns")
names = ["stage V + weights", "sxy pass", "items", "scale + butterfly + barrier", "slab copies + barrier", "sum copies + store"]
for k in range(nst - 1):
    d = s[:, :nw, k + 1] - s[:, :nw, k]
    print("%-20s median %6.0f ns  p10 %6.0f  p90 %6.0f   per-wave median: %s" % (names[k] if k < len(names) else k, np.median(d),
          np.percentile(d, 10), np.percentile(d, 90), " ".join("%4.0f" % v for v in np.median(d, axis=0))))
