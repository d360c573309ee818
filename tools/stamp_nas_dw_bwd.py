import os, sys
os.environ["SR_HOTPATH_DEBUG_LIB"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import _lib as L, packing as P
f, n, wgs = 32, 32, 256
t = P.nas_tables(f)
yin = torch.randn(n, 48, 48, f, device="cuda").bfloat16()
GZ = torch.randn(3, n, 48, 48, f, device="cuda").bfloat16()
gy = torch.randn_like(yin); gyin = torch.empty_like(yin)
dwp = torch.rand(t["dwp"].size, device="cuda")
part = torch.empty(wgs, t["dw_slab"], device="cuda")
st = torch.zeros(wgs * 16 * 16 * 2, dtype=torch.int64, device="cuda")
lib = L.lib()
def run():
    L.check(lib.sr_nas_dw_bwd(yin.data_ptr(), GZ.data_ptr(), gy.data_ptr(), gyin.data_ptr(), dwp.data_ptr(), part.data_ptr(), wgs, n, 48, 48, f, 1, L.stream_ptr()), "dw_bwd")
for it in range(5): run()
torch.cuda.synchronize()
L.check(lib.sr_debug_set_stamps(st.data_ptr(), wgs), "set")
for it in range(3): run()
torch.cuda.synchronize()
L.check(lib.sr_debug_set_stamps(None, 0), "unset")
raw = st.cpu().numpy().reshape(wgs, 16, 16, 2).astype(np.float64)
s = raw[..., 0] * 10.0
nst = int((s[0, 0] > 0).sum()); nw = int((s[0, :, 0] > 0).sum())
print(nst, "stamps", nw, "waves; span", s[:, :nw, nst-1].max() - s[:, :nw, 0].min())
names = ["pack weights + barrier", "stage k=0", "compute k=0 + barrier", "stage k=1", "compute k=1 + barrier", "stage k=2", "compute k=2 (+epilogue stores)", "reduce + store"]
for k in range(nst - 1):
    d = s[:, :nw, k + 1] - s[:, :nw, k]
    print("%-32s median %6.0f ns  p10 %6.0f p90 %6.0f" % (names[k] if k < len(names) else k, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
