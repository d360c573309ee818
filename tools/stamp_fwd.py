"""Diagnostic: per-phase timeline of the fused block forward kernel from in-kernel stamps."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import _lib as L, hotpath as HP
f, n = 24, 32
src = torch.randn(1, HP.tables(f, torch.device("cuda", 0))["src_size"], device="cuda") * 0.1
blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
x = torch.randn(n, 48, 48, f, device="cuda").bfloat16()
y = torch.empty_like(x)
nwg = n * 8
st = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
for it in range(5):
    HP.block_fwd(x, y, blob[0], cinit[0])
    L.check(L.lib().sr_wdsr_block_fwd_stamps(x.data_ptr(), y.data_ptr(), blob[0].data_ptr(), cinit[0].data_ptr(),
                                             n, 48, 48, f, 1, st.data_ptr(), L.stream_ptr()), "stamps")
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(nwg, 8)[:, :6].astype(np.float64) * 10.0      # ns (100 MHz)
t0 = s[:, 0].min()
print("workgroup start spread (ns): min %.0f max %.0f" % (0, s[:, 0].max() - t0))
names = ["stage issue->stored", "barrier1", "phase A", "barrier2", "phase B + store"]
for k in range(5):
    d = s[:, k + 1] - s[:, k]
    print("%-22s mean %7.0f ns  min %7.0f  max %7.0f" % (names[k], d.mean(), d.min(), d.max()))
print("per-WG total mean %.0f ns; first start -> last end %.0f ns" % ((s[:, 5] - s[:, 0]).mean(), s[:, 5].max() - t0))
