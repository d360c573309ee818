"""Diagnostic: per-phase, per-wave timeline of wdsr_bwd_rs_kernel (sr_wdsr_block2_bwd_data) from in-kernel stamps (C2 shape: batch 32, 48x48, 24 units).
Needs the diagnostic library: python -m mobilesuperresolution_amd.build --debug.
    python tools/stamp_bwd.py [batch]"""
import os, sys
os.environ["SR_HOTPATH_DEBUG_LIB"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import _lib as L, hotpath as HP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
f = 24
dev = torch.device("cuda", 0)
src = torch.randn(2, HP.tables(f, dev)["src_size"], device="cuda") * 0.1
blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
xa = torch.randn(n, 48, 48, f, device="cuda").bfloat16(); xb = torch.randn_like(xa); dyb = torch.randn_like(xa)
d1, d0 = torch.empty_like(xa), torch.empty_like(xa)
tiles = n * 8
dta = torch.empty(tiles * 288 * 24, device="cuda", dtype=torch.bfloat16); dtb = torch.empty_like(dta)
st = torch.zeros(tiles * 16 * 16 * 2, dtype=torch.int64, device="cuda")
lib = L.lib()
def run():
    L.check(lib.sr_wdsr_block2_bwd_data(xa.data_ptr(), xb.data_ptr(), dyb.data_ptr(), d1.data_ptr(), d0.data_ptr(), blob[0].data_ptr(),
                                        blob[1].data_ptr(), cinit[0].data_ptr(), cinit[1].data_ptr(), dta.data_ptr(), dtb.data_ptr(),
                                        n, 48, 48, f, 1, L.stream_ptr()), "pair bwd")
for it in range(5):
    run()
torch.cuda.synchronize()
L.check(lib.sr_debug_set_stamps(st.data_ptr(), tiles), "set")
for it in range(3):
    run()
torch.cuda.synchronize()
L.check(lib.sr_debug_set_stamps(None, 0), "unset")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(20):
    run()
e1.record(); torch.cuda.synchronize()
print("pair bwd_data: %.1f us per launch (20 back to back)" % (e0.elapsed_time(e1) * 50))
raw = st.cpu().numpy().reshape(tiles, 16, 16, 2).astype(np.float64)
s = raw[..., 0] * 10.0
nst = int((s[0, 0] > 0).sum()); nw = int((s[0, :, 0] > 0).sum())
t0 = s[:, :nw, 0].min()
print(f"{tiles} workgroups, {nw} waves, {nst} stamps; first start -> last end {s[:, :nw, nst - 1].max() - t0:.0f} ns; start spread {s[:, 0, 0].max() - t0:.0f} ns")
names = ["issue staging (set 1 + set 2)", "wait set 1 + barrier", "P1b dt_b (12 tiles)", "wait set 2 + barrier + load w2 + barrier + issue Wa", "P2b dx_b (12 tiles)", "wait Wa + barrier", "issue x_a + load w1 + P1a dt_a (9 tiles)", "wait x_a + barrier", "P2a dx_a (9 tiles)"]
for k in range(nst - 1):
    d = s[:, :nw, k + 1] - s[:, :nw, k]
    print("%-52s median %6.0f ns  p10 %6.0f  p90 %6.0f   per-wave median: %s" % (names[k] if k < len(names) else k, np.median(d),
          np.percentile(d, 10), np.percentile(d, 90), " ".join("%4.0f" % v for v in np.median(d, axis=0))))
