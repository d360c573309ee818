"""Diagnostic: per-phase, per-wave timeline of the block forward kernels from in-kernel stamps.
Needs the diagnostic library: python -m mobilesuperresolution_amd.build --debug; run with SR_HOTPATH_DEBUG_LIB=1.
    SR_HOTPATH_DEBUG_LIB=1 python tools/stamp_fwd.py [pair|single] [batch]"""
import os, sys
os.environ["SR_HOTPATH_DEBUG_LIB"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import _lib as L, hotpath as HP
mode = sys.argv[1] if len(sys.argv) > 1 else "pair"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
f = 24
dev = torch.device("cuda", 0)
src = torch.randn(2, HP.tables(f, dev)["src_size"], device="cuda") * 0.1
blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
x = torch.randn(n, 48, 48, f, device="cuda").bfloat16()
ya = torch.empty_like(x); yb = torch.empty_like(x)
nwg = n * 8
st = torch.zeros(nwg * 16 * 16 * 2, dtype=torch.int64, device="cuda")
lib = L.lib()
def run():
    if mode in ("rs1", "rs2"):
        nb = int(mode[2])
        L.check(lib.sr_wdsr_fwd_rs(x.data_ptr(), ya.data_ptr(), yb.data_ptr(), blob[0].data_ptr(), blob[1].data_ptr(),
                                   cinit[0].data_ptr(), cinit[1].data_ptr(), None, None, nb, n, 48, 48, f, 1, L.stream_ptr()), "rs")
    elif mode == "pair":
        L.check(lib.sr_wdsr_block2_fwd(x.data_ptr(), ya.data_ptr(), yb.data_ptr(), blob[0].data_ptr(), blob[1].data_ptr(),
                                       cinit[0].data_ptr(), cinit[1].data_ptr(), None, None, n, 48, 48, f, 1, L.stream_ptr()), "pair")
    else:
        L.check(lib.sr_wdsr_block_fwd(x.data_ptr(), ya.data_ptr(), blob[0].data_ptr(), cinit[0].data_ptr(), n, 48, 48, f, 1,
                                      L.stream_ptr()), "single")
for it in range(5):
    run()
torch.cuda.synchronize()
L.check(lib.sr_debug_set_stamps(st.data_ptr(), nwg), "set")
for it in range(3):
    run()
torch.cuda.synchronize()
L.check(lib.sr_debug_set_stamps(None, 0), "unset")
raw = st.cpu().numpy().reshape(nwg, 16, 16, 2).astype(np.float64)
s = raw[..., 0] * 10.0                                                    # ns (100 MHz)
cyc = raw[..., 1]
nst = int((s[0, 0] > 0).sum())
nw = int((s[0, :, 0] > 0).sum())
t0 = s[:, :nw, 0].min()
print(f"{mode} batch {n}: {nwg} workgroups, {nw} waves, {nst} stamps; first start -> last end {s[:, :nw, nst - 1].max() - t0:.0f} ns")
print("workgroup start spread %.0f ns" % (s[:, 0, 0].max() - t0))
names = (["stage issue", "wait+barrier", "A1", "barrier", "B1", "barrier", "A2", "barrier", "B2+store"] if mode in ("pair", "rs2")
         else ["stage issue", "wait+barrier", "A", "barrier", "B+store"])
for k in range(nst - 1):
    d = s[:, :nw, k + 1] - s[:, :nw, k]
    act = d[d < 1e6]
    per_wave = np.median(d, axis=0)
    dcy = cyc[:, :nw, k + 1] - cyc[:, :nw, k]
    print("%-14s median %6.0f ns = %6.0f cycles (%.2f GHz)  p10 %6.0f  p90 %6.0f   per-wave median ns: %s" % (
          names[k] if k < len(names) else k, np.median(act), np.median(dcy), np.median(dcy) / max(np.median(act), 1), np.percentile(act, 10),
          np.percentile(act, 90), " ".join("%4.0f" % v for v in per_wave)))
dt = s[:, :nw, nst - 1] - s[:, :nw, 0]
dc = cyc[:, :nw, nst - 1] - cyc[:, :nw, 0]
print("shader clock over the kernel: median %.2f GHz" % np.median(dc / dt))
tot = s[:, :nw, nst - 1].max(axis=1) - s[:, :nw, 0].min(axis=1)
print("per-WG total median %.0f ns" % np.median(tot))
