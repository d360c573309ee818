"""Diagnostic: one steady-state round (round 8) of the streaming forward kernel from explicit stamp slots.
    python -m mobilesuperresolution_amd.build --debug; python tools/stamp_stream.py [batch]"""
import os, sys
os.environ["SR_HOTPATH_DEBUG_LIB"] = "1"
os.environ.setdefault("SR_STREAM8", "1")   # the stamped kernel is the eight-wave form (diagnostic build)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import _lib as L, hotpath as HP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
f = 24
dev = torch.device("cuda", 0)
src = torch.randn(2, HP.tables(f, dev)["src_size"], device="cuda") * 0.1
blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
x = torch.randn(n, 48, 48, f, device="cuda").bfloat16()
ya, yb = torch.empty_like(x), torch.empty_like(x)
nwg = n
st = torch.zeros(nwg * 16 * 16 * 2, dtype=torch.int64, device="cuda")
lib = L.lib()
def run():
    L.check(lib.sr_wdsr_fwd_rs(x.data_ptr(), ya.data_ptr(), yb.data_ptr(), blob[0].data_ptr(), blob[1].data_ptr(), cinit[0].data_ptr(),
                               cinit[1].data_ptr(), None, None, 2, n, 48, 48, f, 1, L.stream_ptr()), "rs")
for it in range(5):
    run()
torch.cuda.synchronize()
L.check(lib.sr_debug_set_stamps(st.data_ptr(), nwg), "set")
for it in range(3):
    run()
torch.cuda.synchronize()
L.check(lib.sr_debug_set_stamps(None, 0), "unset")
raw = st.cpu().numpy().reshape(nwg, 16, 16, 2).astype(np.float64)
ns, cyc = raw[..., 0] * 10.0, raw[..., 1]
names = ["dma issued", "phase 1 done", "phase 2 done", "vmcnt wait", "barrier"]
print("round 8, per wave: median ns since the previous slot (a stamp itself costs ~200 ns)")
for w in range(8):
    row = []
    for k in range(5):
        row.append(np.median(ns[:, w, k + 1] - ns[:, w, k]))
    print("wave %d (group %d): " % (w, w >> 2) + "  ".join("%s %5.0f" % (nm, v) for nm, v in zip(names, row)) +
          "   round total %5.0f ns, clock %.2f GHz" % (np.median(ns[:, w, 5] - ns[:, w, 0]), np.median((cyc[:, w, 5] - cyc[:, w, 0]) / (ns[:, w, 5] - ns[:, w, 0]))))
print("prologue (start -> first round) median %.0f ns; rounds 0 .. NB+4: %.0f ns; whole workgroup %.0f ns" % (
    np.median(ns[:, 0, 9] - ns[:, 0, 8]), np.median(ns[:, 0, 7] - ns[:, 0, 6]), np.median(ns[:, 0, 7] - ns[:, 0, 8])))
