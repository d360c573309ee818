"""Diagnostic: inner timeline of the one-block register-resident forward (rs1) from explicit stamp slots (SR_STAMP_AT).
    python -m mobilesuperresolution_amd.build --debug; python tools/stamp_fwd_inner.py [batch]"""
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
x = torch.randn(n, 48, 48, f, device="cuda").bfloat16()
yb = torch.empty_like(x)
nwg = n * 8
st = torch.zeros(nwg * 16 * 16 * 2, dtype=torch.int64, device="cuda")
lib = L.lib()
def run():
    L.check(lib.sr_wdsr_fwd_rs(x.data_ptr(), None, yb.data_ptr(), blob[0].data_ptr(), None, cinit[0].data_ptr(), None, None, None, 1, n, 48, 48,
                               f, 1, L.stream_ptr()), "rs")
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
# program order of the slots: 0 start, 1 staged, 2 after barrier, [phase A: 11 x frags read, 12 first tile's chain done, 13 stored], 3 A done,
# 4 after barrier, [phase B: 6 start, 7 addresses, 8 first chain done, 9 loop done, 10 stored], 5 end
order = [0, 1, 2, 11, 12, 13, 3, 4, 6, 7, 8, 9, 10, 5]
names = ["start", "staged", "barrier0", "A: x frags issued", "A: chain 1 done", "A: tile 1 stored", "A done", "barrier1", "B: start", "B: addresses",
         "B: chain 1 done", "B: loop done", "B: stored", "end"]
for w in (0, 7):
    print(f"wave {w}: median ns (cycles) since the previous slot")
    for a, b, nm in zip(order[:-1], order[1:], names[1:]):
        d = ns[:, w, b] - ns[:, w, a]
        dc = cyc[:, w, b] - cyc[:, w, a]
        print("  %-20s %6.0f ns %6.0f cyc" % (nm, np.median(d), np.median(dc)))
