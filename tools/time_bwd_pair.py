"""Timing of the two-block backward-data kernel (C2 shape), for A/B runs of kernel variants via SR_HOTPATH_LIB_PATH."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import _lib as L, hotpath as HP
n, f = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 24
dev = torch.device("cuda", 0)
src = torch.randn(2, HP.tables(f, dev)["src_size"], device="cuda") * 0.1
blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
xa = torch.randn(n, 48, 48, f, device="cuda").bfloat16(); xb = torch.randn_like(xa); dyb = torch.randn_like(xa)
d1, d0 = torch.empty_like(xa), torch.empty_like(xa)
dta = torch.empty(n * 8 * 288 * 24, device="cuda", dtype=torch.bfloat16); dtb = torch.empty_like(dta)
lib = L.lib()
def run():
    L.check(lib.sr_wdsr_block2_bwd_data(xa.data_ptr(), xb.data_ptr(), dyb.data_ptr(), d1.data_ptr(), d0.data_ptr(), blob[0].data_ptr(),
                                        blob[1].data_ptr(), cinit[0].data_ptr(), cinit[1].data_ptr(), dta.data_ptr(), dtb.data_ptr(),
                                        n, 48, 48, f, 1, L.stream_ptr()), "pair bwd")
for _ in range(20): run()
res = []
for rep in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) * 5)
res.sort()
print("pair bwd_data batch %d: median %.2f us per launch (min %.2f)" % (n, res[3], res[0]))
