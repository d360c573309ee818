"""Driver for PMC passes: the two-block forward launch at a given batch, back to back (python3 tools/pmc_fwd_pair.py [batch] [launches])"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import _lib as L, hotpath as HP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
f = 24
dev = torch.device("cuda", 0)
src = torch.randn(2, HP.tables(f, dev)["src_size"], device="cuda") * 0.1
blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
x = torch.randn(n, 48, 48, f, device="cuda").bfloat16()
ya, yb = torch.empty_like(x), torch.empty_like(x)
ts = torch.empty((2, n, 8, 288, 24), device='cuda', dtype=torch.bfloat16) if os.environ.get('SAVE_T') else None
for _ in range(2):
    L.check(L.lib().sr_wdsr_fwd_rs_repeat(x.data_ptr(), ya.data_ptr(), yb.data_ptr(), blob[0].data_ptr(), blob[1].data_ptr(), cinit[0].data_ptr(),
                                          cinit[1].data_ptr(), ts[0].data_ptr() if ts is not None else None,
                                          ts[1].data_ptr() if ts is not None else None, 2, n, 48, 48, f, 1, reps, L.stream_ptr()), "repeat")
torch.cuda.synchronize()
