"""Workload for the rocprofv3 --pmc passes: the two-block forward kernel alone, C2 shape (batch 32, 48x48,
F=24, bf16), 64 back-to-back launches over three 3.5 MB buffers; the SAVE_T variant (it also writes the two t images)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import _lib as L, hotpath as HP
f, n = 24, 32
src = torch.randn(2, HP.tables(f, torch.device("cuda", 0))["src_size"], device="cuda") * 0.05
src[:, -2], src[:, -1] = 0.0, 1.0
blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
x = torch.randn(n, 48, 48, f, device="cuda").bfloat16()
ya, yb = torch.empty_like(x), torch.empty_like(x)
ts = torch.empty((2, n, 8, 288, 24), device="cuda", dtype=torch.bfloat16)      # SAVE_T: the variant the training step launches
L.check(L.lib().sr_wdsr_fwd_rs_repeat(x.data_ptr(), ya.data_ptr(), yb.data_ptr(), blob[0].data_ptr(), blob[1].data_ptr(),
                                      cinit[0].data_ptr(), cinit[1].data_ptr(), ts[0].data_ptr(), ts[1].data_ptr(), 2, n, 48, 48, f, 1, 64,
                                      L.stream_ptr()), "repeat")
torch.cuda.synchronize()
print("done")
