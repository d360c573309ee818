"""Diagnostic: block-forward kernel chain time over a batch sweep (bf16, F = 24)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from mobilesuperresolution_amd import _lib as L
from mobilesuperresolution_amd.models import get_model
dev = torch.device("cuda", 0)
m = get_model(B.model_ns("bf16")).to(dev)
st = m._state(dev)
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for batch in (32, 64, 128, 256, 512):
    a = torch.randn(batch, 48, 48, 24, device=dev).bfloat16(); bb = torch.empty_like(a)
    reps = 64
    f1 = lambda: L.check(L.lib().sr_wdsr_block_fwd_repeat(a.data_ptr(), bb.data_ptr(), st.blob_body[0].data_ptr(),
                         st.cinit_body[0].data_ptr(), batch, 48, 48, 24, 1, reps, L.stream_ptr()), "r1")
    u1 = timeit(f1) / reps
    print(f"batch {batch}: {u1 * 1e6:.2f} us/launch, {2 * batch * 48 * 48 * 24 * 2 / u1 / 1e9:.0f} GB/s")
