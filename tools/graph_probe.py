"""Diagnostic: what a hipGraph replay buys over stream launches, (1) for the block-forward chain, (2) for the
whole training step (torch.cuda.CUDAGraph around fwd + L1 + bwd + Adam)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from mobilesuperresolution_amd import _lib as L
from mobilesuperresolution_amd.models import get_model

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = get_model(B.model_ns("bf16")).to(dev).train()
x = torch.rand(B.BATCH, 3, B.LR, B.LR, device=dev)
hr = torch.rand(B.BATCH, 3, B.LR * B.SCALE, B.LR * B.SCALE, device=dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)


def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.l1_loss(model(x), hr)
    loss.backward()
    opt.step()
    return loss


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print(f"eager step: {timeit(step):.1f} us")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    static_loss = step()
print(f"graphed step: {timeit(g.replay):.1f} us   loss {float(static_loss):.5f}")

# block-forward chain
st = model._state(dev)
a = torch.randn(B.BATCH, B.LR, B.LR, B.UNITS, device=dev).bfloat16()
b, c = torch.empty_like(a), torch.empty_like(a)
reps = 64
def chain1():
    L.check(L.lib().sr_wdsr_block_fwd_repeat(a.data_ptr(), b.data_ptr(), st.blob_body[0].data_ptr(), st.cinit_body[0].data_ptr(),
                                             B.BATCH, B.LR, B.LR, B.UNITS, 1, reps, L.stream_ptr()), "c1")
def chain2():
    L.check(L.lib().sr_wdsr_block2_fwd_repeat(a.data_ptr(), b.data_ptr(), c.data_ptr(), st.blob_body[0].data_ptr(),
                                              st.blob_body[1].data_ptr(), st.cinit_body[0].data_ptr(), st.cinit_body[1].data_ptr(),
                                              B.BATCH, B.LR, B.LR, B.UNITS, 1, reps, L.stream_ptr()), "c2")
for name, ch in (("single", chain1), ("pair", chain2)):
    e = timeit(ch, 20) / reps
    gg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gg):
        ch()
    r = timeit(gg.replay, 20) / reps
    print(f"{name}: eager {e:.2f} us/launch, graph {r:.2f} us/launch")
