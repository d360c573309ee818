"""Training-step throughput of the C2 network over the batch size (bf16): the network switches from two-block to
single-block launches above 384 workgroups."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from mobilesuperresolution_amd.models import get_model
dev = torch.device("cuda", 0)
for batch in (16, 32, 48, 64, 128, 256):
    torch.manual_seed(0)
    m = get_model(B.model_ns("bf16")).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
    x = torch.rand(batch, 3, 48, 48, device=dev); hr = torch.rand(batch, 3, 192, 192, device=dev)
    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.l1_loss(m(x), hr).backward()
        opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 30
    print(f"batch {batch}: {t * 1e3:.3f} ms/step, {batch * 0.036864 / t:.0f} HR-Mpix/s")
