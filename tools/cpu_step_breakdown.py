"""Diagnostic: host-side time of each part of the training step (no GPU sync inside the step): is the step
host-bound?  Usage: python tools/cpu_step_breakdown.py [--ddp]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from mobilesuperresolution_amd.models import get_model
import torch.distributed as dist

ddp = "--ddp" in sys.argv
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if ddp:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group(backend="nccl", init_method="env://")
torch.manual_seed(0)
model = get_model(B.model_ns("bf16")).to(dev).train()
net = model
if ddp:
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0, gradient_as_bucket_view=True,
                                                    broadcast_buffers=False)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
x = torch.rand(B.BATCH, 3, B.LR, B.LR, device=dev)
hr = torch.rand(B.BATCH, 3, B.LR * B.SCALE, B.LR * B.SCALE, device=dev)
acc = [0.0] * 5
def step(rec):
    t0 = time.perf_counter(); opt.zero_grad(set_to_none=True)
    t1 = time.perf_counter(); sr = net(x)
    t2 = time.perf_counter(); loss = torch.nn.functional.l1_loss(sr, hr)
    t3 = time.perf_counter(); loss.backward()
    t4 = time.perf_counter(); opt.step()
    t5 = time.perf_counter()
    if rec:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[i] += d
for _ in range(20):
    step(False)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    step(True)
tcpu = time.perf_counter() - t0
torch.cuda.synchronize()
tall = time.perf_counter() - t0
names = ["zero_grad", "forward", "l1_loss", "backward", "opt.step"]
print(("DDP " if ddp else "plain ") + ", ".join(f"{k} {v / n * 1e6:.0f} us" for k, v in zip(names, acc)))
print(f"host issue time {tcpu / n * 1e6:.0f} us/step, wall {tall / n * 1e6:.0f} us/step")
if ddp:
    dist.destroy_process_group()
