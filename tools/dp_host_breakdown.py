"""Diagnostic: host-side cost of each piece of the data-parallel train_step (one rank, RCCL), no syncs inside the loop."""
import ctypes, os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl")
import bench as B
from mobilesuperresolution_amd import _lib as L
from mobilesuperresolution_amd.models import get_model
dev = torch.device("cuda", 0)
m = get_model(B.model_ns("bf16")).to(dev).train()
st = m.make_train_state(1e-3)
x = torch.rand(32, 3, 48, 48, device=dev); hr = torch.rand(32, 3, 192, 192, device=dev)
for pg in (None, dist.group.WORLD):
    for _ in range(10): m.train_step(x, hr, st, process_group=pg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): m.train_step(x, hr, st, process_group=pg)
    t_host = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 200
    print(f"process_group={'WORLD' if pg else None}: host issue {t_host * 1e3:.3f} ms/step, wall {t_all * 1e3:.3f} ms/step")
g = torch.empty(191368, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200):
    h = dist.all_reduce(g, op=dist.ReduceOp.AVG, async_op=True); h.wait()
t = (time.perf_counter() - t0) / 200; torch.cuda.synchronize()
print(f"all_reduce(async)+wait host cost {t * 1e6:.1f} us; wall {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
dist.destroy_process_group()
