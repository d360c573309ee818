"""Diagnostic: run the data-parallel train_step on one rank (RCCL) so that rocprofv3 --kernel-trace can show its timeline."""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl")
import bench as B
from mobilesuperresolution_amd.models import get_model
dev = torch.device("cuda", 0)
m = get_model(B.model_ns("bf16")).to(dev).train()
st = m.make_train_state(1e-3)
x = torch.rand(32, 3, 48, 48, device=dev); hr = torch.rand(32, 3, 192, 192, device=dev)
pg = dist.group.WORLD if os.environ.get("DP", "1") == "1" else None
for _ in range(60): m.train_step(x, hr, st, process_group=pg)
torch.cuda.synchronize()
dist.destroy_process_group()
