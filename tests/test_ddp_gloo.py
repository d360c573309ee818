"""World-size-2 (gloo, CPU) coverage of the N>1 path: the model exposes ONE flat parameter
(mobilesuperresolution_amd/layout.py).  A CPU stand-in that computes the oracle's forward from views of
that flat parameter checks (i) the layout against the per-tensor oracle, (ii) that DDP over the single
flat parameter averages gradients to the full-batch gradient (reference semantics: pretrain.py:216,239)."""
import argparse
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from mobilesuperresolution_amd.layout import get_layout
from oracle import wdsr_oracle as O


class FlatOracle(nn.Module):
    def __init__(self, ns):
        super().__init__()
        self.layout = get_layout(ns.num_residual_units, ns.num_blocks, ns.scale)
        self.scale, self.mean = ns.scale, ns.image_mean
        torch.manual_seed(0)
        ref = O.OracleBasicModel(ns)
        flat = torch.zeros(self.layout.total)
        for k, v in ref.state_dict().items():
            off, shape = self.layout.entries[k]
            flat[off:off + v.numel()] = v.reshape(-1)
        self.flat = nn.Parameter(flat)

    def views(self):
        return {k: self.flat[off:off + int(np.prod(shape))].view(shape) for k, (off, shape) in self.layout.entries.items()}

    def forward(self, x):
        return O.basic_model_forward(x, self.views(), self.scale, self.mean)


def _ns():
    return argparse.Namespace(image_mean=0.5, num_channels=3, scale=4, num_blocks=2, num_residual_units=24)


def test_flat_layout_matches_per_tensor_oracle():
    ns = _ns()
    m = FlatOracle(ns)
    torch.manual_seed(0)
    ref = O.OracleBasicModel(ns)
    x = torch.rand(2, 3, 12, 10)
    y, yr = m(x), ref(x)
    assert torch.equal(y, yr)
    y.abs().mean().backward()
    yr.abs().mean().backward()
    for k, p in ref.named_parameters():
        off, shape = m.layout.entries[k]
        assert torch.allclose(m.flat.grad[off:off + p.numel()].view(shape), p.grad, rtol=1e-5, atol=1e-8), k
    assert m.layout.total == sum(p.numel() for p in ref.parameters())


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    ns = _ns()
    m = FlatOracle(ns)
    ddp = nn.parallel.DistributedDataParallel(m)
    g = torch.Generator().manual_seed(123)
    x = torch.rand(4, 3, 12, 10, generator=g)
    hr = torch.rand(4, 3, 48, 40, generator=g)
    xs, hs = x[rank * 2:(rank + 1) * 2], hr[rank * 2:(rank + 1) * 2]       # disjoint shard per rank
    torch.nn.functional.l1_loss(ddp(xs), hs).backward()
    if rank == 0:
        torch.save(m.flat.grad.clone(), out)
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_world2_matches_full_batch(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    m = FlatOracle(_ns())
    g = torch.Generator().manual_seed(123)
    x = torch.rand(4, 3, 12, 10, generator=g)
    hr = torch.rand(4, 3, 48, 40, generator=g)
    torch.nn.functional.l1_loss(m(x), hr).backward()
    assert torch.allclose(got, m.flat.grad, rtol=1e-4, atol=1e-7)
