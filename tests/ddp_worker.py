"""Worker of tests/test_gpu_ddp.py (not a test module): one rank of a DistributedDataParallel run of the HIP BASIC_MODEL.
Launched with torch.distributed.run; on a one-GPU box both ranks share cuda:0 and talk through gloo."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--mode", default="wrapper", choices=["wrapper", "fused"])
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(os.environ.get("SR_DDP_BACKEND", "gloo"), init_method="env://")
    torch.cuda.set_device(0)                                  # one-GPU box: every rank on cuda:0
    from mobilesuperresolution_amd.models import get_model
    from mobilesuperresolution_amd.models import basic_wdsr_b as BW
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks

    ns = argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=8, num_residual_units=24,
                            hot_dtype=args.dtype, hot_grad_segments=2)
    if args.mode == "fused":
        return fused_mode(args, ns, rank, world)
    torch.manual_seed(0)
    m = get_model(ns).cuda().train()
    events = []
    lo_bwd = BW._NetLoFunction.backward

    def lo_bwd_logged(ctx, g):
        events.append(["lo_backward"])
        return lo_bwd(ctx, g)
    BW._NetLoFunction.backward = staticmethod(lo_bwd_logged)
    # pretrain.py:239 wraps with the defaults; the two ~0.4 MB segments fall into ONE default bucket (first-bucket cap 1 MB),
    # so the harness sizes the buckets to the segments: that is the only DDP argument the hot path asks for.  DDP closes a
    # bucket once it has REACHED the cap, so the cap must not exceed the smaller segment; and with find_unused_parameters
    # off the first iteration always runs one bucket, the split takes effect when DDP rebuilds its buckets after it.
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], output_device=0, bucket_cap_mb=m.ddp_bucket_cap_mb(),
                                                    gradient_as_bucket_view=True, broadcast_buffers=False)

    def hook(state, bucket):
        events.append(["bucket", bucket.index(), bucket.buffer().numel()])
        return default_hooks.allreduce_hook(state, bucket)
    ddp.register_comm_hook(None, hook)
    g = torch.Generator().manual_seed(123)
    n = 4 * world
    x = torch.rand(n, 3, 24, 36, generator=g)
    hr = torch.rand(n, 3, 96, 144, generator=g)
    per = n // world
    xs, hs = x[rank * per:(rank + 1) * per].cuda(), hr[rank * per:(rank + 1) * per].cuda()
    for it in range(3):                                       # from the second iteration on DDP has rebuilt its buckets
        for p in m.parameters():
            p.grad = None
        events.clear()
        torch.nn.functional.l1_loss(ddp(xs), hs).backward()
        torch.cuda.synchronize()
    got = torch.cat([m.flat_lo.grad, m.flat_hi.grad]).cpu()
    if rank == 0:
        # the same global batch on ONE process, one-parameter model, no DDP
        ns.hot_grad_segments = 1
        torch.manual_seed(0)
        ref = get_model(ns).cuda().train()
        torch.nn.functional.l1_loss(ref(x.cuda()), hr.cuda()).backward()
        rg = ref.flat.grad.cpu()
        err = (got - rg).abs().max().item() / rg.abs().max().item()
        json.dump({"rel_err": err, "events": events, "n_params": [m.flat_lo.numel(), m.flat_hi.numel()]}, open(args.out, "w"))
    dist.barrier()
    dist.destroy_process_group()


def _single(ref, x, hr, rs):
    """the reference side steps WITHOUT the process group (whole batch on one rank)"""
    import torch.distributed as d
    real = d.is_initialized
    d.is_initialized = lambda: False                           # train_step then takes the one-rank route
    try:
        return ref.train_step(x.cuda(), hr.cuda(), rs).item()
    finally:
        d.is_initialized = real


def fused_mode(args, ns, rank, world):
    """model.train_step(..., process_group): every rank steps on its shard; afterwards the replicas are equal and match a
    single process stepping on the whole batch"""
    from mobilesuperresolution_amd.models import get_model
    ns.hot_grad_segments = 1
    torch.manual_seed(0)
    m = get_model(ns).cuda().train()
    st = m.make_train_state(lr=1e-3 * world)                   # pretrain.py:216
    g = torch.Generator().manual_seed(321)
    n = 4 * world
    per = n // world
    losses = []
    batches = [(torch.rand(n, 3, 24, 36, generator=g), torch.rand(n, 3, 96, 144, generator=g)) for _ in range(3)]
    for x, hr in batches:
        losses.append(m.train_step(x[rank * per:(rank + 1) * per].cuda(), hr[rank * per:(rank + 1) * per].cuda(), st).item())
    mine = m.flat.detach().cpu()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if rank == 0:
        torch.manual_seed(0)
        ref = get_model(ns).cuda().train()
        rs = ref.make_train_state(lr=1e-3 * world)
        ref_losses = [_single(ref, x, hr, rs) for x, hr in batches]
        rf = ref.flat.detach().cpu()
        json.dump({"replicas_equal": all(torch.equal(gathered[0], t) for t in gathered[1:]),
                   "param_err": (mine - rf).abs().max().item(), "param_scale": rf.abs().max().item(),
                   "loss_rank0": losses, "loss_full": ref_losses}, open(args.out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
