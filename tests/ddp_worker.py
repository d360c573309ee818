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
    ap.add_argument("--mode", default="wrapper", choices=["wrapper", "fused", "fused_overlap", "nas_phases"])
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(os.environ.get("SR_DDP_BACKEND", "gloo"), init_method="env://")
    torch.cuda.set_device(0)                                  # one-GPU box: every rank on cuda:0
    from mobilesuperresolution_amd.models import get_model
    from mobilesuperresolution_amd.models import basic_wdsr_b as BW
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks

    ns = argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=8, num_residual_units=24,
                            hot_dtype=args.dtype, hot_grad_segments=2)
    if args.mode in ("fused", "fused_overlap"):
        return fused_mode(args, ns, rank, world, overlap=args.mode == "fused_overlap")
    if args.mode == "nas_phases":
        return nas_phases_mode(args, rank, world)
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


def fused_mode(args, ns, rank, world, overlap=False):
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
        losses.append(m.train_step(x[rank * per:(rank + 1) * per].cuda(), hr[rank * per:(rank + 1) * per].cuda(), st,
                                   overlap=overlap).item())
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
                   "param_bad_frac": ((mine - rf).abs() > 2e-5 * rf.abs().max() + 2e-6).float().mean().item(), "lr": 1e-3 * world,
                   "loss_rank0": losses, "loss_full": ref_losses}, open(args.out, "w"))
    dist.barrier()
    dist.destroy_process_group()


def nas_phases_mode(args, rank, world):
    """The three phases of search.py:290-405 on the HIP NAS_MODEL under DistributedDataParallel: width search
    (length_grad(False)), length search (length_grad(True), mask_grad(True)), kernel training (both off), with the model
    unwrapped (`.module`) and wrapped again at each change (search.py:329-333,372-376)."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    from mobilesuperresolution_amd.models import get_model, wrap_ddp
    ns = argparse.Namespace(model_type="NAS_MODEL", image_mean=0.5, num_channels=3, scale=2, num_blocks=4, num_residual_units=24,
                            hot_dtype=args.dtype, width_search=True, length_search=True, pretrained=False)
    g = torch.Generator().manual_seed(77)
    n = 2 * world
    x, hr = torch.rand(n, 3, 24, 36, generator=g), torch.rand(n, 3, 48, 72, generator=g)
    per = n // world
    xs, hs = x[rank * per:(rank + 1) * per].cuda(), hr[rank * per:(rank + 1) * per].cuda()

    def iterate(ddp, opt, its=3):
        for _ in range(its):
            opt.zero_grad(set_to_none=True)
            out, speed = ddp(xs)
            (torch.nn.functional.l1_loss(out, hs) + 1e-3 * speed.sum()).backward()
            opt.step()
        torch.cuda.synchronize()

    def snapshot(m):
        return {k: v.detach().clone() for k, v in m.named_reference_tensors()}

    def changed(m, before):
        return sorted(k for k, v in m.named_reference_tensors() if not torch.equal(v.detach(), before[k]))

    def replicas_equal(m):
        mine = torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).cpu()
        got = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        return all(torch.equal(got[0], t) for t in got[1:])

    res = {}
    # the reference-shaped model (one Parameter per tensor) breaks stock DDP on its second iteration; with the flat body
    # parameter there is no gradient-less Parameter left and stock wrapping runs
    torch.manual_seed(0)
    model = get_model(ns).cuda().train()
    model.length_grad(False)
    stock = DDP(model, device_ids=[0], output_device=0)
    try:
        iterate(stock, torch.optim.Adam(stock.parameters(), 1e-3), its=2)
        res["stock_ddp_error"] = None
    except RuntimeError as e:
        res["stock_ddp_error"] = str(e)[:160]
    del stock
    dist.barrier()

    torch.manual_seed(0)
    model = get_model(ns).cuda().train()
    # phase 1: width-only search (search.py:290-327)
    model.length_grad(False)
    ddp = wrap_ddp(model, device_ids=[0], output_device=0)
    before = snapshot(model)
    iterate(ddp, torch.optim.Adam(ddp.parameters(), 1e-3 * 10 / world))
    res["phase1_changed"], res["phase1_equal"] = changed(model, before), replicas_equal(model)
    # phase 2: length search (search.py:329-368)
    model = ddp.module
    model.length_grad(True)
    model.mask_grad(True)
    ddp = wrap_ddp(model, device_ids=[0], output_device=0)
    res["phase2_frozen_kinds"] = sorted(model._frozen)
    before = snapshot(model)
    iterate(ddp, torch.optim.Adam(ddp.parameters(), 1e-3))
    res["phase2_changed"], res["phase2_equal"] = changed(model, before), replicas_equal(model)
    # phase 3: kernel training (search.py:370-405)
    model = ddp.module
    model.length_grad(False)
    model.mask_grad(False)
    ddp = wrap_ddp(model, device_ids=[0], output_device=0)
    before = snapshot(model)
    iterate(ddp, torch.optim.Adam(ddp.parameters(), 1e-3))
    res["phase3_changed"], res["phase3_equal"] = changed(model, before), replicas_equal(model)
    if rank == 0:
        json.dump(res, open(args.out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
