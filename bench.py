#!/usr/bin/env python3
"""Headline benchmark of the MI355X SR hot path: HR megapixels/s of a WDSR-B x4 training step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic 48x48 LR patches already resident
in HBM: forward (head, 16 fused residual blocks, fused tail+skip+PixelShuffle), L1 loss, backward,
Adam step -- what pretrain.py's train() does per batch (reference pretrain.py:61-80).  Workload at
every N: BASELINE.json configs[1] (x4, 16 blocks / 24 units, bf16 storage, batch 32 per GPU, weak
scaling).  The timed step is `model.train_step` (loss folded into the tail backward, Adam kernel): one C call at
N = 1; at N > 1 forward + backward, ONE RCCL all-reduce (average) of the 0.77 MB flat gradient on the compute stream,
Adam kernel (`--overlap`: the all-reduce in two halves, the first under the early half of the backward; both are
timed at N > 1 and the other one is printed as `alt_route_ms_per_step`, so the scaling run decides the default;
`--ddp-wrapper`: torch's DistributedDataParallel as pretrain.py:239 wraps it).  The line also carries the reference's own
loop on the drop-in route (`reference_surface_ms_per_step`: pretrain.py:69-82 with training.L1Loss / training.Adam and
`loss.item()` every step), the plain nn.Module + torch loss + torch Adam route (`unfused_ms_per_step`), the fused step with
the per-step `loss.item()` (`ms_per_step_with_item_sync`), and the per-step distribution (median, p10, p90, HIP events).

Rank 0 prints ONE JSON line (metric, value, ..., roofline, cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HR_MPIX_PER_PATCH = (48 * 4) ** 2 / 1e6          # 0.036864
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BATCH, LR, SCALE, BLOCKS, UNITS = 32, 48, 4, 16, 24


def model_ns(dtype="bf16", segments=1):
    return argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=SCALE,
                              num_blocks=BLOCKS, num_residual_units=UNITS, hot_dtype=dtype, hot_grad_segments=segments)


def algorithmic_bytes(n, h, w, f, nb, r, s):
    """Algorithmic HBM bytes per launch: minimal tensor reads + writes, weights excluded."""
    px = n * h * w
    return {
        "sr_wdsr_block_fwd": 2 * px * f * s,                         # read x, write y
        "sr_wdsr_block_bwd_data": 3 * px * f * s,                    # read x, dy; write dx
        "sr_wdsr_block_wgrad": nb * 2 * px * f * s,                  # read x, dy of every block (one call)
        "sr_head_fwd": px * (3 * 4 + f * s),
        "sr_tail_fwd": px * (f * s + 3 * 4 + 3 * r * r * 4),
        "sr_tail_bwd_data": px * (3 * r * r * 4 + f * s),
        "sr_tail_wgrad": px * (3 * r * r * 4 + f * s + 3 * 4),
        "sr_head_wgrad": px * (f * s + 3 * 4),
    }


def cpu_baseline(budget_s=20.0):
    """The oracle (plain PyTorch fp32 restatement of the reference, parity-pinned) timed on this box's
    host cores on the same workload shape: full train step, batch 32, 16 blocks / 24 units.  The thread
    count is calibrated first (one step each at 8/16/32 threads; small convs oversubscribe badly beyond that)
    and the fastest is used for the sample."""
    from oracle.wdsr_oracle import OracleBasicModel
    torch.manual_seed(0)
    m = OracleBasicModel(model_ns()).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    x = torch.rand(BATCH, 3, LR, LR)
    hr = torch.rand(BATCH, 3, LR * SCALE, LR * SCALE)

    def step():
        opt.zero_grad()
        loss = torch.nn.functional.l1_loss(m(x), hr)
        loss.backward()
        opt.step()

    ncpu = os.cpu_count() or 1
    best_t, best_n = None, None
    cands = sorted({min(n, ncpu) for n in (8, 16, 32)})          # more threads were never faster (oversubscription)
    torch.set_num_threads(cands[0])
    step()
    for nt in cands:
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, best_n = dt, nt
    torch.set_num_threads(best_n)
    t0, n = time.perf_counter(), 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 30:
            break
    out = {"value": round(BATCH * HR_MPIX_PER_PATCH * n / el, 4), "unit": "HR-Mpix/s",
           "cores": best_n, "kind": "port",
           "sample": f"{n} full train steps (fwd+L1+bwd+Adam) of the same workload, batch {BATCH}, fp32, "
                     f"{el:.1f} s on {best_n} threads (fastest of 8/16/32 threads on a {ncpu}-CPU host)"}
    # BASELINE config 1 (the reference's own CPU-runnable case): x4, 4 blocks / 24 units, ONE 48x48 patch, fp32
    ns1 = model_ns()
    ns1.num_blocks = 4
    torch.manual_seed(0)
    m1 = OracleBasicModel(ns1).train()
    opt1 = torch.optim.Adam(m1.parameters(), lr=1e-3)
    x1, hr1 = torch.rand(1, 3, LR, LR), torch.rand(1, 3, LR * SCALE, LR * SCALE)

    def step1():
        opt1.zero_grad()
        torch.nn.functional.l1_loss(m1(x1), hr1).backward()
        opt1.step()

    def rate(fn, budget):
        fn()
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget and k < 400:
            fn()
            k += 1
        return k, time.perf_counter() - t0
    best1 = None
    for nt in sorted({min(v, ncpu) for v in (1, 4, 8)}):      # a single patch does not feed many threads
        torch.set_num_threads(nt)
        k, el1 = rate(step1, 1.0)
        if best1 is None or k / el1 > best1[0]:
            best1 = (k / el1, nt)
    torch.set_num_threads(best1[1])
    k, el1 = rate(step1, 3.0)
    m1.eval()
    with torch.no_grad():
        kf, elf = rate(lambda: m1(x1), 2.0)
    out["c1"] = {"workload": "BASELINE config 1: x4, 4 blocks / 24 units, batch 1, 48x48, fp32", "cores": best1[1],
                 "train_step_HR_Mpix_s": round(HR_MPIX_PER_PATCH * k / el1, 4), "forward_HR_Mpix_s": round(HR_MPIX_PER_PATCH * kf / elf, 4),
                 "sample": f"{k} train steps in {el1:.1f} s, {kf} forwards in {elf:.1f} s"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ddp-wrapper", action="store_true",
                    help="N > 1: train through torch's DistributedDataParallel wrapper instead of model.train_step's own all-reduce")
    ap.add_argument("--overlap", action="store_true",
                    help="N > 1, fused route: all-reduce the gradient in two halves, the first under the early half's backward "
                         "(default for this model size: one all-reduce after the backward, which is faster -- DESIGN.md section 6)")
    ap.add_argument("--force-ddp", action="store_true",
                    help="wrap in DistributedDataParallel even with one rank (measures the DDP/RCCL overhead on one GPU)")
    args = ap.parse_args()
    # ONE JSON line on stdout: RCCL prints a version banner to the process's stdout (file descriptor 1) when its first
    # communicator comes up, so fd 1 is pointed at stderr for the run and the line goes to a saved copy of the real stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus} (one rank per GPU)")
    if os.environ.get("SR_BENCH_SHARED_GPU") == "1":    # rehearsal of the N > 1 control flow on a one-GPU box (with gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_ddp = world > 1 or args.force_ddp
    if use_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend=os.environ.get("SR_BENCH_BACKEND", "nccl"), init_method="env://")   # nccl == RCCL

    from mobilesuperresolution_amd import hotpath as HP
    from mobilesuperresolution_amd.models import get_model

    torch.manual_seed(0)                                # identical replicas (DDP broadcasts rank 0 anyway)
    # Data parallel.  Default: `model.train_step(..., process_group)` -- the fused step with the gradient all-reduce (RCCL,
    # average) in two halves, the first running under the early half of the backward.  --ddp-wrapper: the reference's own
    # DistributedDataParallel wrapper (pretrain.py:239) over the two-segment parameter mode (two autograd nodes, one
    # bucket each: the same overlap through DDP's hooks), with torch's loss and optimizer.
    wrapper = use_ddp and args.ddp_wrapper
    model = get_model(model_ns(args.dtype, 2 if wrapper else 1)).to(dev).train()
    net = model
    if wrapper:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
                                                        gradient_as_bucket_view=True, broadcast_buffers=False,
                                                        bucket_cap_mb=model.ddp_bucket_cap_mb())
    elif use_ddp:
        dist.broadcast(model.flat.data, src=0)              # identical replicas, as DDP's constructor guarantees
    lr_rate = 1e-3 * world                              # pretrain.py:216 linear scaling
    try:
        opt = torch.optim.Adam(model.parameters(), lr=lr_rate, fused=True)
    except Exception:
        opt = torch.optim.Adam(model.parameters(), lr=lr_rate, foreach=True)
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)   # each rank its own shard of synthetic patches
    x = torch.rand(BATCH, 3, LR, LR, generator=g).to(dev)
    hr = torch.rand(BATCH, 3, LR * SCALE, LR * SCALE, generator=g).to(dev)

    def step_unfused():
        opt.zero_grad(set_to_none=True)
        sr = net(x)
        loss = torch.nn.functional.l1_loss(sr, hr)
        loss.backward()
        opt.step()
        return loss

    # the reference's loop UNCHANGED (pretrain.py:69-82) with its two objects swapped for the library's drop-ins
    # (mobilesuperresolution_amd.training.L1Loss / .Adam): forward, criterion, backward, optimizer.step, loss.item()
    from mobilesuperresolution_amd import training as T
    crit_ref = T.L1Loss()
    opt_ref = T.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr_rate)

    def step_reference_surface():
        opt_ref.zero_grad()
        sr = net(x)
        loss = 0
        loss_sr_l1 = 1.0 * crit_ref(sr, hr)
        loss += loss_sr_l1
        loss.backward()
        opt_ref.step()
        opt_ref.zero_grad()
        return loss.item()

    fused = not wrapper                                  # the library's own step (one rank: one call; N ranks: + 2 all-reduces)
    state = model.make_train_state(lr=lr_rate) if fused else None

    pg = dist.group.WORLD if (use_ddp and fused) else None   # --force-ddp on one rank: the data-parallel route all the same

    def step():
        return model.train_step(x, hr, state, process_group=pg, overlap=(True if args.overlap else None)) if fused else step_unfused()

    def sync():
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    # clock settle: the driver's default (--warmup 5 --steps 20) is 9 ms of GPU work right after process start, shorter than the
    # GPU's clock ramp; ~40 ms of untimed forward passes (inference, nothing saved) come first so that the warm-up and timed steps
    # run at the clocks a training run sees.  Reported as `clock_settle_forwards`; SR_BENCH_SETTLE=0 turns it off.
    settle = int(os.environ.get("SR_BENCH_SETTLE", 400))
    with torch.no_grad():
        for _ in range(settle):
            model(x)
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(loss.detach())
    alt_ms = None
    if use_ddp and fused and model.nb_split:
        # the other data-parallel route (overlapped halves <-> one all-reduce), same steps: the scaling run decides the default
        alt = not bool(args.overlap)

        def step_alt():
            return model.train_step(x, hr, state, process_group=pg, overlap=alt)
        for _ in range(max(args.warmup // 2, 2)):
            step_alt()
        sync()
        ta = time.perf_counter()
        for _ in range(args.steps):
            step_alt()
        sync()
        alt_el = time.perf_counter() - ta
        if world > 1:
            t = torch.tensor([alt_el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            alt_el = t.item()
        alt_ms = alt_el / args.steps * 1e3

    # the same step, untimed for `value`: per-step distribution (HIP events on the launch stream, no host sync inside the
    # loop), with the per-step loss.item() sync of pretrain.py:82, and through the plain nn.Module / torch.optim route
    def per_step_ms(fn, n):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))

    def wall_ms(fn, n):
        sync()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        return (time.perf_counter() - t) / n * 1e3
    nd = max(args.steps, 50)
    dist_ms = per_step_ms(step, nd)
    item_ms = wall_ms(lambda: step().item(), nd)
    unfused_ms = ref_surface_ms = None
    if fused and not use_ddp:
        for _ in range(5):                               # (torch's optimizer initialises its state on first use)
            step_unfused()
        unfused_ms = wall_ms(step_unfused, nd)
        for _ in range(5):
            step_reference_surface()
        ref_surface_ms = wall_ms(step_reference_surface, nd)

    # forward-only (inference) throughput, same batch
    model.eval()
    with torch.no_grad():
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
        tf = time.perf_counter()
        for _ in range(args.steps):
            model(x)
        torch.cuda.synchronize()
        fwd_elapsed = time.perf_counter() - tf
    model.train()

    # whole forward / backward calls, for the record: EVERY rank runs these steps (under DDP they contain the
    # gradient all-reduce, a collective), only rank 0 keeps the timings
    from mobilesuperresolution_amd import _lib as L
    timer = L.KernelTimer()
    if rank == 0:
        L.set_timer(timer)
    for _ in range(5):
        step_unfused() if not (use_ddp and fused) else step()
    L.set_timer(None)
    sync()

    if rank == 0:
        # ---- roofline leg: the graded kernel (fused residual block forward), launched back to back from
        # C on torch's stream and bracketed by HIP events on that stream ----
        from mobilesuperresolution_amd import _lib as L
        s = 2 if args.dtype == "bf16" else 4
        alg = algorithmic_bytes(BATCH, LR, LR, UNITS, BLOCKS, SCALE, s)
        st = model._state(dev)
        tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        a = torch.randn(BATCH, LR, LR, UNITS, device=dev).to(tdt)
        b, c = torch.empty_like(a), torch.empty_like(a)
        reps = 320
        blob, cin = st.blob_body, st.cinit_body
        pair = args.dtype == "bf16" and UNITS == 24          # the two-block kernel is what the step launches

        def rs_chain(nblk, src, d1, d2, batch, ts=None):
            L.launch("sr_wdsr_fwd_rs_repeat", L.lib().sr_wdsr_fwd_rs_repeat, src.data_ptr(), d1.data_ptr(), d2.data_ptr(),
                     blob[BLOCKS - 2].data_ptr(), blob[BLOCKS - 1].data_ptr(), cin[BLOCKS - 2].data_ptr(),
                     cin[BLOCKS - 1].data_ptr(), ts[0].data_ptr() if ts is not None else None,
                     ts[1].data_ptr() if ts is not None and nblk == 2 else None, nblk, batch, LR, LR, UNITS, L.DTYPE_CODE[tdt], reps,
                     L.stream_ptr())

        # the variant the TIMED training step launches also keeps the two t images for the weight-gradient kernels (SAVE_T):
        # that one is quoted as `frac`; the inference variant (nothing saved) beside it
        tiles = (LR // 12) * (LR // 24)
        ts32 = torch.empty((2, BATCH, tiles, 288, 24), device=dev, dtype=torch.bfloat16) if pair else None

        def chain1():
            if pair:
                return rs_chain(1, a, b, c, BATCH)
            L.launch("sr_wdsr_block_fwd_repeat", L.lib().sr_wdsr_block_fwd_repeat, a.data_ptr(), b.data_ptr(),
                     blob[BLOCKS - 1].data_ptr(), cin[BLOCKS - 1].data_ptr(), BATCH, LR, LR, UNITS,
                     L.DTYPE_CODE[tdt], reps, L.stream_ptr())

        def chain2():
            rs_chain(2, a, b, c, BATCH, ts32)

        def chain2_inf():
            rs_chain(2, a, b, c, BATCH)

        def timed(chain, repeats=5):
            """median over `repeats` chains of `reps` back-to-back launches each (a first chain warms caches and clocks)"""
            chain()
            torch.cuda.synchronize()
            ts = []
            for _ in range(repeats):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()                                   # torch's current stream == the launch stream
                chain()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / reps)
            return sorted(ts)[len(ts) // 2]
        us1 = timed(chain1)
        us = timed(chain2) if pair else us1
        us_inf = timed(chain2_inf) if pair else None
        units = 2 if pair else 1                              # residual blocks per launch
        alg_launch = units * alg["sr_wdsr_block_fwd"]
        achieved = alg_launch / (us * 1e-6) / 1e9
        calls = {k: round(v[1] * 1e3, 1) for k, v in timer.summary().items()}
        traffic = None                                   # PMC bytes per launch, collected offline with rocprofv3 --pmc
        pmc = os.path.join(ROOT, "profiles", ("r03_pmc_fwd_rs2.json" if os.path.exists(os.path.join(ROOT, "profiles", "r03_pmc_fwd_rs2.json"))
                                              else "r02_pmc_fwd_rs2.json") if pair else "r01_pmc_block_fwd.json")
        if args.dtype == "bf16" and os.path.exists(pmc):
            traffic = json.load(open(pmc))["traffic_bytes_per_launch"]
        kname = "wdsr_fwd_rs_kernel<24,144,20,2,true>" if pair else f"wdsr_block_fwd_kernel<{args.dtype},24,144,20>"
        roofline = {"kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "alg_bytes_per_launch": alg_launch, "blocks_per_launch": units, "avg_launch_us": round(us, 2),
                    "single_block_kernel": {"avg_launch_us": round(us1, 2),
                                            "achieved": round(alg["sr_wdsr_block_fwd"] / (us1 * 1e-6) / 1e9, 1)},
                    "how": f"median of 5 chains of {reps} back-to-back launches from one C call each, HIP events on the launch stream "
                           "(includes inter-launch gaps); algorithmic bytes = SURVEY 8(d) per-block figure "
                           "(read x + write y) x blocks per launch; the kernel is the SAVE_T variant the timed training step launches "
                           "(it also writes the two t images, 7 MB, which the algorithmic figure does not count); the three 3.5 MB "
                           "activation buffers the chain ping-pongs stay in L2 / Infinity Cache"}
        if pair:
            roofline["inference_variant"] = {"kernel": "wdsr_fwd_rs_kernel<24,144,20,2,false>", "avg_launch_us": round(us_inf, 2),
                                             "frac": round(alg_launch / (us_inf * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        if pair:
            # What bounds the kernel AS BUILT is the matrix pipe, not HBM: per 12x24 tile a workgroup issues 19 MFMAs
            # (conv1: 5 e-tiles x 2 k-steps, conv2: 9 k-steps) per 32-pixel tile of t and 12 per 32-pixel tile of the
            # dense-K 3x3, over the halo'd regions 28x16 -> 26x14 -> 24x12 (14 / 12 / 12 / 9 pixel tiles): padding (24->32,
            # 144->160, 20->32 channels) and halo recompute included.  32x32x16 bf16 = 32 768 flop, 32 cycles on one of
            # the CU's four matrix pipes at 2.4 GHz.  `frac` stays the HBM figure the target (0.60) is stated in.
            mfma_per_wg = 19 * (14 + 12) + 12 * (12 + 9)        # dense-K 3x3: 12 k-steps (15 before round 3)
            wgs = BATCH * (LR // 12) * (LR // 24)
            rounds = -(-wgs // 256)                           # one 133 KB-LDS workgroup per CU at a time
            floor_us = rounds * mfma_per_wg / 4 * 32 / 2.4e3
            issued_tflops = wgs * mfma_per_wg * 32768 / (us * 1e-6) / 1e12
            roofline.update({"bound": "mfma", "hbm_frac": roofline["frac"], "mfma_floor_us": round(floor_us, 2),
                             "mfma_frac": round(floor_us / us, 4), "mfma_issued_tflops": round(issued_tflops, 1),
                             "mfma_peak_tflops": 2500.0, "mfma_per_workgroup": mfma_per_wg,
                             "flop_per_alg_byte": round(wgs * mfma_per_wg * 32768 / alg_launch, 1)})
            big = 512                                         # the same kernel with 16 workgroups per CU
            a5 = torch.randn(big, LR, LR, UNITS, device=dev).to(tdt)
            b5, c5 = torch.empty_like(a5), torch.empty_like(a5)
            keep, reps = reps, 64
            for _ in range(4):                                # 64 launches of ~0.06 ms per chain: let the clocks settle on this grid
                rs_chain(2, a5, b5, c5, big)
            us5 = timed(lambda: rs_chain(2, a5, b5, c5, big))
            ts5 = torch.empty((2, big, tiles, 288, 24), device=dev, dtype=torch.bfloat16)
            us5s = timed(lambda: rs_chain(2, a5, b5, c5, big, ts5))
            reps = keep
            alg5 = units * algorithmic_bytes(big, LR, LR, UNITS, BLOCKS, SCALE, s)["sr_wdsr_block_fwd"]
            # one whole image per workgroup: the streaming kernel (csrc/wdsr_fwd_stream.h), 4 464 MFMAs per image and two blocks
            floor5 = 4464 * (big / 256) / 4 * 32 / 2.4e3
            roofline["batch512"] = {"kernel": "wdsr_fwd_stream12_kernel<24,144,20,false>", "avg_launch_us": round(us5, 2),
                                    "achieved": round(alg5 / (us5 * 1e-6) / 1e9, 1),
                                    "frac": round(alg5 / (us5 * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "mfma_floor_us": round(floor5, 2),
                                    "mfma_frac": round(floor5 / us5, 4),
                                    "training_variant": {"kernel": "wdsr_fwd_stream12_kernel<24,144,20,true>", "avg_launch_us": round(us5s, 2),
                                                         "frac": round(alg5 / (us5s * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}}
            del a5, b5, c5, ts5
        else:
            roofline["bound"] = "hbm"
        kernels = {"call_us": calls}
        out = {
            "metric": "HR megapixels/sec (WDSR-B x4, 48x48 LR patches), full training step",
            "value": round(world * BATCH * HR_MPIX_PER_PATCH * args.steps / elapsed, 2),
            "unit": "HR-Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "step_route": ("model.train_step: loss folded into the tail backward + Adam kernel" +
                           ((", gradient all-reduce (avg) in two halves, the first under the early half of the backward" if args.overlap
                             else ", one gradient all-reduce (avg, 0.77 MB) on the compute stream before the Adam kernel") if use_ddp else ", one C call"))
                          if fused else "DistributedDataParallel wrapper (two gradient segments): forward / F.l1_loss / backward + "
                                        "bucketed all-reduce / torch Adam",
            "clock_settle_forwards": settle,
            "per_step_ms": {"n": nd, "median": round(dist_ms[nd // 2], 4), "p10": round(dist_ms[nd // 10], 4),
                            "p90": round(dist_ms[(nd * 9) // 10], 4)},
            "ms_per_step_with_item_sync": round(item_ms, 4),
            "alt_route_ms_per_step": None if alt_ms is None else round(alt_ms, 4),
            "alt_route": None if alt_ms is None else ("one all-reduce after the backward" if args.overlap else
                                                      "all-reduce in two halves, the first under the early half of the backward"),
            "unfused_ms_per_step": None if unfused_ms is None else round(unfused_ms, 4),
            "reference_surface_ms_per_step": None if ref_surface_ms is None else round(ref_surface_ms, 4),
            "reference_surface": "pretrain.py:69-82 verbatim (zero_grad / model(lr) / criterion / backward / optimizer.step / "
                                 "loss.item() every step) with nn.L1Loss -> training.L1Loss and optim.Adam -> training.Adam",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"WDSR-B x{SCALE}, {BLOCKS} blocks / {UNITS} units, {LR}x{LR} LR patches, "
                                   f"batch {BATCH} per GPU, fwd+L1+bwd+Adam" + (f", DDP/{'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend()}" if use_ddp else ""),
                       "global_batch": BATCH * world, "parallelism": f"dp{world}"},
            "forward_only_HR_Mpix_s": round(BATCH * HR_MPIX_PER_PATCH * args.steps / fwd_elapsed, 2),
            "final_loss": round(final_loss, 5),
            "roofline": roofline,
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), file=real_stdout, flush=True)
    if use_ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
