"""Diagnostic: back-to-back launch time of the block kernels (bf16) at several batch sizes plus the two
whole-network calls and the full training step at C2, all in ONE process (HIP events on the launch stream).
    python tools/kbench.py [--f 24] [--batches 32,512]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from mobilesuperresolution_amd import _lib as L
from mobilesuperresolution_amd.models import get_model

ap = argparse.ArgumentParser()
ap.add_argument("--f", type=int, default=24)
ap.add_argument("--batches", default="32,512")
ap.add_argument("--reps", type=int, default=64)
ap.add_argument("--step", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda", 0)
F = args.f
ns = B.model_ns("bf16")
ns.num_residual_units = F
m = get_model(ns).to(dev).train()
st = m._state(dev)
x0 = torch.rand(2, 3, 48, 48, device=dev)
m(x0)                                            # packs the weights
lib = L.lib()


def ev_time(fn, n=7, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


out = {"F": F}
for batch in [int(b) for b in args.batches.split(",")]:
    a = torch.randn(batch, 48, 48, F, device=dev).bfloat16()
    b1 = torch.empty_like(a); b2 = torch.empty_like(a)
    reps = args.reps
    alg = 2 * batch * 48 * 48 * F * 2
    f1 = lambda: L.check(lib.sr_wdsr_block_fwd_repeat(a.data_ptr(), b1.data_ptr(), st.blob_body[0].data_ptr(),
                         st.cinit_body[0].data_ptr(), batch, 48, 48, F, 1, reps, L.stream_ptr()), "r1")
    u1 = ev_time(f1) / reps
    row = {"fwd1_us": round(u1 * 1e6, 2), "fwd1_GBs": round(alg / u1 / 1e9), "fwd1_frac": round(alg / u1 / 8e12, 4)}
    if F == 24:
        f2 = lambda: L.check(lib.sr_wdsr_block2_fwd_repeat(a.data_ptr(), b1.data_ptr(), b2.data_ptr(),
                             st.blob_body[0].data_ptr(), st.blob_body[1].data_ptr(), st.cinit_body[0].data_ptr(),
                             st.cinit_body[1].data_ptr(), batch, 48, 48, F, 1, reps, L.stream_ptr()), "r2")
        u2 = ev_time(f2) / reps
        row.update({"fwd2_us": round(u2 * 1e6, 2), "fwd2_GBs": round(2 * alg / u2 / 1e9), "fwd2_frac": round(2 * alg / u2 / 8e12, 4)})
    for nblk in ((1, 2) if F == 24 else (1,)):
        fr = lambda: L.check(lib.sr_wdsr_fwd_rs_repeat(a.data_ptr(), b1.data_ptr(), b2.data_ptr(), st.blob_body[0].data_ptr(),
                             st.blob_body[1].data_ptr(), st.cinit_body[0].data_ptr(), st.cinit_body[1].data_ptr(), None, None, nblk, batch, 48, 48,
                             F, 1, reps, L.stream_ptr()), "rs")
        u = ev_time(fr) / reps
        row.update({f"rs{nblk}_us": round(u * 1e6, 2), f"rs{nblk}_frac": round(nblk * alg / u / 8e12, 4)})
    out[f"batch{batch}"] = row
    del a, b1, b2

if args.step:
    x = torch.rand(32, 3, 48, 48, device=dev); hr = torch.rand(32, 3, 192, 192, device=dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
    timer = L.KernelTimer(); 
    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.l1_loss(m(x), hr).backward()
        opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50
    L.set_timer(timer)
    for _ in range(20): step()
    L.set_timer(None)
    out["step_ms"] = round(t * 1e3, 4)
    st = m.make_train_state(lr=1e-3)
    for _ in range(5): m.train_step(x, hr, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): m.train_step(x, hr, st)
    torch.cuda.synchronize(); out["fused_step_ms"] = round((time.perf_counter() - t0) / 50 * 1e3, 4)
    t0 = time.perf_counter()
    for _ in range(50): m.train_step(x, hr, st).item()
    out["fused_step_with_item_ms"] = round((time.perf_counter() - t0) / 50 * 1e3, 4)
    out["calls_us"] = {k: round(v[1] * 1e3, 1) for k, v in timer.summary().items()}
    with torch.no_grad():
        m.eval()
        for _ in range(5): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): m(x)
        torch.cuda.synchronize(); out["infer_ms"] = round((time.perf_counter() - t0) / 50 * 1e3, 4)
print(json.dumps(out))
