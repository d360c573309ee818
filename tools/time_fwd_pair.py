"""A/B timing of the two-block forward launch (sr_wdsr_fwd_rs_repeat) at several batches, for one or more library builds in
ONE process (interleaved rounds, rule 24).   python tools/time_fwd_pair.py lib1.so [lib2.so ...] -- batch [batch ...]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
libs = args[:args.index("--")] if "--" in args else [None]
batches = [int(a) for a in (args[args.index("--") + 1:] if "--" in args else args)] or [32, 512]
from mobilesuperresolution_amd import _lib as L, hotpath as HP
f = 24
dev = torch.device("cuda", 0)
src = torch.randn(2, HP.tables(f, dev)["src_size"], device="cuda") * 0.1
blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
handles = []
for p in libs:
    if p is None:
        handles.append(("default", L.lib()))
    else:
        h = ctypes.CDLL(os.path.abspath(p))
        L._declare(h)
        handles.append((os.path.basename(p), h))
for n in batches:
    x = torch.randn(n, 48, 48, f, device="cuda").bfloat16()
    ya, yb = torch.empty_like(x), torch.empty_like(x)
    tsv = torch.empty((2, n, 8, 288, 24), device='cuda', dtype=torch.bfloat16) if os.environ.get('SAVE_T') else None
    reps = 320 if n <= 64 else 32
    res = {name: [] for name, _ in handles}
    for rnd in range(6):
        for name, h in handles:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = h.sr_wdsr_fwd_rs_repeat(x.data_ptr(), None if os.environ.get('NOYA') else ya.data_ptr(), yb.data_ptr(), blob[0].data_ptr(), blob[1].data_ptr(), cinit[0].data_ptr(),
                                         cinit[1].data_ptr(), tsv[0].data_ptr() if tsv is not None else None,
                                         tsv[1].data_ptr() if tsv is not None else None, 2, n, 48, 48, f, 1, reps, L.stream_ptr())
            assert rc == 0, rc
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[name].append(e0.elapsed_time(e1) * 1e3 / reps)
    alg = 2 * 2 * n * 48 * 48 * f * 2
    for name, ts in res.items():
        ts = sorted(ts)
        med = ts[len(ts) // 2]
        print(f"batch {n:4d} {name:28s} median {med:8.2f} us  min {ts[0]:8.2f}  -> {alg / med / 1e3:7.1f} GB/s = {alg / med / 1e3 / 8000:.4f} of the HBM roof")
