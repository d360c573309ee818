"""Diagnostic: per-phase timeline of the block weight-gradient kernels from in-kernel stamps."""
import os, sys
os.environ["SR_HOTPATH_DEBUG_LIB"] = "1"          # the diagnostic build (python -m mobilesuperresolution_amd.build --debug)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import _lib as L, hotpath as HP
f, n, nb, wgs = 24, 32, 16, 16
tb = HP.tables(f, torch.device("cuda", 0))
src = torch.randn(nb, tb["src_size"], device="cuda") * 0.1
blob, cinit = HP.pack_blocks(src, f, torch.bfloat16)
xs = torch.randn(nb, n, 48, 48, f, device="cuda").bfloat16()
dys = torch.randn(nb, n, 48, 48, f, device="cuda").bfloat16()
pa = torch.empty(nb, wgs, tb["slab_a"], device="cuda"); pb = torch.empty(nb, wgs, tb["slab_b"], device="cuda")
st = torch.zeros(2 * nb * wgs * 128, dtype=torch.int64, device="cuda")
for it in range(3):
    L.check(L.lib().sr_wdsr_block_wgrad_stamps(xs.data_ptr(), dys.data_ptr(), blob.data_ptr(), cinit.data_ptr(), pa.data_ptr(),
            pb.data_ptr(), nb, wgs, n, 48, 48, xs.stride(0), dys.stride(0), blob.stride(0), cinit.stride(0), st.data_ptr(),
            L.stream_ptr()), "stamps")
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(2, nb * wgs, 128).astype(np.float64) * 10.0
for role in range(2):
    a = s[role]
    print(f"ROLE {role}: kernel span {a[:, :100].max() - a[:, 0].min():.0f} ns; prologue (weights) {np.mean(a[:,1]-a[:,0]):.0f} ns")
    names = ["top barrier wait", "issue prefetch", "phase1", "barrier", "phase2", "store next + loop"]
    tiles = 16
    per = np.zeros(6)
    for t in range(tiles):
        base = 1 + 6 * t
        for k in range(5):
            per[k] += np.mean(a[:, base + k + 1] - a[:, base + k])
        if t + 1 < tiles:
            per[5] += np.mean(a[:, base + 6] - a[:, base + 5])
    for k in range(6):
        print(f"   {names[k]:18s} {per[k] / tiles:7.0f} ns per tile")
    e = 1 + 6 * tiles - 1
    names2 = ["wait others (barrier)", "zero slab + barrier", "atomics + barrier", "global store"]
    for k in range(4):
        print(f"   epilogue/{names2[k]:24s} {np.mean(a[:, e + k + 1] - a[:, e + k]):7.0f} ns")
