"""Time the slab reduction (unpack_all_kernel + wn_bwd_kernel through sr_param_grads) on the tables of a BASIC_MODEL:
tools/time_unpack.py [units] -- per segment alone and all four together, 50 launches back to back."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd import _lib as L
from mobilesuperresolution_amd.models import get_model

units = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda", 0)
ns = argparse.Namespace(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=16, num_residual_units=units,
                        hot_dtype="bf16")
m = get_model(ns).to(dev).train()
st = m._state(dev)
lay = m.layout
n = st.net
for t in (st.part_a, st.part_b, st.part_tail, st.part_head):
    t.normal_()
gflat = torch.zeros_like(m.flat)


def seg(part, key, off, stride, slab, wgs, reps):
    sidx, dst = st.g[key]
    return L.UnpackSeg(part.data_ptr(), sidx.data_ptr(), dst.data_ptr(), off, stride, slab, wgs, sidx.numel(), reps)


segs = {"a": seg(st.part_a, "ga", lay.src_body_off, lay.src_body_stride, lay.slab_a, st.wgs_body, lay.NB),
        "b": seg(st.part_b, "gb", lay.src_body_off, lay.src_body_stride, lay.slab_b, st.wgs_body, lay.NB),
        "tail": seg(st.part_tail, "gt", lay.src_tail_off, 0, lay.slab_tail, st.wgs_tail, 1),
        "head": seg(st.part_head, "gh", lay.src_head_off, 0, lay.slab_head, st.wgs_head, 1)}
print({k: dict(n=s.n, wgs=s.wgs, reps=s.reps, slab=s.slab, MB=round(s.n * s.wgs * s.reps * 4 / 1e6, 2)) for k, s in segs.items()})


def run(names, reps=50):
    arr = (L.UnpackSeg * len(names))(*[segs[k] for k in names])
    def go():
        L.launch("sr_param_grads", L.lib().sr_param_grads, m.flat.data_ptr(), st.dsrc.data_ptr(), gflat.data_ptr(), st.chan_tab.data_ptr(),
                 lay.chan_tab.shape[0], st.bias_tab.data_ptr(), lay.bias_tab.shape[0], arr, len(names), L.stream_ptr())
    for _ in range(5):
        go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        go()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for names in (["head"], ["a"], ["b"], ["tail"], ["a", "b"], ["a", "b", "tail", "head"]):
    print(f"units {units}: {'+'.join(names):16s} {run(names):7.2f} us per (unpack + weight-norm backward) pair of launches")
