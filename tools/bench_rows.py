"""Measure the other SURVEY 8 rows on one MI355X (numbers quoted in DESIGN.md; bench.py stays the C2 line):
C2 in fp32 parity mode, C3 (32 units), C4 (BasicVSR propagation over 5-frame 64x64 clips, given flows), C5 (NAS
supernet step), flow_warp alone, block-forward kernels over a batch sweep.  Prints one JSON object."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from mobilesuperresolution_amd import _lib as L
from mobilesuperresolution_amd.models import get_model, ConvResidualBlocks, flow_warp
from mobilesuperresolution_amd.models.basicvsr_arch import propagate

dev = torch.device("cuda", 0)


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def sr_step(ns, batch, lr=48):
    torch.manual_seed(0)
    m = get_model(ns).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
    x = torch.rand(batch, 3, lr, lr, device=dev)
    hr = torch.rand(batch, 3, lr * ns.scale, lr * ns.scale, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        out = m(x)
        out = out[0] if isinstance(out, tuple) else out
        torch.nn.functional.l1_loss(out, hr).backward()
        opt.step()
    t = timeit(step)
    dropin = None
    if ns.model_type == "NAS_MODEL":                     # search.py:72-89 with training.L1Loss in place of nn.L1Loss (the loss folded
        from mobilesuperresolution_amd.training import L1Loss   # into the tail backward) and the speed term
        crit = L1Loss()

        def step_dropin():
            opt.zero_grad(set_to_none=True)
            sr, speed = m(x)
            (1.0 * crit(sr, hr) + 0.1 * speed.sum()).backward()
            opt.step()
        dropin = timeit(step_dropin)
    fused = None
    if hasattr(m, "train_step"):                         # the fused route: loss folded into the tail backward + Adam kernel, one C call
        st = m.make_train_state(1e-3)
        fused = timeit(lambda: m.train_step(x, hr, st))
    m.eval()
    with torch.no_grad():
        tf = timeit(lambda: m(x))
    mp = batch * (lr * ns.scale) ** 2 / 1e6
    row = {"train_ms": round(t * 1e3, 4), "train_HR_Mpix_s": round(mp / t, 1), "fwd_ms": round(tf * 1e3, 4),
           "fwd_HR_Mpix_s": round(mp / tf, 1)}
    if dropin is not None:
        row.update({"train_dropin_loss_ms": round(dropin * 1e3, 4), "train_dropin_loss_HR_Mpix_s": round(mp / dropin, 1)})
    if fused is not None:
        row.update({"train_step_ms": round(fused * 1e3, 4), "train_step_HR_Mpix_s": round(mp / fused, 1)})
    return row


out = {}
base = dict(model_type="BASIC_MODEL", image_mean=0.5, num_channels=3, scale=4)
out["C2_bf16"] = sr_step(argparse.Namespace(**base, num_blocks=16, num_residual_units=24, hot_dtype="bf16"), 32)
out["C2_fp32_parity_mode"] = sr_step(argparse.Namespace(**base, num_blocks=16, num_residual_units=24, hot_dtype="fp32"), 32)
out["C3_bf16_32units_per_gpu"] = sr_step(argparse.Namespace(**base, num_blocks=16, num_residual_units=32, hot_dtype="bf16"), 32)
out["C1_shape_batch1_4blocks_bf16"] = sr_step(argparse.Namespace(**base, num_blocks=4, num_residual_units=24, hot_dtype="bf16"), 1)
nas = argparse.Namespace(model_type="NAS_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=16,
                         num_residual_units=32, width_search=True, pretrained=False, hot_dtype="bf16")
try:
    out["C5_nas_bf16_32units"] = sr_step(nas, 32)
except Exception as e:                                   # keep the other rows if a ctor flag is missing
    out["C5_nas_bf16_32units"] = {"error": repr(e)}

# C4: recurrent propagation, 5 frames of 64x64, 24 features, 8 residual blocks per trunk, given flows
b, n, h, w = 8, 5, 64, 64
bt = ConvResidualBlocks(27, 24, 8, hot_dtype="bf16").to(dev)
ft = ConvResidualBlocks(27, 24, 8, hot_dtype="bf16").to(dev)
clip = torch.rand(b, n, 3, h, w, device=dev)
fl_f = (torch.rand(b, n - 1, 2, h, w, device=dev) * 4 - 2)
fl_b = (torch.rand(b, n - 1, 2, h, w, device=dev) * 4 - 2)
params = list(bt.parameters()) + list(ft.parameters())


def vsr_step():
    for p in params:
        p.grad = None
    ob, of = propagate(clip, fl_f, fl_b, bt, ft, flow_warp)
    (sum(o.sum() for o in ob) + sum(o.sum() for o in of)).backward()
gones = torch.ones(b, 24, h, w, device=dev)


def vsr_step_injected():
    """the same gradient (ones) handed to the ten feature maps directly: without the harness's ten reductions and nine adds"""
    for p in params:
        p.grad = None
    ob, of = propagate(clip, fl_f, fl_b, bt, ft, flow_warp)
    torch.autograd.backward(ob + of, [gones] * (len(ob) + len(of)))
t_sum = timeit(vsr_step, n=10, warm=3)
t = timeit(vsr_step_injected, n=10, warm=3)
with torch.no_grad():
    tf = timeit(lambda: propagate(clip, fl_f, fl_b, bt, ft, flow_warp), n=10, warm=3)
out["C4_vsr_propagation_bf16"] = {"clips": b, "frames": n, "train_ms": round(t * 1e3, 3), "train_with_sum_loss_ms": round(t_sum * 1e3, 3),
                                  "fwd_ms": round(tf * 1e3, 3), "LR_frames_per_s_fwd": round(b * n / tf, 1)}
# SPyNet on C4's frame pairs: 8 clips x 4 pairs x 2 directions of 64 x 64 (7x7 conv pyramid as MFMA kernels)
from mobilesuperresolution_amd.models import SpyNet, BasicVSR_origin
sp = SpyNet().to(dev).eval()
r1, r2 = torch.rand(64, 3, 64, 64, device=dev), torch.rand(64, 3, 64, 64, device=dev)
tsp = timeit(lambda: sp(r1, r2), n=10, warm=3)
# 2 x 49 x (8*32 + 32*64 + 64*32 + 32*16 + 16*2) flop per pixel and level
gf = 64 * sum((64 >> l) ** 2 for l in range(6)) * 2 * 49 * (8 * 32 + 32 * 64 + 64 * 32 + 32 * 16 + 16 * 2) / 1e9
out["spynet_64_pairs_64x64"] = {"ms": round(tsp * 1e3, 3), "GFLOP": round(gf, 1), "TFLOP_s": round(gf / tsp / 1e3, 1)}
bvo = BasicVSR_origin(num_feat=24, num_block=8, hot_dtype="bf16").to(dev).eval()
with torch.no_grad():
    tb = timeit(lambda: bvo(clip, 256, 256), n=5, warm=2)
out["BasicVSR_origin_fwd_with_spynet_8clips"] = {"ms": round(tb * 1e3, 3)}
out["C2_bf16_batch256"] = sr_step(argparse.Namespace(**base, num_blocks=16, num_residual_units=24, hot_dtype="bf16"), 256)
feat = torch.rand(32, 24, 64, 64, device=dev)
flow = torch.rand(32, 64, 64, 2, device=dev) * 4 - 2
with torch.no_grad():
    tw = timeit(lambda: flow_warp(feat, flow))
out["flow_warp_32x24x64x64_fp32"] = {"us": round(tw * 1e6, 1), "GB_s_algorithmic": round((2 * feat.numel() * 4 + flow.numel() * 4) / tw / 1e9, 1)}

# block-forward kernels over a batch sweep (bf16, F = 24): how far the kernel itself is from the HBM roof
m = get_model(B.model_ns("bf16")).to(dev)
st = m._state(dev)
sweep = {}
for batch in (32, 64, 128, 256, 512):
    a = torch.randn(batch, 48, 48, 24, device=dev).bfloat16()
    bb, c = torch.empty_like(a), torch.empty_like(a)
    reps = 64
    f1 = lambda: L.check(L.lib().sr_wdsr_block_fwd_repeat(a.data_ptr(), bb.data_ptr(), st.blob_body[0].data_ptr(),
                         st.cinit_body[0].data_ptr(), batch, 48, 48, 24, 1, reps, L.stream_ptr()), "r1")
    f2 = lambda: L.check(L.lib().sr_wdsr_block2_fwd_repeat(a.data_ptr(), bb.data_ptr(), c.data_ptr(), st.blob_body[0].data_ptr(),
                         st.blob_body[1].data_ptr(), st.cinit_body[0].data_ptr(), st.cinit_body[1].data_ptr(), batch, 48, 48,
                         24, 1, reps, L.stream_ptr()), "r2")
    u1, u2 = timeit(f1, 5, 2) / reps, timeit(f2, 5, 2) / reps
    alg = 2 * batch * 48 * 48 * 24 * 2
    sweep[str(batch)] = {"single_us": round(u1 * 1e6, 2), "single_GB_s": round(alg / u1 / 1e9), "pair_us": round(u2 * 1e6, 2),
                         "pair_GB_s": round(2 * alg / u2 / 1e9)}
out["block_fwd_batch_sweep_bf16"] = sweep

# f3 / f4 rows of SURVEY section 8: the device input pipeline (32 training items cut from a resident uint8 cache: crop, flips, axis swap,
# uint8 -> float, LR 48x48 + HR 192x192) and the evaluation metrics on a batch of 32 HR patches
try:
    import random
    from mobilesuperresolution_amd.datasets import DevicePatchCache
    from mobilesuperresolution_amd import metrics as MET
    gimg = torch.Generator().manual_seed(5)
    hr_chw = [(torch.rand(3, 1356 + 12 * i, 2040, generator=gimg) * 255) for i in range(4)]      # DIV2K-sized images
    lr_imgs = [torch.nn.functional.avg_pool2d(h, 4).byte().permute(1, 2, 0).contiguous() for h in hr_chw]
    hr_imgs = [h.byte().permute(1, 2, 0).contiguous() for h in hr_chw]
    cache = DevicePatchCache(lr_imgs, hr_imgs, 48, 4, ignored_boundary_size=0, num_patches=8, device=dev)
    rng = random.Random(0)
    idx = list(range(32))
    tb = timeit(lambda: cache.batch(idx, rng), n=30, warm=5)
    nbytes = 32 * 3 * (48 * 48 + 192 * 192) * (1 + 4)                   # uint8 read + fp32 written
    out["f3_patch_batch_32_items"] = {"us_incl_host_draws": round(tb * 1e6, 1), "GB_s_algorithmic": round(nbytes / tb / 1e9, 1)}
    sr_ = torch.rand(32, 3, 192, 192, device=dev)
    hr_ = torch.rand(32, 3, 192, 192, device=dev)
    tp = timeit(lambda: MET.psnr(sr_, hr_), n=30, warm=5)
    ty = timeit(lambda: MET.psnr_y(sr_, hr_), n=30, warm=5)
    out["f4_metrics_32x3x192x192"] = {"psnr_us": round(tp * 1e6, 1), "psnr_y_us": round(ty * 1e6, 1),
                                      "psnr_GB_s_algorithmic": round(2 * sr_.numel() * 4 / tp / 1e9, 1)}
except Exception as e:
    out["f3_f4_rows"] = {"error": repr(e)}
print(json.dumps(out))
