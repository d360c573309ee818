"""Hardware probes: the MFMA / transposed-LDS-read lane maps that csrc/sr_common.h and
packing.py assume, checked against the numpy emulation used by the CPU tests."""
import numpy as np
import pytest
import torch

from tests import mfma_emu as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from mobilesuperresolution_amd import _lib
    return _lib


def test_mfma_bf16_lane_map(L):
    g = torch.Generator().manual_seed(0)
    a = torch.randint(-8, 9, (64, 8), generator=g).float()       # exact in bf16, asymmetric
    b = torch.randint(-8, 9, (64, 8), generator=g).float()
    out = torch.zeros(64, 16, device="cuda")
    ad, bd = a.cuda().bfloat16().contiguous(), b.cuda().bfloat16().contiguous()
    L.check(L.lib().sr_probe_mfma_bf16(L.ptr(ad), L.ptr(bd), L.ptr(out), L.stream_ptr()), "probe")
    exp = M.mma16(a.double().numpy(), b.double().numpy(), np.zeros((64, 16)))
    assert np.array_equal(out.cpu().double().numpy(), exp)


def test_mfma_f32_lane_map(L):
    g = torch.Generator().manual_seed(1)
    a = torch.randint(-100, 101, (64, 8), generator=g).float()
    b = torch.randint(-100, 101, (64, 8), generator=g).float()
    out = torch.zeros(64, 16, device="cuda")
    ad, bd = a.cuda().contiguous(), b.cuda().contiguous()
    L.check(L.lib().sr_probe_mfma_f32(L.ptr(ad), L.ptr(bd), L.ptr(out), L.stream_ptr()), "probe")
    exp = M.mma16(a.double().numpy(), b.double().numpy(), np.zeros((64, 16)))
    assert np.array_equal(out.cpu().double().numpy(), exp)


def test_tr_read_lane_map(L):
    """ds_read_b64_tr_b16: in each 16-lane group, lane 4q+p supplies row q / columns 4p..4p+3,
    lane i receives column i of the 4 rows."""
    rows, cols = 64, 64
    img = torch.arange(rows * cols, dtype=torch.float32).reshape(rows, cols) % 251
    lane = np.arange(64)
    grp, gi = lane >> 4, lane & 15
    q, p = gi >> 2, gi & 3
    # group g reads 4 arbitrary non-contiguous rows at column base 16*(g%2)
    rowsel = np.array([[3 * g + 5, 3 * g + 17, 3 * g + 2, 3 * g + 40] for g in range(4)])
    colbase = 16 * (grp % 2)
    off = rowsel[grp, q] * cols + colbase + 4 * p
    out = torch.zeros(64, 4, device="cuda", dtype=torch.bfloat16)
    imgd = img.cuda().bfloat16().contiguous()
    offd = torch.from_numpy(off.astype(np.int32)).cuda()
    L.check(L.lib().sr_probe_tr_read(L.ptr(imgd), rows * cols, L.ptr(offd), L.ptr(out), L.stream_ptr()), "probe")
    exp = np.zeros((64, 4))
    for l in range(64):
        for qq in range(4):
            exp[l, qq] = img[rowsel[grp[l], qq], colbase[l] + gi[l]]
    got = out.float().cpu().numpy()
    assert np.array_equal(got, exp), (got[:20], exp[:20])


def test_copy_bandwidth_probe(L):
    n = 1 << 28
    src = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
    dst = torch.empty_like(src)
    for _ in range(2):
        L.check(L.lib().sr_probe_copy(L.ptr(src), L.ptr(dst), n, L.stream_ptr()), "copy")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.check(L.lib().sr_probe_copy(L.ptr(src), L.ptr(dst), n, L.stream_ptr()), "copy")
    e1.record()
    torch.cuda.synchronize()
    gbs = 2 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    print(f"\ncopy bandwidth: {gbs:.0f} GB/s")
    assert torch.equal(src, dst)
    assert gbs > 1000
