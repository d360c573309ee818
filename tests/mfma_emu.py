"""Lane-level numpy emulation of the gfx950 MFMA conventions the kernels rely on
(csrc/sr_common.h) and of the wave algorithms built on them.  TEST INFRASTRUCTURE:
it lets the CPU suite check the host packing tables (mobilesuperresolution_amd/packing.py)
and the kernels' index arithmetic against the oracle without a GPU.  The lane maps
themselves are verified on hardware by tests/test_gpu_probe.py.
"""
import numpy as np
import torch

LANE = np.arange(64)
R = LANE & 31
HH = LANE >> 5


def rnd(x, dtype):
    """round to the kernel's storage type and come back as float64"""
    if dtype == "bf16":
        return torch.from_numpy(np.asarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float64).numpy()
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def mma16(a, b, acc):
    """a, b: (64, 8) fragments; acc: (64, 16).  D = A*B + C for one 16-deep k-step."""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    for j in range(8):
        A[R, 8 * HH + j] = a[:, j]
        B[8 * HH + j, R] = b[:, j]
    D = A @ B
    out = acc.copy()
    for i in range(16):
        out[:, i] += D[(i & 3) + 8 * (i >> 2) + 4 * HH, R]
    return out


def acc_to_frag(acc, s, dtype):
    return rnd(acc[:, 8 * s:8 * s + 8], dtype)


def cinit(tab, hh=HH):
    """tab: float[2][16] flattened -> (64,16) accumulator"""
    return np.asarray(tab, dtype=np.float64).reshape(2, 16)[hh]


def wfrag(packed, idx):
    return np.asarray(packed[idx * 512:(idx + 1) * 512], dtype=np.float64).reshape(64, 8)


def emu_block_fwd(x_nhwc, packed_w, cinit_tab, geom, dtype="f32", TH=12, TW=24):
    """Mirror of wdsr_block_fwd_kernel (csrc/wdsr_block.hip) for one image x_nhwc (H, W, F)."""
    g = geom
    H, W, F = x_nhwc.shape
    HW_, HH_ = TW + 2, TH + 2
    NPXH = HW_ * HH_
    NPXH_PAD = (NPXH + 31) // 32 * 32
    KX, LP, CPT, FC = g.KX, g.LP, g.CPT, g.FC
    w1b, w2b, w3b = 0, g.NET * g.KS1, g.NET * g.KS1 + g.KS2
    pw = rnd(packed_w, dtype)
    y = np.zeros((H, W, F))
    for ty0 in range(0, H, TH):
        for tx0 in range(0, W, TW):
            Xs = np.zeros((NPXH_PAD, KX))
            valid = np.zeros(NPXH_PAD, dtype=bool)
            for hp in range(NPXH):
                Y, X = ty0 - 1 + hp // HW_, tx0 - 1 + hp % HW_
                if 0 <= Y < H and 0 <= X < W:
                    Xs[hp, :F] = x_nhwc[Y, X]
                    valid[hp] = True
            if g.fold_b1:
                Xs[:, F] = 1.0
            Xs = rnd(Xs, dtype)
            Ts = np.zeros((NPXH_PAD, LP))
            for pt in range(NPXH_PAD // 32):
                hp = pt * 32 + R
                xb = [np.stack([Xs[hp, (2 * s + HH) * 8 + j] for j in range(8)], 1) for s in range(g.KS1)]
                tacc = cinit(cinit_tab[0:32])
                for et in range(g.NET):
                    hacc = np.zeros((64, 16)) if g.fold_b1 else cinit(cinit_tab[32 + 32 * et:64 + 32 * et])
                    for s in range(g.KS1):
                        hacc = mma16(wfrag(pw, w1b + et * g.KS1 + s), xb[s], hacc)
                    hacc = np.maximum(hacc, 0)
                    for s in range(2):
                        if 2 * et + s < g.KS2:
                            tacc = mma16(wfrag(pw, w2b + 2 * et + s), acc_to_frag(hacc, s, dtype), tacc)
                for gq in range(CPT):
                    for jj in range(4):
                        v = np.where(valid[hp], rnd(tacc[:, 4 * gq + jj], dtype), 0.0)
                        Ts[hp, gq * 8 + HH * 4 + jj] = v
            for ot in range((TH // 4) * (TW // 8)):
                oy = (ot // (TW // 8)) * 4 + (R >> 3)
                ox = (ot % (TW // 8)) * 8 + (R & 7)
                hbase = oy * HW_ + ox
                oacc = np.zeros((64, 16))
                for s in range(g.KS3):
                    q = 2 * s + HH
                    b = np.zeros((64, 8))
                    for l in range(64):
                        if q[l] < 9 * CPT:
                            tap, c = q[l] // CPT, q[l] % CPT
                            b[l] = Ts[hbase[l] + (tap // 3) * HW_ + tap % 3, c * 8:c * 8 + 8]
                        else:
                            c = q[l] - 9 * CPT
                            c = c if c < FC else 0
                            b[l] = Xs[hbase[l] + HW_ + 1, c * 8:c * 8 + 8]
                    oacc = mma16(wfrag(pw, w3b + s), b, oacc)
                for l in range(64):
                    Y, X = ty0 + oy[l], tx0 + ox[l]
                    if Y < H and X < W:
                        for gq in range(FC):
                            for jj in range(4):
                                y[Y, X, gq * 8 + HH[l] * 4 + jj] = oacc[l, 4 * gq + jj]
    return rnd(y, dtype)
