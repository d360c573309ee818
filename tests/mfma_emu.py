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


# =====================================================================================
# backward of the residual block
# =====================================================================================
def _kch(s, hh, j):
    return 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)


def _stage_bwd(x_nhwc, dy_nhwc, g, ty0, tx0, TH, TW, dtype):
    """LDS images of the backward kernels: dy with a 1-pixel halo [NPXH_PAD(+pad)][F] and the core x
    tile [TH*TW][KX] (ones channel at F when b1 is folded).  Returned flat with row strides F / KX."""
    H, W, F = x_nhwc.shape
    HW_, HH_ = TW + 2, TH + 2
    NPXH_PAD = (HW_ * HH_ + 31) // 32 * 32
    DY = np.zeros((NPXH_PAD + 2, F))
    for hp in range(HW_ * HH_):
        Y, X = ty0 - 1 + hp // HW_, tx0 - 1 + hp % HW_
        if 0 <= Y < H and 0 <= X < W:
            DY[hp] = dy_nhwc[Y, X]
    XC = np.zeros((TH * TW + 1, g.KX))
    valid = np.zeros(TH * TW, dtype=bool)
    for pc in range(TH * TW):
        Y, X = ty0 + pc // TW, tx0 + pc % TW
        if Y < H and X < W:
            XC[pc, :F] = x_nhwc[Y, X]
            valid[pc] = True
    if g.fold_b1:
        XC[:, F] = 1.0
    return rnd(DY, dtype).reshape(-1), rnd(XC, dtype).reshape(-1), valid


def _chunk(flat, base_elem):
    """(64, 8) fragment: 8 consecutive elements starting at per-lane element offsets"""
    return np.stack([flat[base_elem + j] for j in range(8)], 1)


def _dt_tile(DY, pw, sec, tab, g, hbase, HW_):
    F, FC = g.F, g.FC
    dtacc = np.zeros((64, 16))
    for s in range(tab["KS3B"]):
        q = 2 * s + HH
        u, c = q // FC, q % FC
        off = np.where(q < 9 * FC, (hbase + (u // 3) * HW_ + u % 3) * F + c * 8, hbase * F)
        dtacc = mma16(wfrag(pw, sec["W3T"] + s), _chunk(DY, off), dtacc)
    return dtacc


def emu_block_bwd_data(x_nhwc, dy_nhwc, packed_w, cinit_tab, tab, dtype="f32", TH=12, TW=24):
    """Mirror of wdsr_block_bwd_data_kernel: dx = dy + W1^T [relu'(h) * (W2^T conv3x3^T(dy))]."""
    g, sec = tab["geom"], tab["sec"]
    H, W, F = x_nhwc.shape
    HW_ = TW + 2
    pw = rnd(packed_w, dtype)
    dx = np.zeros((H, W, F))
    for ty0 in range(0, H, TH):
        for tx0 in range(0, W, TW):
            DY, XC, _ = _stage_bwd(x_nhwc, dy_nhwc, g, ty0, tx0, TH, TW, dtype)
            for ot in range((TH // 4) * (TW // 8)):
                oy = (ot // (TW // 8)) * 4 + (R >> 3)
                ox = (ot % (TW // 8)) * 8 + (R & 7)
                hbase, pc = oy * HW_ + ox, oy * TW + ox
                dtacc = _dt_tile(DY, pw, sec, tab, g, hbase, HW_)
                dtb = [acc_to_frag(dtacc, s, dtype) for s in range(2)]
                xb = [_chunk(XC, pc * g.KX + (2 * s + HH) * 8) for s in range(g.KS1)]
                dxacc = np.zeros((64, 16))
                for et in range(g.NET):
                    hacc = np.zeros((64, 16)) if g.fold_b1 else cinit(cinit_tab[32 + 32 * et:64 + 32 * et])
                    for s in range(g.KS1):
                        hacc = mma16(wfrag(pw, sec["W1"] + et * g.KS1 + s), xb[s], hacc)
                    dh = np.zeros((64, 16))
                    for s in range(2):
                        dh = mma16(wfrag(pw, sec["W2T"] + 2 * et + s), dtb[s], dh)
                    dpre = np.where(hacc > 0, dh, 0.0)
                    for s in range(2):
                        if 2 * et + s < g.KS2:
                            dxacc = mma16(wfrag(pw, sec["W1T"] + 2 * et + s), acc_to_frag(dpre, s, dtype), dxacc)
                for s in range(tab["KSI"]):
                    q = 2 * s + HH
                    c = np.where(q < g.FC, q, 0)
                    dxacc = mma16(wfrag(pw, sec["ID"] + s), _chunk(DY, (hbase + HW_ + 1) * F + c * 8), dxacc)
                for l in range(64):
                    Y, X = ty0 + oy[l], tx0 + ox[l]
                    if Y < H and X < W:
                        for gq in range(g.FC):
                            for jj in range(4):
                                dx[Y, X, gq * 8 + HH[l] * 4 + jj] = dxacc[l, 4 * gq + jj]
    return rnd(dx, dtype)


def _tr_frag(flat, rowbase_of_px, s, ch):
    """transposed fragment: element j = flat[rowbase(px_j) + ch], px_j = chained order of k-step s"""
    return np.stack([flat[rowbase_of_px(_kch(s, HH, j)) + ch] for j in range(8)], 1)


def emu_block_wgrad12(x_nhwc, dy_nhwc, packed_w, cinit_tab, tab, dtype="f32", TH=12, TW=24, slab=None):
    """Mirror of wdsr_block_wgrad12_kernel for one image; returns slab A (float64), accumulating
    into `slab` when given.  Layout: packing.block_grad_tables."""
    g, sec = tab["geom"], tab["sec"]
    H, W, F = x_nhwc.shape
    HW_ = TW + 2
    NET = g.NET
    pw = rnd(packed_w, dtype)
    dW1T = [np.zeros((64, 16)) for _ in range(NET)]
    dW2 = [np.zeros((64, 16)) for _ in range(NET)]
    db1 = np.zeros((NET, 64))
    db2acc = np.zeros((64, 16))
    for ty0 in range(0, H, TH):
        for tx0 in range(0, W, TW):
            DY, XC, valid = _stage_bwd(x_nhwc, dy_nhwc, g, ty0, tx0, TH, TW, dtype)
            for ot in range((TH // 4) * (TW // 8)):
                toy, tox = (ot // (TW // 8)) * 4, (ot % (TW // 8)) * 8
                oy, ox = toy + (R >> 3), tox + (R & 7)
                hbase, pc = oy * HW_ + ox, oy * TW + ox
                dtacc = _dt_tile(DY, pw, sec, tab, g, hbase, HW_)
                dtacc = np.where(valid[pc][:, None], dtacc, 0.0)
                db2acc += dtacc
                DT = np.zeros((33, 32))
                for gq in range(4):
                    for jj in range(4):
                        DT[R, gq * 8 + HH * 4 + jj] = rnd(dtacc[:, 4 * gq + jj], dtype)
                DT = DT.reshape(-1)
                dtA = [_chunk(DT, R * 32 + (2 * s + HH) * 8) for s in range(2)]
                dtT = [_tr_frag(DT, lambda p: p * 32, s, R) for s in range(2)]
                xA = [_chunk(XC, pc * g.KX + (2 * s + HH) * 8) for s in range(g.KS1)]
                pcof = lambda p: ((toy + (p >> 3)) * TW + tox + (p & 7)) * g.KX
                xT = [_tr_frag(XC, pcof, s, R) for s in range(2)]
                for et in range(NET):
                    if g.fold_b1:
                        h2 = np.zeros((64, 16))
                    else:
                        h2 = np.repeat(np.asarray(cinit_tab, dtype=np.float64)[32 + NET * 32 + 32 * et + R][:, None], 16, 1)
                    for s in range(g.KS1):
                        h2 = mma16(xA[s], wfrag(pw, sec["W1"] + et * g.KS1 + s), h2)
                    dh2 = np.zeros((64, 16))
                    for s in range(2):
                        dh2 = mma16(dtA[s], wfrag(pw, sec["W2N"] + 2 * et + s), dh2)
                    dpre2 = np.where(h2 > 0, dh2, 0.0)
                    h2r = np.maximum(h2, 0.0)
                    db1[et] += dpre2.sum(1)
                    for s in range(2):
                        dW1T[et] = mma16(xT[s], acc_to_frag(dpre2, s, dtype), dW1T[et])
                        dW2[et] = mma16(dtT[s], acc_to_frag(h2r, s, dtype), dW2[et])
    out = np.concatenate([a.T.reshape(-1) for a in dW1T] + [a.T.reshape(-1) for a in dW2])   # [tile][reg][lane]
    b1 = (db1[:, :32] + db1[:, 32:]).reshape(-1)
    b2 = np.zeros(32)
    for i in range(16):
        for hh in range(2):
            b2[(i & 3) + 8 * (i >> 2) + 4 * hh] = db2acc[HH == hh, i].sum()
    out = np.concatenate([out, b1, b2])
    return out if slab is None else slab + out


def emu_block_wgrad3(x_nhwc, dy_nhwc, packed_w, cinit_tab, tab, dtype="f32", TH=12, TW=24, slab=None):
    """Mirror of wdsr_block_wgrad3_kernel for one image; returns slab B (9 raw accumulator tiles)."""
    g, sec = tab["geom"], tab["sec"]
    H, W, F = x_nhwc.shape
    HW_ = TW + 2
    pw = rnd(packed_w, dtype)
    dW3T = [np.zeros((64, 16)) for _ in range(9)]
    for ty0 in range(0, H, TH):
        for tx0 in range(0, W, TW):
            DY, XC, valid = _stage_bwd(x_nhwc, dy_nhwc, g, ty0, tx0, TH, TW, dtype)
            for ot in range((TH // 4) * (TW // 8)):
                toy, tox = (ot // (TW // 8)) * 4, (ot % (TW // 8)) * 8
                oy, ox = toy + (R >> 3), tox + (R & 7)
                pc = oy * TW + ox
                xb = [_chunk(XC, pc * g.KX + (2 * s + HH) * 8) for s in range(g.KS1)]
                tacc = cinit(cinit_tab[0:32])
                for et in range(g.NET):
                    hacc = np.zeros((64, 16)) if g.fold_b1 else cinit(cinit_tab[32 + 32 * et:64 + 32 * et])
                    for s in range(g.KS1):
                        hacc = mma16(wfrag(pw, sec["W1"] + et * g.KS1 + s), xb[s], hacc)
                    hacc = np.maximum(hacc, 0)
                    for s in range(2):
                        if 2 * et + s < g.KS2:
                            tacc = mma16(wfrag(pw, sec["W2"] + 2 * et + s), acc_to_frag(hacc, s, dtype), tacc)
                tacc = np.where(valid[pc][:, None], tacc, 0.0)
                TS = np.zeros((33, 32))
                for gq in range(4):
                    for jj in range(4):
                        TS[R, gq * 8 + HH * 4 + jj] = rnd(tacc[:, 4 * gq + jj], dtype)
                TS = TS.reshape(-1)
                tT = [_tr_frag(TS, lambda p: p * 32, s, R) for s in range(2)]
                for u in range(9):
                    hof = lambda p, u=u: ((toy + (p >> 3) + u // 3) * HW_ + tox + (p & 7) + u % 3) * F
                    for s in range(2):
                        dW3T[u] = mma16(tT[s], _tr_frag(DY, hof, s, R), dW3T[u])
    out = np.concatenate([a.T.reshape(-1) for a in dW3T])          # [tile][reg][lane]
    return out if slab is None else slab + out


def emu_conv3_dense(t_nhwc, x_nhwc, packed_w3d, L, F, dtype="f32"):
    """Mirror of rw_phase_b / RwBAddrD / rw_resid_init (csrc/wdsr_fwd_rs.h): the dense-K 3x3 conv + bias + residual of one image.
    t_nhwc (H, W, L), x_nhwc (H, W, F).  The t image is laid out as the kernel's LDS region: row-major over a (H + 2) x (W + 2)
    region (zeros outside the image), rows of L real channels; output tiles are 32 consecutive pixels of the H x W core region
    numbered row-major over a region of width RWO = W (region pixel (hy, hx) <-> window top-left t row (hy, hx))."""
    H, W, _ = t_nhwc.shape
    TD = (L + 3) // 4 * 4                                            # BlockCfg::TD / LCD / KPR / HALF / QONE
    LCD = TD // 4
    KPR = (3 * LCD + 1 + 3) // 4
    HALF = 2 * KPR
    QONE = (3 * LCD - HALF) // 2
    assert 3 * LCD >= HALF and (3 * LCD - HALF) % 2 == 1
    RWO, RWI = W, W + 2
    NPI = (H + 2) * RWI
    T = np.zeros(((NPI + 31) // 32 * 32 + 32) * TD)                  # + slack rows, as the LDS image has behind it
    Tv = T[:NPI * TD].reshape(H + 2, RWI, TD)
    Tv[1:H + 1, 1:W + 1, :L] = rnd(t_nhwc, dtype)
    if TD > L:
        Tv[1:H + 1, 1:W + 1, L] = 1.0                                # the kernel's t rows carry the ones channel there (zero weights)
    ones = np.zeros(8)
    ones[0] = 1.0
    X = rnd(x_nhwc, dtype)
    pw = rnd(packed_w3d, dtype)
    y = np.zeros((H, W, F))
    NPO = H * RWO
    for tile in range((NPO + 31) // 32):
        hp = tile * 32 + R
        hpc = np.where(hp < NPO, hp, 0)
        hy, hx = hpc // RWO, hpc % RWO
        acc = np.zeros((64, 16))
        for g in range(4):                                          # rw_resid_init: rows f = 8 g + 4 hh + k of this pixel's x row
            for k in range(4):
                f = 8 * g + 4 * HH + k
                acc[:, 4 * g + k] = np.where(f < F, X[hy, hx, np.minimum(f, F - 1)], 0.0) if g < F // 8 else 0.0
        b0 = (hy * RWI + hx) * TD + HH * (HALF * 4)
        for s in range(3 * KPR):
            off = (s // KPR) * RWI * TD + (s % KPR) * 8
            b = np.zeros((64, 8))
            for l in range(64):
                b[l, :4] = T[b0[l] + off:b0[l] + off + 4]
                if s == 2 * KPR + QONE and HH[l]:
                    b[l, 4:] = ones[:4]
                else:
                    b[l, 4:] = T[b0[l] + off + 4:b0[l] + off + 8]
            acc = mma16(wfrag(pw, s), b, acc)
        for l in range(64):
            if hp[l] < NPO:
                for g in range(F // 8):
                    for k in range(4):
                        y[hy[l], hx[l], 8 * g + 4 * HH[l] + k] = acc[l, 4 * g + k]
    return y
