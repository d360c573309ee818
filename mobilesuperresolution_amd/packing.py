"""Host-side index tables that lay weights out as MFMA operand fragments.

The HIP kernels never see a (Cout, Cin, kh, kw) tensor.  They read "packed
fragments": for fragment #i, lane l = 0..63 and element j = 0..7 the value at
``packed[(i*64 + l)*8 + j]``.  This module builds, once per layer geometry, the
int64 tables ``idx`` such that ``packed = src[idx]`` where ``src`` is the
layer's canonical parameter vector (effective weight after weight-norm, then
biases, then the constants 0.0 and 1.0).  A single torch gather therefore does
zero padding, bias folding, k-order permutation and transposition at once, and
its autograd transpose scatters packed gradients back.

Fragment / accumulator lane maps (wave64, r = l & 31, hh = l >> 5), see
csrc/sr_common.h:
  natural k of element j in k-step s :  16 s + 8 hh + j
  chained k (operand = previous accumulator): 16 s + 8 (j>>2) + 4 hh + (j&3)
  accumulator reg i -> row (i&3) + 8 (i>>2) + 4 hh, column r
"""
from __future__ import annotations

from dataclasses import dataclass
from functools import lru_cache
from typing import Dict

import numpy as np

LANES = 64


def _grid(nfrag: int):
    """(s, r, hh, j) index arrays of shape (nfrag, 64, 8)."""
    s = np.arange(nfrag).reshape(-1, 1, 1)
    lane = np.arange(LANES).reshape(1, -1, 1)
    j = np.arange(8).reshape(1, 1, -1)
    s, lane, j = np.broadcast_arrays(s, lane, j)
    return s, lane & 31, lane >> 5, j


def k_natural(s, hh, j):
    return 16 * s + 8 * hh + j


def k_chained(s, hh, j):
    return 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)


def acc_row(i, hh):
    return (i & 3) + 8 * (i >> 2) + 4 * hh


def cinit_index(rows_src: np.ndarray, zero: int) -> np.ndarray:
    """C-init table float[2][16] for one 32-row tile: entry (hh, i) = src[rows_src[row]]
    (rows_src has 32 entries; use `zero` for rows without a bias)."""
    hh = np.arange(2).reshape(2, 1)
    i = np.arange(16).reshape(1, 16)
    return rows_src[acc_row(i, hh)].reshape(-1)


@dataclass(frozen=True)
class BlockGeom:
    """WDSR-B residual block geometry (models/basic_wdsr_b.py:98-140 in the reference):
    1x1 F->E, ReLU, 1x1 E->L, 3x3 L->F, + identity."""
    F: int
    E: int
    L: int

    @property
    def KX(self):            # x channels held in LDS (16-multiple; spare slots carry the ones channel)
        return self.F if self.F % 16 == 0 else (self.F // 16 + 1) * 16

    @property
    def fold_b1(self):       # bias of conv1 rides on the spare "ones" input channel
        return self.F % 16 != 0

    @property
    def KS1(self):
        return self.KX // 16

    @property
    def NET(self):           # 32-row tiles over the expand dimension
        return (self.E + 31) // 32

    @property
    def KS2(self):
        return (self.E + 15) // 16

    @property
    def LP(self):            # t channels held in LDS: L real + 1 ones channel, padded to 8
        return ((self.L + 1 + 7) // 8) * 8

    @property
    def CPT(self):           # 8-channel chunks per tap
        return self.LP // 8

    @property
    def FC(self):
        return self.F // 8

    @property
    def KS3(self):
        return (9 * self.CPT + self.FC + 1) // 2

    # canonical per-block source vector: w1 | w2 | w3 | b1 | b2 | b3 | 0 | 1
    @property
    def off(self) -> Dict[str, int]:
        F, E, L = self.F, self.E, self.L
        o, d = 0, {}
        for name, n in (("w1", E * F), ("w2", L * E), ("w3", F * L * 9), ("b1", E), ("b2", L), ("b3", F),
                        ("zero", 1), ("one", 1)):
            d[name] = o
            o += n
        d["size"] = o
        return d


def _sel(cond, a, b):
    return np.where(cond, a, b)


@lru_cache(maxsize=None)
def block_fwd_tables(F: int, E: int, L: int):
    """Tables for wdsr_block_fwd_kernel.  Returns dict(w=idx of packed weight blob,
    cinit=idx of float C-init tables, plus fragment counts)."""
    g = BlockGeom(F, E, L)
    o = g.off
    Z, ONE = o["zero"], o["one"]
    assert g.L < 32 and g.F <= 32 and F % 8 == 0

    # W1 as A operand: rows e (tile et), k natural over x channels (+ ones channel carrying b1)
    s, r, hh, j = _grid(g.NET * g.KS1)
    et, ks = s // g.KS1, s % g.KS1
    e = 32 * et + r
    k = k_natural(ks, hh, j)
    w1 = np.full(s.shape, Z, dtype=np.int64)
    ok = e < E
    w1 = _sel(ok & (k < F), o["w1"] + np.minimum(e, E - 1) * F + np.minimum(k, F - 1), w1)
    if g.fold_b1:
        w1 = _sel(ok & (k == F), o["b1"] + np.minimum(e, E - 1), w1)

    # W2 as A operand: rows l, k = e in chained order (k-step 2*et+s covers rows 16 s.. of h-tile et)
    s, r, hh, j = _grid(g.KS2)
    e = k_chained(s, hh, j)
    w2 = _sel((r < L) & (e < E), o["w2"] + np.minimum(r, L - 1) * E + np.minimum(e, E - 1), Z)

    # W3 as A operand: rows f_out, k = chunks q = 2s + hh; q < 9*CPT: (tap, 8 channels of t),
    # then FC chunks of x carrying the identity (residual).  t channel L is the ones channel -> b3
    # at the centre tap.
    s, r, hh, j = _grid(g.KS3)
    q = 2 * s + hh
    tap, c = q // g.CPT, q % g.CPT
    l = 8 * c + j
    is_tap = q < 9 * g.CPT
    w3 = np.full(s.shape, Z, dtype=np.int64)
    rf = np.minimum(r, F - 1)
    w3 = _sel(is_tap & (r < F) & (l < L),
              o["w3"] + (rf * L + np.minimum(l, L - 1)) * 9 + np.minimum(tap, 8), w3)
    w3 = _sel(is_tap & (r < F) & (l == L) & (tap == 4), o["b3"] + rf, w3)
    fin = 8 * (q - 9 * g.CPT) + j
    w3 = _sel((~is_tap) & (r < F) & (fin == r), ONE, w3)

    w = np.concatenate([w1.reshape(-1), w2.reshape(-1), w3.reshape(-1)])

    # C-init tables: [b2c (ones at row L)] + [b1c per e-tile when b1 is not folded]
    rows = np.full(32, Z, dtype=np.int64)
    rows[:L] = o["b2"] + np.arange(L)
    rows[L] = ONE
    cin = [cinit_index(rows, Z)]
    if not g.fold_b1:
        for t in range(g.NET):
            rows = np.full(32, Z, dtype=np.int64)
            ee = 32 * t + np.arange(32)
            rows[ee < E] = o["b1"] + ee[ee < E]
            cin.append(cinit_index(rows, Z))
    return dict(w=w, cinit=np.concatenate(cin), n_w1=g.NET * g.KS1, n_w2=g.KS2, n_w3=g.KS3, geom=g)


@lru_cache(maxsize=None)
def block_tables(F: int, E: int, L: int):
    """Tables for all residual-block kernels (forward, backward-data, weight-gradient).

    Blob sections (in 512-element fragments), forward first so wdsr_block_fwd_kernel's offsets hold:
      W1 | W2 | W3 | W3T | W2T | W1T | ID | W2N | W3D
    C-init floats: b2c[32] | b1c[NET][32] (only when b1 is not folded) | b1n[NET*32] (same condition).
    """
    fwd = block_fwd_tables(F, E, L)
    g = fwd["geom"]
    o = g.off
    Z, ONE = o["zero"], o["one"]
    KS3B = (9 * g.FC + 1) // 2
    KSI = (g.FC + 1) // 2

    # W3T as A operand of dt^T[l, px] = sum_{u, f} W3[f, l, 8-u] dy[px + u - 1, f]: rows l, chunk q = 2s+hh
    # -> read offset u = q // FC (row-major 3x3 into the halo'd dy tile), channels f = 8 (q % FC) + j.
    s, r, hh, j = _grid(KS3B)
    q = 2 * s + hh
    u, c = q // g.FC, q % g.FC
    f = 8 * c + j
    w3t = _sel((q < 9 * g.FC) & (r < L),
               o["w3"] + (np.minimum(f, F - 1) * L + np.minimum(r, L - 1)) * 9 + (8 - np.minimum(u, 8)), Z)

    # W2T as A operand of dh^T[e, px] = sum_l W2[l, e] dt^T[l, px]: rows e (tile et), k = l chained (2 k-steps)
    s, r, hh, j = _grid(g.NET * 2)
    et, ks = s // 2, s % 2
    e = 32 * et + r
    l = k_chained(ks, hh, j)
    w2t = _sel((e < E) & (l < L), o["w2"] + np.minimum(l, L - 1) * E + np.minimum(e, E - 1), Z)

    # W1T as A operand of dx^T[f, px] += sum_e W1[e, f] dpre^T[e, px]: rows f, k = e chained (k-step 2et+s)
    s, r, hh, j = _grid(g.KS2)
    e = k_chained(s, hh, j)
    w1t = _sel((r < F) & (e < E), o["w1"] + np.minimum(e, E - 1) * F + np.minimum(r, F - 1), Z)

    # identity over the FC chunks of dy (gradient of the skip connection)
    s, r, hh, j = _grid(KSI)
    q = 2 * s + hh
    ident = _sel((q < g.FC) & (r < F) & (8 * q + j == r), ONE, Z)

    # W2N as B operand of dh[px, e] = sum_l dt[px, l] W2[l, e]: columns e (tile et), k = l natural
    s, r, hh, j = _grid(g.NET * 2)
    et, ks = s // 2, s % 2
    e = 32 * et + r
    l = k_natural(ks, hh, j)
    w2n = _sel((e < E) & (l < L), o["w2"] + np.minimum(l, L - 1) * E + np.minimum(e, E - 1), Z)

    # W3D ("dense K", round 3): the 3x3 conv as ONE contraction over (window row ky, row chunk rc): the three taps of a window
    # row are 3 L contiguous elements of the t image (L real channels per pixel), cut into 4-channel chunks rc = 0 .. 3 L / 4 - 1;
    # k-step s = 4 ky + q gives lane half hh the chunks rc = 8 hh + 2 q + (j >> 2), i.e. 16 contiguous bytes at (window row
    # base) + 64 hh + 16 q: the same compile-time offset for both halves.  At L = 20 a row has 15 chunks: rc = 15 (the next
    # pixel's first chunk) gets zero weights, except in the last row where the kernel reads a "ones" chunk there whose first
    # slot carries b3.  12 k-steps where the 8-channel chunks of W3 (LP = 24 with the ones channel, plus the identity chunks
    # of the residual) need 15: the residual is the accumulator's initial value instead (csrc/wdsr_fwd_rs.h, rw_phase_b).
    # General form (round 3, 32 units): TD = L rounded up to a multiple of 4 (channels >= L of a pixel get zero weights), LCD = TD / 4
    # chunks per tap, KPR = ceil((3 LCD + 1) / 4) k-steps per window row, lane half hh of k-step q reads the chunk slots
    # 2 KPR hh + 2 q + (j >> 2); the ones chunk is slot 3 LCD of the last row (BlockCfg::TD / LCD / KPR / HALF in csrc/wdsr_block.h).
    w3d = None
    TD = (L + 3) // 4 * 4
    LCD = TD // 4
    KPR = (3 * LCD + 1 + 3) // 4
    HALF = 2 * KPR
    if 3 * LCD >= HALF and (3 * LCD - HALF) % 2 == 1:
        KS3D = 3 * KPR
        s, r, hh, j = _grid(KS3D)
        ky, q = s // KPR, s % KPR
        rc = HALF * hh + 2 * q + (j >> 2)
        jj = j & 3
        tap, l = 3 * ky + np.minimum(rc, 3 * LCD - 1) // LCD, 4 * (rc % LCD) + jj
        rf = np.minimum(r, F - 1)
        w3d = _sel((rc < 3 * LCD) & (l < L) & (r < F), o["w3"] + (rf * L + np.minimum(l, L - 1)) * 9 + np.minimum(tap, 8), Z)
        w3d = _sel((rc == 3 * LCD) & (ky == 2) & (jj == 0) & (r < F), o["b3"] + rf, w3d)
    else:
        KS3D = 0

    w = np.concatenate([fwd["w"]] + [a.reshape(-1) for a in (w3t, w2t, w1t, ident, w2n)] + ([w3d.reshape(-1)] if w3d is not None else []))
    cin = [fwd["cinit"]]
    if not g.fold_b1:
        ee = np.arange(g.NET * 32)
        cin.append(np.where(ee < E, o["b1"] + np.minimum(ee, E - 1), Z))
    sec, off = {}, 0
    for name, n in (("W1", g.NET * g.KS1), ("W2", g.KS2), ("W3", g.KS3), ("W3T", KS3B), ("W2T", g.NET * 2),
                    ("W1T", g.KS2), ("ID", KSI), ("W2N", g.NET * 2), ("W3D", KS3D)):
        sec[name] = off
        off += n
    assert off * 512 == w.size
    return dict(w=w, cinit=np.concatenate(cin), sec=sec, nfrag=off, geom=g, KS3B=KS3B, KSI=KSI, KS3D=KS3D)


# ---- accumulator-layout slabs written by the weight-gradient kernels -> canonical gradients ----
def _acc_pos(tile, row, col):
    """offset of element (row, col) of 32x32 accumulator tile #tile in a [tile][reg 16][lane 64] slab
    (lane-contiguous so the in-workgroup LDS reduction is bank-conflict free)"""
    hh = (row >> 2) & 1
    i = (row & 3) + 4 * (row >> 3)
    return (tile * 16 + i) * 64 + col + 32 * hh


@lru_cache(maxsize=None)
def block_grad_tables(F: int, E: int, L: int):
    """Gather tables: canonical gradient vector (same layout as BlockGeom.off, without the two
    constants) = slab[idx].  Slab A (wgrad12): dW1T tiles [f rows, e cols] x NET, then dW2 tiles
    [l rows, e cols] x NET, then db1[NET*32], db2[32].  Slab B (wgrad3): 9 tiles [l rows, f cols]
    indexed by read offset u (tap = 8 - u); the ones channel l = L at the centre carries db3."""
    g = BlockGeom(F, E, L)
    NET = g.NET
    e, f = np.meshgrid(np.arange(E), np.arange(F), indexing="ij")
    w1 = _acc_pos(e // 32, f, e % 32)
    l, e2 = np.meshgrid(np.arange(L), np.arange(E), indexing="ij")
    w2 = _acc_pos(NET + e2 // 32, l, e2 % 32)
    base = 2 * NET * 1024
    b1 = base + np.arange(E)
    b2 = base + NET * 32 + np.arange(L)
    slab_a = base + NET * 32 + 32
    f3, l3, tap = np.meshgrid(np.arange(F), np.arange(L), np.arange(9), indexing="ij")
    w3 = _acc_pos(8 - tap, l3, f3)
    b3 = _acc_pos(np.full(F, 4), np.full(F, L), np.arange(F))
    return dict(a=np.concatenate([w1.reshape(-1), w2.reshape(-1), b1, b2]), a_size=slab_a,
                a_order=("w1", "w2", "b1", "b2"),
                b=np.concatenate([w3.reshape(-1), b3]), b_size=9 * 1024, b_order=("w3", "b3"), geom=g)


# =====================================================================================
# head (3x3, 3 -> F) and tail (3x3 F -> 3r^2) + skip (5x5, 3 -> 3r^2) + PixelShuffle(r)
# reference: models/basic_wdsr_b.py:32-42 (head), :55-64 (tail), :66-78 (skip), :80-92 (shuffle, mean)
# =====================================================================================
# The LR image is staged in LDS as [pixel][4]: 3 colours minus the mean, and a ones channel that
# carries the bias.  A fragment chunk (8 elements) is two horizontally adjacent pixels.

@dataclass(frozen=True)
class EndsGeom:
    F: int
    R: int                      # upscale factor

    @property
    def CO(self):               # conv channels before the shuffle
        return 3 * self.R * self.R

    @property
    def NT(self):
        return (self.CO + 31) // 32

    @property
    def COP(self):              # dconv channels held in LDS
        return (self.CO + 7) // 8 * 8

    @property
    def CC(self):
        return self.COP // 8

    @property
    def FC(self):
        return self.F // 8

    @property
    def KST(self):              # k-steps of the fused tail+skip product
        return (9 * self.FC + 15 + 1) // 2

    @property
    def KSTB(self):             # k-steps of tail backward-data
        return (9 * self.CC + 1) // 2

    @property
    def tail_off(self):         # canonical tail source: wt | ws | btot | 0 | 1
        CO, F = self.CO, self.F
        o, d = 0, {}
        for name, n in (("wt", CO * F * 9), ("ws", CO * 3 * 25), ("b", CO), ("zero", 1), ("one", 1)):
            d[name] = o
            o += n
        d["size"] = o
        return d

    @property
    def head_off(self):         # canonical head source: wh | bh | 0 | 1
        F = self.F
        return {"wh": 0, "b": F * 27, "zero": F * 27 + F, "one": F * 27 + F + 1, "size": F * 27 + F + 2}


@lru_cache(maxsize=None)
def ends_tables(F: int, R: int):
    g = EndsGeom(F, R)
    CO, FC, CC = g.CO, g.FC, g.CC
    ot, oh = g.tail_off, g.head_off

    # ---- head: rows f, chunks q = 2s+hh < 6: ky = q//2, m = q%2 -> kx = 2m + (j>>2), ci = j&3 ----
    s, r, hh, j = _grid(3)
    q = 2 * s + hh
    ky, m = q // 2, q % 2
    kx, ci = 2 * m + (j >> 2), j & 3
    ok = (r < F) & (kx < 3)
    rf = np.minimum(r, F - 1)
    head = np.full(s.shape, oh["zero"], dtype=np.int64)
    head = _sel(ok & (ci < 3), oh["wh"] + ((rf * 3 + np.minimum(ci, 2)) * 3 + ky) * 3 + np.minimum(kx, 2), head)
    head = _sel(ok & (ci == 3) & (ky == 1) & (kx == 1), oh["b"] + rf, head)

    # ---- tail + skip: rows ch (NT tiles); chunks q < 9 FC: tail (tap, 8 feature channels);
    #      then 15 image chunks: ky = q'//3, m = q'%3 -> kx = 2m + (j>>2), ci = j&3 ----
    s, r, hh, j = _grid(g.NT * g.KST)
    ti, ks = s // g.KST, s % g.KST
    ch = 32 * ti + r
    chc = np.minimum(ch, CO - 1)
    q = 2 * ks + hh
    tail = np.full(s.shape, ot["zero"], dtype=np.int64)
    is_t = q < 9 * FC
    tap, c = q // FC, q % FC
    f = 8 * c + j
    tail = _sel(is_t & (ch < CO), ot["wt"] + (chc * F + np.minimum(f, F - 1)) * 9 + np.minimum(tap, 8), tail)
    qs = q - 9 * FC
    ky, m = qs // 3, qs % 3
    kx, ci = 2 * m + (j >> 2), j & 3
    is_s = (~is_t) & (qs < 15) & (ch < CO) & (kx < 5)
    tail = _sel(is_s & (ci < 3),
                ot["ws"] + ((chc * 3 + np.minimum(ci, 2)) * 5 + np.clip(ky, 0, 4)) * 5 + np.minimum(kx, 4), tail)
    tail = _sel(is_s & (ci == 3) & (ky == 2) & (kx == 2), ot["b"] + chc, tail)

    # ---- tail backward-data: rows f, chunks q < 9 CC: read offset u = q // CC, channels 8 (q%CC)+j;
    #      dfeat[px, f] = sum_{u, ch} Wt[ch, f, 8-u] dconv[px + u - 1, ch] ----
    s, r, hh, j = _grid(g.KSTB)
    q = 2 * s + hh
    u, c = q // CC, q % CC
    ch = 8 * c + j
    tbd = _sel((q < 9 * CC) & (r < F) & (ch < CO),
               ot["wt"] + (np.minimum(ch, CO - 1) * F + np.minimum(r, F - 1)) * 9 + (8 - np.minimum(u, 8)),
               ot["zero"])

    tail_w = np.concatenate([tail.reshape(-1), tbd.reshape(-1)])
    return dict(head=head.reshape(-1), tail=tail_w, geom=g, n_head=3, n_tail=g.NT * g.KST, n_tbd=g.KSTB)


@lru_cache(maxsize=None)
def ends_grad_tables(F: int, R: int):
    """Gather tables for the slabs of the tail / head weight-gradient kernels.
    Tail slab: tiles [(tap * NT + ti)] (rows ch, cols f) for the 9 taps, then [(ky * NT + ti)] (rows ch,
    cols (kx, ci) = 4 kx + ci) for the 5 skip rows.  Head slab: 3 tiles [ky] (rows f, cols 4 kx + ci).
    The bias is the ones-channel column at the centre tap."""
    g = EndsGeom(F, R)
    CO, NT = g.CO, g.NT
    ch, f, tap = np.meshgrid(np.arange(CO), np.arange(F), np.arange(9), indexing="ij")
    wt = _acc_pos(tap * NT + ch // 32, ch % 32, f)
    base = 9 * NT
    ch, ci, ky, kx = np.meshgrid(np.arange(CO), np.arange(3), np.arange(5), np.arange(5), indexing="ij")
    ws = _acc_pos(base + ky * NT + ch // 32, ch % 32, 4 * kx + ci)
    chb = np.arange(CO)
    bt = _acc_pos(base + 2 * NT + chb // 32, chb % 32, 4 * 2 + 3)
    tail = np.concatenate([wt.reshape(-1), ws.reshape(-1), bt])
    fh, ci, ky, kx = np.meshgrid(np.arange(F), np.arange(3), np.arange(3), np.arange(3), indexing="ij")
    wh = _acc_pos(ky, fh, 4 * kx + ci)
    bh = _acc_pos(np.full(F, 1), np.arange(F), np.full(F, 4 * 1 + 3))
    head = np.concatenate([wh.reshape(-1), bh])
    return dict(tail=tail, tail_size=(9 * NT + 5 * NT) * 1024, head=head, head_size=3 * 1024, geom=g)


# =====================================================================================
# BasicVSR trunk 3x3 convs (models/basicvsr_arch.py:108-147): CI_real in {24, 27} -> 24, plain bias
# canonical source: w (24, CI_real, 3, 3) | b (24) | 0 | 1
# =====================================================================================
@lru_cache(maxsize=None)
def c3_tables(ci_real: int):
    CO = 24
    n_w = CO * ci_real * 9
    o = {"w": 0, "b": n_w, "zero": n_w + CO, "one": n_w + CO + 1, "size": n_w + CO + 2}
    # forward: rows co, chunks q = 2s+hh < 36: tap = q // 4, c = q % 4, ci = 8c + j (ones channel = ci_real)
    s, r, hh, j = _grid(18)
    q = 2 * s + hh
    tap, c = q // 4, q % 4
    ci = 8 * c + j
    rc = np.minimum(r, CO - 1)
    fw = np.full(s.shape, o["zero"], dtype=np.int64)
    fw = _sel((r < CO) & (ci < ci_real), o["w"] + (rc * ci_real + np.minimum(ci, ci_real - 1)) * 9 + tap, fw)
    fw = _sel((r < CO) & (ci == ci_real) & (tap == 4), o["b"] + rc, fw)
    # backward-data: rows ci, chunks q < 27: u = q // 3, c = q % 3, co = 8c + j, weight tap 8 - u
    s, r, hh, j = _grid(14)
    q = 2 * s + hh
    u, c = q // 3, q % 3
    co = 8 * c + j
    bw = _sel((q < 27) & (r < ci_real),
              o["w"] + (co * ci_real + np.minimum(r, ci_real - 1)) * 9 + (8 - np.minimum(u, 8)), o["zero"])
    w = np.concatenate([fw.reshape(-1), bw.reshape(-1)])
    # gradient gather: dW[co, ci, tap] = tile[8 - tap](row ci, col co); db[co] = tile[4](row ci_real, col co)
    co_, ci_, tap_ = np.meshgrid(np.arange(CO), np.arange(ci_real), np.arange(9), indexing="ij")
    gw = _acc_pos(8 - tap_, ci_, co_)
    gb = _acc_pos(np.full(CO, 4), np.full(CO, ci_real), np.arange(CO))
    return dict(w=w, grad=np.concatenate([gw.reshape(-1), gb]), off=o)


# =====================================================================================
# NAS supernet block (models/wdsr_b.py:375-496): canonical source per block
#   wdw3 (F,9) | wdw5 (F,25) | wdw7 (F,49) | bdw (3,F) | wpw (3,F,F) | bpw (3,F) | mg (F) | ms (F) | m1 (F) | 0 | 1
# =====================================================================================
@lru_cache(maxsize=None)
def nas_tables(F: int):
    o, off = {}, 0
    for name, n in (("wdw3", F * 9), ("wdw5", F * 25), ("wdw7", F * 49), ("bdw", 3 * F), ("wpw", 3 * F * F),
                    ("bpw", 3 * F), ("mg", F), ("ms", F), ("m1", F), ("zero", 1), ("one", 1)):
        o[name] = off
        off += n
    o["size"] = off
    Z = o["zero"]
    ch = np.arange(32)
    chc = np.minimum(ch, F - 1)
    dwp = []
    for key, kk in (("wdw3", 9), ("wdw5", 25), ("wdw7", 49)):
        tap = np.arange(kk).reshape(-1, 1)
        dwp.append(np.where(ch < F, o[key] + chc * kk + tap, Z).reshape(-1))       # [tap][32]
    for k in range(3):
        dwp.append(np.where(ch < F, o["bdw"] + k * F + chc, Z))
    for key in ("m1", "mg", "ms"):
        dwp.append(np.where(ch < F, o[key] + chc, Z))
    dwp = np.concatenate(dwp)
    assert dwp.size == (83 + 3 + 3) * 32
    # pointwise fragments: forward (rows co, k = ci natural), backward (rows ci, k = co chained)
    fw, bw = [], []
    for k in range(3):
        s, r, hh, j = _grid(2)
        ci = k_natural(s, hh, j)
        fw.append(_sel((r < F) & (ci < F), o["wpw"] + (k * F + np.minimum(r, F - 1)) * F + np.minimum(ci, F - 1), Z))
        co = k_chained(s, hh, j)
        bw.append(_sel((r < F) & (co < F), o["wpw"] + (k * F + np.minimum(co, F - 1)) * F + np.minimum(r, F - 1), Z))
    frags = np.concatenate([a.reshape(-1) for a in fw + bw])
    tabs = []
    for k in range(3):
        rows = np.where(ch < F, o["bpw"] + k * F + chc, Z)
        tabs.append(cinit_index(rows, Z))
    for key in ("ms", "mg"):
        tabs.append(cinit_index(np.where(ch < F, o[key] + chc, Z), Z))
    tabs = np.concatenate(tabs)
    # gradient gathers
    co_, ci_ = np.meshgrid(np.arange(F), np.arange(F), indexing="ij")
    g_wpw = np.stack([k * 1088 + _acc_pos(0, co_, ci_) for k in range(3)]).reshape(-1)
    g_bpw = np.stack([k * 1088 + 1024 + np.arange(F) for k in range(3)]).reshape(-1)
    g_r = np.stack([k * 1088 + 1056 + np.arange(F) for k in range(3)]).reshape(-1)
    tb = (0, 9, 34)
    g_wdw = [np.stack([(tb[i] + np.arange(kk)) * 32 + c for c in range(F)]).reshape(-1) for i, kk in enumerate((9, 25, 49))]
    g_bdw = np.stack([(83 + k) * 32 + np.arange(F) for k in range(3)]).reshape(-1)
    return dict(off=o, dwp=dwp, frags=frags, tabs=tabs, g_wpw=g_wpw, g_bpw=g_bpw, g_r=g_r, g_wdw=g_wdw, g_bdw=g_bdw,
                g_sA=86 * 32 + np.arange(F), g_sB=87 * 32 + np.arange(F), pw_slab=3 * 1088 + 4, dw_slab=88 * 32, sxy=3 * 1088)


def nas_prep_tables(F: int, nb: int, layout, blocks=None):
    """Tables for the native parameter plumbing of the supernet body (csrc/wdsr_prep.h through sr_param_pack / sr_param_grads).
    `layout`: NAS_MODEL._layout = (key suffix, offset in the flat parameter, count, (nb, ...) shape) per kind, stacked over
    blocks.  Source row of block b = nas_tables(F)['off'] columns; gradient row = the same columns followed by the extra sums
    r[3][F] | sxy | sA[F] | sB[F] the mask / gate gradients are made of.
    `blocks`: the blocks that run, in order (eval drops skipped blocks; default all `nb`): row j of the source / gradient
    buffers belongs to block blocks[j].
    Returns chan_tab int32[nb * 6F][4] = {v_off, g_off, K, dst}, bias_tab int32[nb * 6F][3] = {flat index, -1, dst},
    the two slab scatters (sidx, dst) and the row sizes."""
    t = nas_tables(F)
    o = t["off"]
    size = o["size"]
    lay = {name: (off, n, shape) for name, off, n, shape in layout}
    ex = dict(r=size, sxy=size + 3 * F, sA=size + 3 * F + 1, sB=size + 4 * F + 1)
    ds = size + 5 * F + 1
    chan, bias = [], []
    c = np.arange(F)
    for row, b in enumerate(range(nb) if blocks is None else blocks):
        for ki, k in enumerate((3, 5, 7)):
            for j, (K, dst0) in ((0, (k * k, o[f"wdw{k}"])), (2, (F, o["wpw"] + ki * F * F))):
                ov, _n, _s = lay[f"body.{k}.0.body.{j}.weight_v"]
                og, _n, _s = lay[f"body.{k}.0.body.{j}.weight_g"]
                chan.append(np.stack([ov + (b * F + c) * K, og + b * F + c, np.full(F, K), row * size + dst0 + c * K], axis=1))
                ob, _n, _s = lay[f"body.{k}.0.body.{j}.bias"]
                bias.append(np.stack([ob + b * F + c, np.full(F, -1), row * size + o["bdw" if j == 0 else "bpw"] + ki * F + c], axis=1))
    chan_tab = np.concatenate(chan).astype(np.int32)
    bias_tab = np.concatenate(bias).astype(np.int32)
    # the weight-norm backward reads d(src) at the same dst offsets but in rows of `ds` columns
    chan_bwd, bias_bwd = chan_tab.copy(), bias_tab.copy()
    chan_bwd[:, 3] = (chan_tab[:, 3] // size) * ds + chan_tab[:, 3] % size
    bias_bwd[:, 2] = (bias_tab[:, 2] // size) * ds + bias_tab[:, 2] % size
    pw_sidx = np.concatenate([t["g_wpw"], t["g_bpw"], t["g_r"], [t["sxy"]]])
    pw_dst = np.concatenate([o["wpw"] + np.arange(3 * F * F), o["bpw"] + np.arange(3 * F), ex["r"] + np.arange(3 * F), [ex["sxy"]]])
    dw_sidx = np.concatenate(list(t["g_wdw"]) + [t["g_bdw"], t["g_sA"], t["g_sB"]])
    dw_dst = np.concatenate([o["wdw3"] + np.arange(9 * F), o["wdw5"] + np.arange(25 * F), o["wdw7"] + np.arange(49 * F),
                             o["bdw"] + np.arange(3 * F), ex["sA"] + np.arange(F), ex["sB"] + np.arange(F)])
    # pairs in slab order: neighbouring threads of the scatter kernel then read neighbouring slab entries
    op, od = np.argsort(pw_sidx, kind="stable"), np.argsort(dw_sidx, kind="stable")
    pw_sidx, pw_dst, dw_sidx, dw_dst = pw_sidx[op], pw_dst[op], dw_sidx[od], dw_dst[od]
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    return dict(chan_tab=chan_tab, bias_tab=bias_tab, chan_bwd=i32(chan_bwd), bias_bwd=i32(bias_bwd), pw_sidx=i32(pw_sidx),
                pw_dst=i32(pw_dst), dw_sidx=i32(dw_sidx), dw_dst=i32(dw_dst), size=size, ds=ds, extra=ex, off=o)



# =====================================================================================
# SPyNet's 7x7 convolutions (reference: models/spynet_arch.py:17-22) -> csrc/spynet_conv.h
# =====================================================================================
@lru_cache(maxsize=None)
def conv7_tables(cin: int, cout: int):
    """Gather table for one 7x7 layer: packed = src[idx] with src = weight (cout, cin, 7, 7).reshape(-1) | 0.0, laid out as the
    kernel consumes it: [ky][k-step][32-row tile][lane][8].  k-step s (cin >= 16) = (tap kx = s // (cin/16), channels
    16 (s % (cin/16)) + 8 hh + j); cin = 8: two adjacent taps per k-step, kx = 2 s + hh (kx = 7: zero), channel j."""
    assert cin == 8 or cin % 16 == 0
    mt = (cout + 31) // 32
    kpr = 4 if cin == 8 else 7 * (cin // 16)
    zero = cout * cin * 49
    ky = np.arange(7).reshape(7, 1, 1, 1, 1)
    s = np.arange(kpr).reshape(1, kpr, 1, 1, 1)
    m = np.arange(mt).reshape(1, 1, mt, 1, 1)
    lane = np.arange(64).reshape(1, 1, 1, 64, 1)
    j = np.arange(8).reshape(1, 1, 1, 1, 8)
    ky, s, m, lane, j = np.broadcast_arrays(ky, s, m, lane, j)
    r, hh = lane & 31, lane >> 5
    co = 32 * m + r
    if cin == 8:
        kx, ci = 2 * s + hh, j
    else:
        cc = cin // 16
        kx, ci = s // cc, 16 * (s % cc) + 8 * hh + j
    ok = (co < cout) & (kx < 7)
    idx = np.where(ok, ((np.minimum(co, cout - 1) * cin + ci) * 7 + ky) * 7 + np.minimum(kx, 6), zero)
    return dict(idx=idx.reshape(-1).astype(np.int64), mt=mt, kpr=kpr)
