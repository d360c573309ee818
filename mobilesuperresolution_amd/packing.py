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
