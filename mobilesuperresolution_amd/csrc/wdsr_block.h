// Fused WDSR-B residual block kernels (gfx950).  Reference op being replaced:
// Block.forward, models/basic_wdsr_b.py:108-144 (wn-1x1 F->E, ReLU, wn-1x1 E->L, wn-3x3 L->F, + x)
// and its autograd backward.  Activations are NHWC; the E-wide intermediate never leaves registers,
// the L-wide one never leaves LDS.
#pragma once
#include "sr_common.h"

template <int F_, int E_, int L_>
struct BlockCfg {
  static constexpr int F = F_, E = E_, L = L_;
  static constexpr bool FOLD_B1 = (F % 16 != 0);            // conv1 bias rides on the spare ones channel
  static constexpr int KX = FOLD_B1 ? (F / 16 + 1) * 16 : F; // x channels per pixel in LDS
  static constexpr int KS1 = KX / 16;
  static constexpr int NET = (E + 31) / 32;
  static constexpr int KS2 = (E + 15) / 16;
  static constexpr int LP = ((L + 1 + 7) / 8) * 8;           // t channels in LDS (L + ones channel)
  static constexpr int CPT = LP / 8;
  static constexpr int FC = F / 8;
  static constexpr int KS3 = (9 * CPT + FC + 1) / 2;
  // spatial tile of one workgroup
  static constexpr int TH = 12, TW = 24;
  static constexpr int HW = TW + 2, HH = TH + 2;
  static constexpr int NPXH = HW * HH;                       // halo'd pixels
  static constexpr int NPXH_PAD = (NPXH + 31) / 32 * 32;
  static constexpr int NPT_H = NPXH_PAD / 32;                // 32-pixel tiles over the halo'd region
  static constexpr int NPT_O = (TH / 4) * (TW / 8);          // 4x8-pixel output tiles
  // packed weight blob (fragments of 512 elements)
  static constexpr int W1_OFF = 0, W2_OFF = NET * KS1, W3_OFF = W2_OFF + KS2, NFRAG_FWD = W3_OFF + KS3;
  static constexpr int CINIT_FWD = 32 + (FOLD_B1 ? 0 : NET * 32);
};

// stage the halo'd x tile [NPXH_PAD][KX] into LDS: zero outside the image, ones channel at index F
template <typename T, typename C>
SR_DEV void stage_x_halo(T* Xs, const T* __restrict__ xin, int H, int W, int ty0, int tx0, int tid) {
  typedef typename FragOf<T>::type FragT;
  constexpr int CHX = C::KX / 8;
  for (int idx = tid; idx < C::NPXH_PAD * CHX; idx += 256) {
    const int hp = idx / CHX, c = idx - hp * CHX;
    FragT v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (T)0.f;
    if (c < C::FC) {
      if (hp < C::NPXH) {
        const int hy = hp / C::HW, hx = hp - hy * C::HW;
        const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
        if (Y >= 0 && Y < H && X >= 0 && X < W)
          v = *reinterpret_cast<const FragT*>(xin + ((size_t)Y * W + X) * C::F + c * 8);
      }
    } else if (C::FOLD_B1 && c == C::FC) {
      v[0] = (T)1.f;
    }
    *reinterpret_cast<FragT*>(Xs + hp * C::KX + c * 8) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// forward:  y = conv3x3(W3, conv1x1(W2, relu(conv1x1(W1, x) + b1)) + b2) + b3 + x
// grid = (tiles_y * tiles_x, N), block = 256 (4 waves, one per SIMD)
// ---------------------------------------------------------------------------------------------
template <typename T, int F, int E, int L>
__global__ __launch_bounds__(256) void wdsr_block_fwd_kernel(
    const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ wblob,
    const float* __restrict__ cinit, int H, int W, int tiles_x) {
  typedef BlockCfg<F, E, L> C;
  typedef typename FragOf<T>::type FragT;
  typedef typename FragOf<T>::half_type HalfT;
  // one LDS array (x tile, then t tile) so that every fragment address is an offset into it
  __shared__ __attribute__((aligned(16))) T smem[C::NPXH_PAD * (C::KX + C::LP)];
  T* const Xs = smem;
  T* const Ts = smem + C::NPXH_PAD * C::KX;
  constexpr int TS0 = C::NPXH_PAD * C::KX;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const T* xin = x + (size_t)n * H * W * F;

  stage_x_halo<T, C>(Xs, xin, H, W, ty0, tx0, tid);
  __syncthreads();

  // ---- phase A: t = W2 relu(W1 x + b1) + b2 on every halo'd pixel (zero outside the image) ----
  for (int pt = wave; pt < C::NPT_H; pt += 4) {
    const int hp = pt * 32 + r;
    FragT xb[C::KS1];
#pragma unroll
    for (int s = 0; s < C::KS1; ++s) xb[s] = lds_chunk<T>(Xs, hp * C::KX + (2 * s + hh) * 8);
    f32x16 tacc = load_cinit(cinit, hh);
#pragma unroll
    for (int et = 0; et < C::NET; ++et) {
      f32x16 hacc = C::FOLD_B1 ? zero16() : load_cinit(cinit + 32 + et * 32, hh);
#pragma unroll
      for (int s = 0; s < C::KS1; ++s)
        hacc = mma16<T>(load_wfrag<T>(wblob, C::W1_OFF + et * C::KS1 + s, lane), xb[s], hacc);
#pragma unroll
      for (int i = 0; i < 16; ++i) hacc[i] = fmaxf(hacc[i], 0.f);
      if (2 * et < C::KS2)
        tacc = mma16<T>(load_wfrag<T>(wblob, C::W2_OFF + 2 * et, lane), acc_to_frag<T, 0>(hacc), tacc);
      if (2 * et + 1 < C::KS2)
        tacc = mma16<T>(load_wfrag<T>(wblob, C::W2_OFF + 2 * et + 1, lane), acc_to_frag<T, 1>(hacc), tacc);
    }
    bool valid = false;
    if (hp < C::NPXH) {
      const int hy = hp / C::HW, hx = hp - hy * C::HW;
      const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
      valid = (Y >= 0 && Y < H && X >= 0 && X < W);
    }
#pragma unroll
    for (int g = 0; g < C::CPT; ++g) {
      HalfT v = acc_group<T>(tacc, g);
      if (!valid) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (T)0.f;
      }
      *reinterpret_cast<HalfT*>(Ts + hp * C::LP + g * 8 + hh * 4) = v;
    }
  }
  __syncthreads();

  // ---- phase B: y = sum_taps W3_tap t(shifted) + b3 (ones channel) + x (identity chunks) ----
  for (int ot = wave; ot < C::NPT_O; ot += 4) {
    const int oy = (ot / (C::TW / 8)) * 4 + (r >> 3), ox = (ot % (C::TW / 8)) * 8 + (r & 7);
    const int hbase = oy * C::HW + ox;
    f32x16 oacc = zero16();
#pragma unroll
    for (int s = 0; s < C::KS3; ++s) {
      const int q = 2 * s + hh;
      int off;
      if (q < 9 * C::CPT) {
        const int tap = q / C::CPT, c = q - tap * C::CPT;
        off = TS0 + (hbase + (tap / 3) * C::HW + (tap % 3)) * C::LP + c * 8;
      } else {
        int c = q - 9 * C::CPT;
        if (c >= C::FC) c = 0;  // weights there are zero; any finite data will do
        off = (hbase + C::HW + 1) * C::KX + c * 8;
      }
      FragT b = lds_chunk<T>(smem, off);
      oacc = mma16<T>(load_wfrag<T>(wblob, C::W3_OFF + s, lane), b, oacc);
    }
    const int Y = ty0 + oy, X = tx0 + ox;
    if (Y < H && X < W) {
      T* yo = y + (((size_t)n * H + Y) * W + X) * F;
#pragma unroll
      for (int g = 0; g < C::FC; ++g) *reinterpret_cast<HalfT*>(yo + g * 8 + hh * 4) = acc_group<T>(oacc, g);
    }
  }
}
