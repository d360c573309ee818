// NAS supernet residual block (gfx950).  Reference ops replaced: Split_Block.forward_body with
// Conv_sep(seperate=True) branches and the BinaryConv2d masks, models/wdsr_b.py:375-496, models/ops.py:7-43,
// inside MyAggregationLayer.forward (:517-546) and NAS_MODEL.forward's per-block global mask (:116):
//
//   x  = mg * yin                      (global channel mask, BinaryConv2d, least_channel 8)
//   x1 = ms * x                        (block split mask, least_channel 0)
//   S  = sum_k p_k relu(pw_k(relu(dw_k(x1) + bd_k)) + bp_k),  k in {3,5,7},  p = softmax(alpha)
//   y  = x + beta2 * ms * S            (beta2 in {0,1}: the hard skip/keep gate; beta1 + beta2 = 1)
//
// Four kernels: depthwise forward (VALU; the three stencils share one LDS halo-3 tile, weights arrive
// through wave-uniform scalar loads), pointwise forward (MFMA, fused mix + masks), and their backward
// halves.  Intermediates v_k = relu(dw_k(x1)) and gz_k = d(loss)/d(dw_k output) live in HBM.
// Float parameter table `dwp` (channels padded to 32): W3 | W5 | W7 | BD[3] | M1 = mg*ms | MG | MS.
#pragma once
#include "wdsr_block.h"

template <int F_> struct NasCfg {
  static constexpr int F = F_, FC = F / 8;
  static constexpr int TH = 12, TW = 24, NPXC = TH * TW, NPT_O = (TH / 4) * (TW / 8);
  static constexpr int PW = TW + 6, PH = TH + 6, NP3 = PW * PH;          // tile with a 3-pixel halo
  static constexpr int P3_ELEMS = (NP3 + 2) * F;
  static constexpr int W3 = 0, W5 = 9 * 32, W7 = W5 + 25 * 32, BD = W7 + 49 * 32, M1 = BD + 3 * 32, MG = M1 + 32,
                       MS = MG + 32, DWP = MS + 32;
  static constexpr int NGRP = (NPXC + 63) / 64;                          // 64-pixel groups of the core tile
  static constexpr int NITEM = NGRP * FC;                                // (pixel group, 8-channel chunk) items
  static constexpr int VT_ELEMS = (NPXC + 1) * 32;                       // core tile, 32 channels per pixel
  // pw backward slab per branch k: tile [co rows, ci cols] | db[32] | r[32]; then one scalar
  static constexpr int PWB_K = 1024 + 64, PWB_SLAB = 3 * PWB_K + 4;
  // dw backward slab: dW[83 taps][32] | db[3][32] | sA[32] | sB[32]
  static constexpr int DWB_SLAB = (83 + 3 + 2) * 32;
};

SR_DEV int nas_woff(int k) { return k == 0 ? 0 : (k == 1 ? 9 * 32 : 9 * 32 + 25 * 32); }

// halo-3 tile [NP3 + 2][F] of `src` scaled per channel by scale[c] (nullptr: 1), zero outside the image
template <typename T, typename C, int NTHREADS>
SR_DEV void nas_stage_halo3(T* dst, const T* __restrict__ src, const float* __restrict__ scale, int H, int W, int ty0,
                            int tx0, int tid) {
  typedef typename FragOf<T>::type FragT;
  constexpr int TOTAL = (C::NP3 + 2) * C::FC;
  for (int idx = tid; idx < TOTAL; idx += NTHREADS) {
    const int hp = idx / C::FC, c = idx - hp * C::FC;
    FragT v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (T)0.f;
    if (hp < C::NP3) {
      const int hy = hp / C::PW, hx = hp - hy * C::PW;
      const int Y = ty0 - 3 + hy, X = tx0 - 3 + hx;
      if (Y >= 0 && Y < H && X >= 0 && X < W) {
        v = *reinterpret_cast<const FragT*>(src + ((size_t)Y * W + X) * C::F + c * 8);
        if (scale) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (T)((float)v[j] * scale[c * 8 + j]);
        }
      }
    }
    *reinterpret_cast<FragT*>(dst + hp * C::F + c * 8) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// depthwise forward: V_k = relu(dw_k(m1 * yin) + bd_k), k = 3, 5, 7.  grid = (tiles, N), 8 waves.
// A wave item = 64 pixels x one 8-channel chunk, so the stencil weights are wave-uniform (s_load).
// ---------------------------------------------------------------------------------------------
template <typename T, int F>
__global__ __launch_bounds__(512) void nas_dw_fwd_kernel(const T* __restrict__ yin, T* __restrict__ V,
                                                         const float* __restrict__ dwp, int H, int W, int tiles_x,
                                                         long vstride) {
  typedef NasCfg<F> C;
  typedef typename FragOf<T>::type FragT;
  __shared__ __attribute__((aligned(16))) T X1[C::P3_ELEMS];
  // stencil weights and biases in LDS: the per-tap wave-uniform s_load from HBM was the whole cost of this kernel
  // (one dependent ~1 us round trip per tap); all lanes of a wave read the same address (LDS broadcast)
  __shared__ __attribute__((aligned(16))) float WS[C::M1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * F;
  for (int i = tid; i < C::M1; i += 512) WS[i] = dwp[i];
  nas_stage_halo3<T, C, 512>(X1, yin + img, dwp + C::M1, H, W, ty0, tx0, tid);
  __syncthreads();
  for (int item0 = wave; item0 < C::NITEM; item0 += 8) {
    const int item = __builtin_amdgcn_readfirstlane(item0);
    const int g = item / C::FC, c = item - g * C::FC;
    const int px = g * 64 + lane;
    const bool valid = px < C::NPXC;
    const int oy = valid ? px / C::TW : 0, ox = valid ? px % C::TW : 0;
    const float* w = WS + c * 8;                                    // wave-uniform
    float z3[8], z5[8], z7[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { z3[j] = w[C::BD + j]; z5[j] = w[C::BD + 32 + j]; z7[j] = w[C::BD + 64 + j]; }
    // one exact pass per stencil (no masked multiply-adds): rows one at a time, the taps of a row unrolled
    auto stencil = [&](float (&z)[8], int ks, int wbase) {
      const int off = 3 - ks / 2;
#pragma unroll 1
      for (int ty = 0; ty < ks; ++ty) {
        const T* xr = X1 + ((oy + off + ty) * C::PW + ox + off) * F + c * 8;
        const float* wr = w + wbase + ty * ks * 32;
#pragma unroll
        for (int tx = 0; tx < 7; ++tx) {
          if (tx < ks) {
            const FragT v = *reinterpret_cast<const FragT*>(xr + tx * F);
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] += wr[tx * 32 + j] * (float)v[j];
          }
        }
      }
    };
    stencil(z7, 7, C::W7);
    stencil(z5, 5, C::W5);
    stencil(z3, 3, C::W3);
    const int Y = ty0 + oy, X = tx0 + ox;
    if (valid && Y < H && X < W) {
      const size_t o = img + ((size_t)Y * W + X) * F + c * 8;
      FragT a, b, d;
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] = (T)fmaxf(z3[j], 0.f); b[j] = (T)fmaxf(z5[j], 0.f); d[j] = (T)fmaxf(z7[j], 0.f); }
      *reinterpret_cast<FragT*>(V + o) = a;
      *reinterpret_cast<FragT*>(V + vstride + o) = b;
      *reinterpret_cast<FragT*>(V + 2 * vstride + o) = d;
    }
  }
}

// core tile of one V_k as [NPXC + 1][32] (channels >= F zero)
template <typename T, typename C, int NTHREADS>
SR_DEV void nas_stage_vt(T* VT, const T* __restrict__ v, int H, int W, int ty0, int tx0, int tid) {
  typedef typename FragOf<T>::type FragT;
  for (int idx = tid; idx < (C::NPXC + 1) * 4; idx += NTHREADS) {
    const int pc = idx >> 2, c = idx & 3;
    FragT f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (T)0.f;
    if (pc < C::NPXC && c < C::FC) {
      const int Y = ty0 + pc / C::TW, X = tx0 + pc % C::TW;
      if (Y < H && X < W) f = *reinterpret_cast<const FragT*>(v + ((size_t)Y * W + X) * C::F + c * 8);
    }
    *reinterpret_cast<FragT*>(VT + idx * 8) = f;
  }
}

// core tiles of the three V_k as [3][NPXC + 1][32], loads issued in batches before their stores (a load / store loop
// compiles to one memory round trip per chunk and thread)
template <typename T, typename C, int NTHREADS>
SR_DEV void nas_stage_vt3(T* VT, const T* __restrict__ v, long vstride, int H, int W, int ty0, int tx0, int tid) {
  typedef typename FragOf<T>::type FragT;
  constexpr int PER = (C::NPXC + 1) * 4, TOTAL = 3 * PER, IT = (TOTAL + NTHREADS - 1) / NTHREADS, B = sizeof(T) == 2 ? 3 : 4;
#pragma unroll 1
  for (int base = 0; base < IT; base += B) {
    FragT f[B];
#pragma unroll
    for (int it = 0; it < B; ++it) {
      const int idx = tid + (base + it) * NTHREADS;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[it][j] = (T)0.f;
      if (base + it < IT && idx < TOTAL) {
        const int kk = idx / PER, q = idx - kk * PER, pc = q >> 2, c = q & 3;
        if (pc < C::NPXC && c < C::FC) {
          const int Y = ty0 + pc / C::TW, X = tx0 + pc % C::TW;
          if (Y < H && X < W) f[it] = *reinterpret_cast<const FragT*>(v + kk * vstride + ((size_t)Y * W + X) * C::F + c * 8);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < B; ++it) {
      const int idx = tid + (base + it) * NTHREADS;
      if (base + it < IT && idx < TOTAL) *reinterpret_cast<FragT*>(VT + (size_t)idx * 8) = f[it];
    }
  }
}

// 16 accumulator-layout values (rows = channels 8g + 4hh + j) of pixel `px` from an NHWC tensor; rows >= F are 0
template <typename T, int F> SR_DEV void nas_load_rows(float (&out)[16], const T* __restrict__ p, int hh) {
  typedef typename FragOf<T>::half_type HalfT;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g < F / 8) {
      const HalfT v = *reinterpret_cast<const HalfT*>(p + g * 8 + hh * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) out[4 * g + j] = (float)v[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) out[4 * g + j] = 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// pointwise forward + mix: y = mg*yin + beta2 * ms * sum_k p_k relu(pw_k V_k + bp_k).
// frags: 6 forward fragments (k, k-step); tabs: bp[3][32] | ms_row[32] | mg_row[32] in C-init layout;
// scal: p0, p1, p2, beta2.  grid = (tiles, N), one wave per 32-pixel tile.
// ---------------------------------------------------------------------------------------------
template <typename T, int F>
__global__ __launch_bounds__(576) void nas_pw_fwd_kernel(const T* __restrict__ yin, const T* __restrict__ V,
                                                         T* __restrict__ y, const T* __restrict__ frags,
                                                         const float* __restrict__ tabs, const float* __restrict__ scal,
                                                         int H, int W, int tiles_x, long vstride) {
  typedef NasCfg<F> C;
  typedef typename FragOf<T>::half_type HalfT;
  __shared__ __attribute__((aligned(16))) T VT[3 * C::VT_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * F;
#pragma unroll
  for (int k = 0; k < 3; ++k) nas_stage_vt<T, C, 576>(VT + k * C::VT_ELEMS, V + k * vstride + img, H, W, ty0, tx0, tid);
  __syncthreads();
  const T* fr = weights_for_tile<false>(frags);
  const int ot = wave;
  const int oy = (ot / (C::TW / 8)) * 4 + (r >> 3), ox = (ot % (C::TW / 8)) * 8 + (r & 7);
  const int pc = oy * C::TW + ox;
  float S[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) S[i] = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    f32x16 acc = load_cinit(tabs + k * 32, hh);
#pragma unroll
    for (int s = 0; s < 2; ++s)
      acc = mma16<T>(load_wfrag<T>(fr, 2 * k + s, lane), lds_chunk<T>(VT + k * C::VT_ELEMS, pc * 32 + (2 * s + hh) * 8), acc);
    const float p = scal[k];
#pragma unroll
    for (int i = 0; i < 16; ++i) S[i] += p * fmaxf(acc[i], 0.f);
  }
  const int Y = ty0 + oy, X = tx0 + ox;
  if (Y < H && X < W) {
    const size_t o = img + ((size_t)Y * W + X) * F;
    const f32x16 ms = load_cinit(tabs + 96, hh), mg = load_cinit(tabs + 128, hh);
    const float b2 = scal[3];
    float xin[16];
    nas_load_rows<T, F>(xin, yin + o, hh);
#pragma unroll
    for (int g = 0; g < C::FC; ++g) {
      HalfT v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (T)(mg[4 * g + j] * xin[4 * g + j] + b2 * ms[4 * g + j] * S[4 * g + j]);
      *reinterpret_cast<HalfT*>(y + o + g * 8 + hh * 4) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// pointwise backward: recompute u_k; gu_k = p_k beta2 ms gy 1(u_k>0); GZ_k = (Wpw_k^T gu_k) 1(V_k>0);
// per-branch slabs: dWpw_k [co rows, ci cols], dbp_k = sum gu_k, r_k[c] = sum gy[c] relu(u_k)[c]; scalar
// sxy = sum gy * mg * yin.  Wave (k, grp) owns branch k and pixel tiles grp, grp+GW, ... (3 x GW waves).
// frags: 6 forward + 6 backward (rows ci, k = co chained).  grid = (wgs), persistent over tiles.
// ---------------------------------------------------------------------------------------------
// 3 branches x GW waves: the 27 (branch, pixel tile) items of a tile in two rounds (15 waves) instead of five (6 waves, round 1)
#ifndef NAS_PWB_GW_BF16
#define NAS_PWB_GW_BF16 4
#endif
template <typename T> struct NasPwb {                 // fp32 (parity mode): the staged tiles are twice as large, LDS holds scratch for 6 waves
  static constexpr int GW = sizeof(T) == 2 ? NAS_PWB_GW_BF16 : 2, THREADS = 3 * GW * 64;
};
template <typename T, int F>
__global__ __launch_bounds__(NasPwb<T>::THREADS) void nas_pw_bwd_kernel(const T* __restrict__ yin, const T* __restrict__ V,
                                                         const T* __restrict__ gy, T* __restrict__ GZ,
                                                         const T* __restrict__ frags, const float* __restrict__ tabs,
                                                         const float* __restrict__ scal, float* __restrict__ partial,
                                                         int N, int H, int W, int tiles_x, int tiles_per_img, long vstride) {
  typedef NasCfg<F> C;
  constexpr int NAS_PWB_GW = NasPwb<T>::GW, NAS_PWB_THREADS = NasPwb<T>::THREADS;
  typedef typename FragOf<T>::half_type HalfT;
  constexpr int SCR = 33 * 32;
  constexpr int WL_OFF = (3 * C::VT_ELEMS + 3 * NAS_PWB_GW * SCR) * (int)sizeof(T);     // 12 weight fragments
  constexpr int TB_OFF = WL_OFF + 12 * 512 * (int)sizeof(T);                             // bp[3][32] | ms | mg (C-init layout)
  constexpr int STAGE_BYTES = TB_OFF + 160 * 4;
  static_assert((3 * NAS_PWB_GW * C::PWB_K + 16) * 4 <= WL_OFF, "the per-wave slab copies overlay the V tiles only");
  constexpr int LDS_BYTES = STAGE_BYTES;
  __shared__ __attribute__((aligned(16))) char smem_raw[LDS_BYTES];
  T* const VT = reinterpret_cast<T*>(smem_raw);
  const T* const WL = reinterpret_cast<const T*>(smem_raw + WL_OFF);
  const float* const TB = reinterpret_cast<const float*>(smem_raw + TB_OFF);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  T* const scr = VT + 3 * C::VT_ELEMS + wave * SCR;
  const int k = wave / NAS_PWB_GW, grp = wave - k * NAS_PWB_GW;          // 3 x GW waves: (branch, pixel-tile residue)
  const T* const VTk = VT + k * C::VT_ELEMS;
  SR_STAMP_DECL;
  SR_STAMP();
  f32x16 dW = zero16();
  float db[16], rk[16], sxy = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { db[i] = 0.f; rk[i] = 0.f; }

  for (int t = blockIdx.x; t < N * tiles_per_img; t += gridDim.x) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
    const size_t img = (size_t)n * H * W * F;
    __syncthreads();
    nas_stage_vt3<T, C, NAS_PWB_THREADS>(VT, V + img, vstride, H, W, ty0, tx0, tid);
    if (t == (int)blockIdx.x) {
      // weights and tables once per workgroup, read from LDS at each use (held in registers they cost a wave per SIMD).
      // The backward fragments (rows ci, k = co) are scaled along k by c_k[co] = p_k beta2 ms[co] on the way: the
      // item loop then works with m = gy 1(u > 0) instead of gu = c_k m (exact in bf16, and no table in registers):
      // GZ = (Wpw^T diag(c_k)) m, dWpw = diag(c_k) (m^T v), dbp = c_k sum m -- the last two scaled once in the epilogue.
      typedef typename FragOf<T>::type FragT;
      const float b2 = scal[3];
      for (int i = tid; i < 12 * 64; i += NAS_PWB_THREADS) {
        FragT v = reinterpret_cast<const FragT*>(frags)[i];
        const int f = i >> 6, l = i & 63;
        if (f >= 6) {
          const int kk = (f - 6) >> 1, st = (f - 6) & 1;
          const float pk = (kk == 0 ? scal[0] : (kk == 1 ? scal[1] : scal[2])) * b2;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (T)((float)v[j] * pk * tabs[96 + (l >> 5) * 16 + 8 * st + j]);
        }
        reinterpret_cast<FragT*>(smem_raw + WL_OFF)[i] = v;
      }
      if (tid < 160) reinterpret_cast<float*>(smem_raw + TB_OFF)[tid] = tabs[tid];
    }
    __syncthreads();
    SR_STAMP();
    const T* fr = WL;
    // sxy = sum gy mg yin: its own short pass over (pixel, 8-channel chunk) items, all waves (inside the item loop its
    // operands cost one branch's waves 32 more registers; the gy lines it touches are the ones the items read next)
    for (int idx = tid; idx < C::NPXC * C::FC; idx += NAS_PWB_THREADS) {
      typedef typename FragOf<T>::type FragT;
      const int pcx = idx / C::FC, c = idx - pcx * C::FC;
      const int Y = ty0 + pcx / C::TW, X = tx0 + pcx % C::TW;
      if (Y < H && X < W) {
        const size_t oo = img + ((size_t)Y * W + X) * F + c * 8;
        const FragT a = *reinterpret_cast<const FragT*>(gy + oo), b = *reinterpret_cast<const FragT*>(yin + oo);
#pragma unroll
        for (int j = 0; j < 8; ++j) sxy += (float)a[j] * TB[128 + (j >> 2) * 16 + 4 * c + (j & 3)] * (float)b[j];
      }
    }
    SR_STAMP();
    // pixel tiles of wave (k, grp): ot = grp - k (mod GW), + GW, ...: the waves that take the extra tile sit on different SIMDs
#pragma unroll 1
    for (int ot = (grp + NAS_PWB_GW - k % NAS_PWB_GW) % NAS_PWB_GW; ot < C::NPT_O; ot += NAS_PWB_GW) {
      const int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
      const int oy = toy + (r >> 3), ox = tox + (r & 7);
      const int pc = oy * C::TW + ox;
      const int Y = ty0 + oy, X = tx0 + ox;
      const bool valid = Y < H && X < W;
      const size_t o = img + ((size_t)(valid ? Y : 0) * W + (valid ? X : 0)) * F;
      f32x16 acc = load_cinit(TB + k * 32, hh);
#pragma unroll
      for (int s = 0; s < 2; ++s) acc = mma16<T>(load_wfrag<T>(fr, 2 * k + s, lane), lds_chunk<T>(VTk, pc * 32 + (2 * s + hh) * 8), acc);
      float g[16];
      nas_load_rows<T, F>(g, gy + o, hh);
      if (!valid) {
#pragma unroll
        for (int i = 0; i < 16; ++i) g[i] = 0.f;
      }
      f32x16 gu;                                       // m = gy 1(u > 0); relu(u) gy = m u
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        gu[i] = acc[i] > 0.f ? g[i] : 0.f;
        rk[i] += gu[i] * acc[i];
        db[i] += gu[i];
      }
      // m leaves the registers first (fragments for the data gradient, pixel-major scratch for the weight gradient)
      const typename FragOf<T>::type gu0 = acc_to_frag<T, 0>(gu), gu1 = acc_to_frag<T, 1>(gu);
      scratch_store<T>(scr, gu, true, r, hh);
      // d(loss)/d(dw_k output) = (Wpw_k^T c_k m) * 1(v_k > 0), rows ci
      f32x16 gv = zero16();
      gv = mma16<T>(load_wfrag<T>(fr, 6 + 2 * k, lane), gu0, gv);
      gv = mma16<T>(load_wfrag<T>(fr, 6 + 2 * k + 1, lane), gu1, gv);
      if (valid) {
#pragma unroll
        for (int gq = 0; gq < C::FC; ++gq) {
          const HalfT vv = *reinterpret_cast<const HalfT*>(VTk + pc * 32 + gq * 8 + hh * 4);
          HalfT z;
#pragma unroll
          for (int j = 0; j < 4; ++j) z[j] = (float)vv[j] > 0.f ? (T)gv[4 * gq + j] : (T)0.f;
          *reinterpret_cast<HalfT*>(GZ + k * vstride + o + gq * 8 + hh * 4) = z;
        }
      }
      // dWpw_k[co, ci] += sum_px gu[px, co] v_k[px, ci]
      auto rows = [](int p) { return p * 32; };
      auto rowv = [=](int p) { return ((toy + (p >> 3)) * C::TW + tox + (p & 7)) * 32; };
      dW = mma16<T>(tr_frag<T>(scr, 0, lane, rows), tr_frag<T>(VTk, 0, lane, rowv), dW);
      dW = mma16<T>(tr_frag<T>(scr, 1, lane, rows), tr_frag<T>(VTk, 1, lane, rowv), dW);
    }
  }
  SR_STAMP();
  if ((int)blockIdx.x < N * tiles_per_img) {          // (a workgroup without tiles staged no table: its sums are zero, 0 x stale LDS may not be)
    const f32x16 ms = load_cinit(TB + 96, hh);
    const float ck = scal[k] * scal[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float c = ck * ms[i]; dW[i] *= c; db[i] *= c; }
  }
  const float red = half_sum32(db, rk, r);             // lane r: db[r] (r < 16) or rk[r - 16], summed over the half's pixels
  const float sxy_w = wave_sum(sxy);
  __syncthreads();
  SR_STAMP();
  // every wave writes its own copy of its branch's slab (plain stores, no turns: LDS float atomics cost ~1500 cycles per
  // wave-instruction on gfx950, and a read-modify-write chain through may-alias LDS pointers ~3 us per wave); the
  // GW copies of a branch are summed on the way out
  float* const slab = reinterpret_cast<float*>(smem_raw);
  float* const sw = slab + wave * C::PWB_K;
#pragma unroll
  for (int i = 0; i < 16; ++i) sw[i * 64 + lane] = dW[i];
  { const int i = r & 15; sw[1024 + 2 * (r & 16) + (i & 3) + 8 * (i >> 2) + 4 * hh] = red; }
  float* const sx = slab + 3 * NAS_PWB_GW * C::PWB_K;
  if (lane == 0) sx[wave] = sxy_w;
  __syncthreads();
  SR_STAMP();
  float* out = partial + (size_t)blockIdx.x * C::PWB_SLAB;
  for (int i = tid; i < 3 * C::PWB_K; i += NAS_PWB_THREADS) {
    const int kk = i / C::PWB_K, e = i - kk * C::PWB_K;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < NAS_PWB_GW; ++g) v += slab[(kk * NAS_PWB_GW + g) * C::PWB_K + e];
    out[i] = v;
  }
  if (tid < 4) {
    float v = 0.f;
    if (tid == 0)
      for (int w = 0; w < 3 * NAS_PWB_GW; ++w) v += sx[w];
    out[3 * C::PWB_K + tid] = v;
  }
  SR_STAMP();
}

// ---------------------------------------------------------------------------------------------
// depthwise backward: g_br = sum_k dw_k^T(GZ_k); g_x = gy + ms g_br; g_yin = mg g_x;
// slab: dWdw[83 taps][32] (filled by nas_dw_wgrad_kernel) | dbd[3][32] | sA[c] = sum g_br mg yin | sB[c] = sum g_x yin.
// (a) bias gradients: lanes = channels;
// (b) data gradient: (pixel group, chunk) items as in the forward.  grid = (wgs), persistent over tiles.
// ---------------------------------------------------------------------------------------------
template <typename T, int F>
__global__ __launch_bounds__(512) void nas_dw_bwd_kernel(const T* __restrict__ yin, const T* __restrict__ GZ,
                                                         const T* __restrict__ gy, T* __restrict__ gyin,
                                                         const float* __restrict__ dwp, float* __restrict__ partial,
                                                         int N, int H, int W, int tiles_x, int tiles_per_img, long vstride) {
  typedef NasCfg<F> C;
  typedef typename FragOf<T>::type FragT;
  constexpr int MAXIT = (C::NITEM + 7) / 8;
  constexpr int STAGE_BYTES = C::P3_ELEMS * (int)sizeof(T);
  constexpr int LDS_BYTES = STAGE_BYTES > C::DWB_SLAB * 4 ? STAGE_BYTES : C::DWB_SLAB * 4;
  __shared__ __attribute__((aligned(16))) char smem_raw[LDS_BYTES];
  T* const GT = reinterpret_cast<T*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ch = lane & 31, half = lane >> 5;
  float accb[3], sA[MAXIT][8], sB[MAXIT][8];
#pragma unroll
  for (int i = 0; i < 3; ++i) accb[i] = 0.f;
#pragma unroll
  for (int i = 0; i < MAXIT; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { sA[i][j] = 0.f; sB[i][j] = 0.f; }

  for (int t = blockIdx.x; t < N * tiles_per_img; t += gridDim.x) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
    const size_t img = (size_t)n * H * W * F;
    float gbr[MAXIT][8];
#pragma unroll
    for (int i = 0; i < MAXIT; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) gbr[i][j] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ks = 3 + 2 * k, off = 3 - ks / 2, wbase = nas_woff(k);
      __syncthreads();
      nas_stage_halo3<T, C, 512>(GT, GZ + k * vstride + img, nullptr, H, W, ty0, tx0, tid);
      __syncthreads();
      // (b) data gradient through the flipped stencil
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int item = __builtin_amdgcn_readfirstlane(wave + 8 * it);
        if (item < C::NITEM) {
          const int g = item / C::FC, c = item - g * C::FC;
          const int px = g * 64 + lane;
          const bool valid = px < C::NPXC;
          const int oy = valid ? px / C::TW : 0, ox = valid ? px % C::TW : 0;
          const float* w = dwp + wbase + c * 8;              // wave-uniform scalar loads (LDS copies measured slower here)
#pragma unroll 1
          for (int ty = 0; ty < ks; ++ty) {
#pragma unroll
            for (int tx = 0; tx < ks; ++tx) {
              const FragT v = *reinterpret_cast<const FragT*>(GT + ((oy + off + ty) * C::PW + ox + off + tx) * F + c * 8);
              const float* wt = w + ((ks - 1 - ty) * ks + (ks - 1 - tx)) * 32;
#pragma unroll
              for (int j = 0; j < 8; ++j) gbr[it][j] += wt[j] * (float)v[j];
            }
          }
        }
      }
      // (a) bias gradient of the depthwise conv: lane = channel, the two lane halves split the pixels, the three
      //     stencils go to waves 0..2 (the weight gradients are nas_dw_wgrad_kernel's)
      if (ch < F && wave == k) {
        for (int p = half; p < C::NPXC; p += 2) {
          const int oy = p / C::TW, ox = p - oy * C::TW;
          accb[k] += (float)GT[((oy + 3) * C::PW + ox + 3) * F + ch];
        }
      }
    }
    // epilogue of the data gradient
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int item = __builtin_amdgcn_readfirstlane(wave + 8 * it);
      if (item < C::NITEM) {
        const int g = item / C::FC, c = item - g * C::FC;
        const int px = g * 64 + lane;
        const int oy = px / C::TW, ox = px % C::TW;
        const int Y = ty0 + oy, X = tx0 + ox;
        if (px < C::NPXC && Y < H && X < W) {
          const size_t o = img + ((size_t)Y * W + X) * F + c * 8;
          const FragT yv = *reinterpret_cast<const FragT*>(yin + o), gv = *reinterpret_cast<const FragT*>(gy + o);
          FragT outv;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float mgc = dwp[C::MG + c * 8 + j], msc = dwp[C::MS + c * 8 + j];
            const float gx = (float)gv[j] + msc * gbr[it][j];
            outv[j] = (T)(mgc * gx);
            sA[it][j] += gbr[it][j] * mgc * (float)yv[j];
            sB[it][j] += gx * (float)yv[j];
          }
          *reinterpret_cast<FragT*>(gyin + o) = outv;
        }
      }
    }
  }
  __syncthreads();
  float* slab = reinterpret_cast<float*>(smem_raw);
  for (int i = tid; i < C::DWB_SLAB; i += 512) slab[i] = 0.f;
  __syncthreads();
  if (ch < F) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (wave == k) {
        const float v = accb[k] + __shfl_xor(accb[k], 32);
        if (half == 0) slab[(83 + k) * 32 + ch] = v;
      }
  }
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int item = wave + 8 * it;
    if (item < C::NITEM) {
      const int c = item % C::FC;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = wave_sum(sA[it][j]), b = wave_sum(sB[it][j]);
        if (lane == 0) { atomicAdd(slab + 86 * 32 + c * 8 + j, a); atomicAdd(slab + 87 * 32 + c * 8 + j, b); }
      }
    }
  }
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * C::DWB_SLAB;
  for (int i = tid; i < C::DWB_SLAB; i += 512) out[i] = slab[i];
}

// ---------------------------------------------------------------------------------------------
// depthwise weight gradients on the matrix cores: dW_k[tap][c] = sum_px GZ_k[px][c] * x1[px + tap][c] is the
// DIAGONAL of the 32x32 product GZ_k^T x1(shifted); computing the whole product with MFMA (pixels contracted
// through transposed LDS reads, as the 3x3 weight gradients do) and keeping the diagonal is ~30x faster than the
// per-tap VALU reduction it replaces (one 2-byte LDS read per multiply-add).  12 waves; wave w owns taps
// w, w + 12, ... of each stencil (1 + 3 + 5 accumulator tiles).  Writes the dWdw[83][32] part of the slab.
// ---------------------------------------------------------------------------------------------
template <typename T, int F, int K>
SR_DEV void nas_dw_wgrad_k(T* X1, T* GT, const T* __restrict__ yin, const T* __restrict__ GZk, const float* __restrict__ dwp,
                           float* __restrict__ out, int N, int H, int W, int tiles_x, int tiles_per_img) {
  typedef NasCfg<F> C;
  typedef typename FragOf<T>::type FragT;
  constexpr int NTHREADS = 768, KS = 3 + 2 * K, OFF = 3 - KS / 2, NTAP = KS * KS, NSLOT = 1 + 2 * K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  f32x16 acc[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) acc[i] = zero16();
  for (int t = blockIdx.x; t < N * tiles_per_img; t += gridDim.x) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
    const size_t img = (size_t)n * H * W * F;
    __syncthreads();
    for (int idx = tid; idx < (C::NP3 + 2) * 4; idx += NTHREADS) {
      const int hp = idx >> 2, c = idx & 3;
      FragT v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (T)0.f;
      if (hp < C::NP3 && c < C::FC) {
        const int hy = hp / C::PW, hx = hp - hy * C::PW;
        const int Y = ty0 - 3 + hy, X = tx0 - 3 + hx;
        if (Y >= 0 && Y < H && X >= 0 && X < W) {
          v = *reinterpret_cast<const FragT*>(yin + img + ((size_t)Y * W + X) * F + c * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (T)((float)v[j] * dwp[C::M1 + c * 8 + j]);
        }
      }
      *reinterpret_cast<FragT*>(X1 + idx * 8) = v;
    }
    nas_stage_vt<T, C, NTHREADS>(GT, GZk + img, H, W, ty0, tx0, tid);
    __syncthreads();
#pragma unroll 1
    for (int ot = 0; ot < C::NPT_O; ++ot) {
      const int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
      auto rowg = [=](int p) { return ((toy + (p >> 3)) * C::TW + tox + (p & 7)) * 32; };
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const FragT a = tr_frag<T>(GT, s, lane, rowg);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
          const int tp = wave + 12 * i;
          if (tp < NTAP) {
            const int ty = tp / KS, tx = tp - ty * KS;
            auto rowx = [=](int p) { return ((toy + (p >> 3) + OFF + ty) * C::PW + tox + (p & 7) + OFF + tx) * 32; };
            acc[i] = mma16<T>(a, tr_frag<T>(X1, s, lane, rowx), acc[i]);
          }
        }
      }
    }
  }
  // diagonal of every tile: accumulator register q of lane (r, hh) is row (q & 3) + 8 (q >> 2) + 4 hh, column r
  const bool mine = hh == ((r >> 2) & 1);
  const int isel = (r & 3) + 4 * (r >> 3);
  const int tbase = (K == 0 ? 0 : (K == 1 ? 9 : 34));
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int tp = wave + 12 * i;
    if (tp < NTAP) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) v = (q == isel) ? acc[i][q] : v;
      if (mine) out[(tbase + tp) * 32 + r] = v;
    }
  }
}

// grid = (wgs, 3): blockIdx.y = stencil (3x3, 5x5, 7x7); each workgroup walks the tiles t = blockIdx.x, += gridDim.x
template <typename T, int F>
__global__ __launch_bounds__(768) void nas_dw_wgrad_kernel(const T* __restrict__ yin, const T* __restrict__ GZ,
                                                           const float* __restrict__ dwp, float* __restrict__ partial,
                                                           int N, int H, int W, int tiles_x, int tiles_per_img,
                                                           long vstride) {
  typedef NasCfg<F> C;
  constexpr int X_ELEMS = (C::NP3 + 2) * 32;
  __shared__ __attribute__((aligned(16))) T smem[X_ELEMS + C::VT_ELEMS];
  T* const X1 = smem;                      // m1 * yin with a 3-pixel halo, [NP3 + 2][32] (channels >= F zero)
  T* const GT = smem + X_ELEMS;            // core tile of GZ_k, [NPXC + 1][32]
  float* out = partial + (size_t)blockIdx.x * C::DWB_SLAB;
  if (blockIdx.y == 0) nas_dw_wgrad_k<T, F, 0>(X1, GT, yin, GZ, dwp, out, N, H, W, tiles_x, tiles_per_img);
  else if (blockIdx.y == 1) nas_dw_wgrad_k<T, F, 1>(X1, GT, yin, GZ + vstride, dwp, out, N, H, W, tiles_x, tiles_per_img);
  else nas_dw_wgrad_k<T, F, 2>(X1, GT, yin, GZ + 2 * vstride, dwp, out, N, H, W, tiles_x, tiles_per_img);
}

// The three stencils from ONE workgroup per tile (bf16; in fp32 the four staged images do not fit the LDS): x1 is staged
// once instead of three times, the 83 taps fill 7 accumulator tiles on each of 12 waves (83 / 84 slots used, against
// 9 / 12 + 25 / 36 + 49 / 60 over three workgroups that could not share a CU for their registers), and the staging loads
// are issued in batches.  Wave w owns the taps w, w + 12, ... of the concatenated tap list 3x3 | 5x5 | 7x7 (the slab's own
// order); the GZ_k fragment is reloaded only where consecutive slots change stencil.  grid = (wgs).
template <int F>
__global__ __launch_bounds__(768) void nas_dw_wgrad3_kernel(const __bf16* __restrict__ yin, const __bf16* __restrict__ GZ,
                                                            const float* __restrict__ dwp, float* __restrict__ partial,
                                                            int N, int H, int W, int tiles_x, int tiles_per_img, long vstride) {
  typedef __bf16 T;
  typedef NasCfg<F> C;
  typedef typename FragOf<T>::type FragT;
  constexpr int NTHREADS = 768, X_ELEMS = (C::NP3 + 2) * 32, NSLOT = 7, NTAP = 83;
  __shared__ __attribute__((aligned(16))) T smem[X_ELEMS + 3 * C::VT_ELEMS];
  T* const X1 = smem;                      // m1 * yin with a 3-pixel halo, [NP3 + 2][32] (channels >= F zero)
  T* const GT = smem + X_ELEMS;            // core tiles of GZ_0..2, [3][NPXC + 1][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31;
  SR_STAMP_DECL;
  SR_STAMP();
  f32x16 acc[NSLOT];
  int xoff[NSLOT], goff[NSLOT];            // wave-uniform: X1 element offset of the tap, GT element offset of its stencil
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    acc[i] = zero16();
    const int tp = wave + 12 * i;
    const int k = tp < 9 ? 0 : (tp < 34 ? 1 : 2), lt = tp - (k == 0 ? 0 : (k == 1 ? 9 : 34)), ks = 3 + 2 * k, off = 3 - ks / 2;
    const int ty = lt / ks, tx = lt - ty * ks;
    xoff[i] = __builtin_amdgcn_readfirstlane(((off + ty) * C::PW + off + tx) * 32);
    goff[i] = __builtin_amdgcn_readfirstlane(k * C::VT_ELEMS);
  }
  for (int t = blockIdx.x; t < N * tiles_per_img; t += gridDim.x) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
    const size_t img = (size_t)n * H * W * F;
    __syncthreads();
    {
      constexpr int TOTAL = (C::NP3 + 2) * 4, IT = (TOTAL + NTHREADS - 1) / NTHREADS;
      FragT f[IT];
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * NTHREADS, hp = idx >> 2, c = idx & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[it][j] = (T)0.f;
        if (hp < C::NP3 && c < C::FC) {
          const int hy = hp / C::PW, hx = hp - hy * C::PW;
          const int Y = ty0 - 3 + hy, X = tx0 - 3 + hx;
          if (Y >= 0 && Y < H && X >= 0 && X < W) f[it] = *reinterpret_cast<const FragT*>(yin + img + ((size_t)Y * W + X) * F + c * 8);
        }
      }
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int idx = tid + it * NTHREADS, c = idx & 3;
        if (idx < TOTAL) {
          FragT v = f[it];
          if (c < C::FC) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (T)((float)v[j] * dwp[C::M1 + c * 8 + j]);
          }
          *reinterpret_cast<FragT*>(X1 + idx * 8) = v;
        }
      }
    }
    nas_stage_vt3<T, C, NTHREADS>(GT, GZ + img, vstride, H, W, ty0, tx0, tid);
    __syncthreads();
    SR_STAMP();
#pragma unroll 1
    for (int ot = 0; ot < C::NPT_O; ++ot) {
      const int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
      const int gbase = (toy * C::TW + tox) * 32, xbase = (toy * C::PW + tox) * 32;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        FragT a;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
          if (wave + 12 * i < NTAP) {
            if (i == 0 || goff[i] != goff[i - 1]) {
              const T* g = GT + goff[i] + gbase;
              a = tr_frag<T>(g, s, lane, [](int p) { return ((p >> 3) * C::TW + (p & 7)) * 32; });
            }
            const T* x = X1 + xoff[i] + xbase;
            acc[i] = mma16<T>(a, tr_frag<T>(x, s, lane, [](int p) { return ((p >> 3) * C::PW + (p & 7)) * 32; }), acc[i]);
          }
        }
      }
    }
    SR_STAMP();
  }
  // diagonal of every tile: accumulator register q of lane (r, hh) is row (q & 3) + 8 (q >> 2) + 4 hh, column r
  float* out = partial + (size_t)blockIdx.x * C::DWB_SLAB;
  const bool mine = (lane >> 5) == ((r >> 2) & 1);
  const int isel = (r & 3) + 4 * (r >> 3);
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int tp = wave + 12 * i;
    if (tp < NTAP) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) v = (q == isel) ? acc[i][q] : v;
      if (mine) out[tp * 32 + r] = v;
    }
  }
  SR_STAMP();
}

// ---------------------------------------------------------------------------------------------
// The 0/1 masks, gates and latency terms of a supernet step (reference models/ops.py:33-43 rounding(), wdsr_b.py:517-534
// ConditionFunction, speed_estimator.py:57-76) for every block from ONE small launch -- as torch ops they are ~30 launches
// of a few microseconds each on tensors of 32 .. 512 values.  Values only: the straight-through gradients stay with
// the caller.  rounding(w, 8): w >= 0.5, or -- if fewer than 8 entries pass -- w >= the 8th largest (ties kept), i.e.
// fewer than 8 entries strictly larger.
// out: mask_hard[F] | c_mask | ms_hard[nb][F] (split.weight >= 0.5) | c_split[nb] | speed_curr[nb] | gates[nb][2]
// ---------------------------------------------------------------------------------------------
constexpr int NAS_SCALARS_MAX = 4096;                 // (nb + 1) * F values staged in LDS

__global__ __launch_bounds__(1024) void nas_scalars_kernel(const float* __restrict__ mask_w, const float* __restrict__ split_w,
                                                           const float* __restrict__ alpha, const float* __restrict__ alpha1,
                                                           const float* __restrict__ alpha2, int nb, int F,
                                                           float* __restrict__ out, float* __restrict__ src, long src_stride,
                                                           int off_mg, float* __restrict__ scal) {
  __shared__ float Wl[NAS_SCALARS_MAX], Rl[NAS_SCALARS_MAX];      // rows 0 .. nb - 1: split.weight of the blocks; row nb: the global mask
  float* const mask_hard = out;
  float* const c_mask_o = out + F;
  float* const ms_hard = out + F + 1;
  float* const c_split_o = ms_hard + (size_t)nb * F;
  float* const speed = c_split_o + nb;
  float* const gates = speed + nb;
  const int tid = threadIdx.x, nthr = blockDim.x, total = (nb + 1) * F;
  for (int i = tid; i < total; i += nthr) Wl[i] = i < nb * F ? split_w[i] : mask_w[i - nb * F];
  __syncthreads();
  for (int e = tid; e < total; e += nthr) {
    const int row = e / F;
    const float* w = Wl + row * F;
    const float wi = Wl[e];
    int pass = 0, above = 0;
    for (int j = 0; j < F; ++j) { pass += w[j] >= 0.5f; above += w[j] > wi; }
    const float hard = wi >= 0.5f ? 1.f : 0.f, r8 = pass >= 8 ? hard : (above < 8 ? 1.f : 0.f);
    Rl[e] = r8;
    if (row < nb) ms_hard[e] = hard;
    else mask_hard[e - nb * F] = r8;
  }
  __syncthreads();
  for (int b = tid; b < nb; b += nthr) {
    float cm = 0.f, cs = 0.f;
    for (int i = 0; i < F; ++i) { cm += Rl[nb * F + i]; cs += Rl[b * F + i]; }
    if (b == 0) *c_mask_o = cm;
    c_split_o[b] = cs;
    const float t = cs + 0.2f * cm;
    speed[b] = t * 9.f * alpha[3 * b] / 40.f + t * 25.f * alpha[3 * b + 1] / 40.f + t * 49.f * alpha[3 * b + 2] / 40.f;
    const float g1 = alpha1[b] >= alpha2[b] ? 1.f : 0.f;
    gates[2 * b] = g1;
    gates[2 * b + 1] = 1.f - g1;
    if (scal) {                                        // per-block kernel scalars: softmax(alpha) | beta2 (training: the gate)
      const float a0 = alpha[3 * b], a1 = alpha[3 * b + 1], a2 = alpha[3 * b + 2], mx = fmaxf(a0, fmaxf(a1, a2));
      const float e0 = expf(a0 - mx), e1 = expf(a1 - mx), e2 = expf(a2 - mx), inv = 1.f / (e0 + e1 + e2);
      scal[4 * b] = e0 * inv; scal[4 * b + 1] = e1 * inv; scal[4 * b + 2] = e2 * inv; scal[4 * b + 3] = 1.f - g1;
    }
  }
  if (src) {                                           // the mask columns of the operand source rows: mg | ms | mg ms | 0 | 1
    for (int e = tid; e < nb * F; e += nthr) {
      const int b = e / F, c = e - b * F;
      float* row = src + (size_t)b * src_stride + off_mg;
      const float mg = Rl[nb * F + c], ms = Wl[e] >= 0.5f ? 1.f : 0.f;
      row[c] = mg; row[F + c] = ms; row[2 * F + c] = mg * ms;
      if (c == 0) { row[3 * F] = 0.f; row[3 * F + 1] = 1.f; }
    }
  }
}

// gradients of the masks / gates / branch weights from the sums the block kernels left in d(source):
// q_k = sum_c ms[c] r_k[c]; g_p = beta2 q; g_beta = (sxy, sxy + sum_k p_k q_k); g_ms = sA + beta2 sum_k p_k r_k; g_mg = sum_b sB
// out: g_p[nb][3] | g_beta[nb][2] | g_ms[nb][F] | g_mg[F]
__global__ __launch_bounds__(1024) void nas_mask_grads_kernel(const float* __restrict__ dsrc, long ds, int off_r, int off_sxy,
                                                              int off_sA, int off_sB, const float* __restrict__ ms,
                                                              const float* __restrict__ p, const float* __restrict__ beta, int nb,
                                                              int F, float* __restrict__ out) {
  float* const g_p = out;
  float* const g_beta = out + 3 * nb;
  float* const g_ms = g_beta + 2 * nb;
  float* const g_mg = g_ms + (size_t)nb * F;
  const int tid = threadIdx.x, nthr = blockDim.x;
  __shared__ float Q[3 * 1024];
  for (int i = tid; i < 3 * nb && i < 3 * 1024; i += nthr) {       // q_k of block b: one thread each, its loads independent
    const int b = i / 3, kq = i - 3 * b;
    const float* row = dsrc + (size_t)b * ds + off_r + kq * F;
    float q = 0.f;
    for (int c = 0; c < F; ++c) q += ms[b * F + c] * row[c];
    Q[i] = q;
    g_p[i] = beta[2 * b + 1] * q;
  }
  __syncthreads();
  for (int b = tid; b < nb; b += nthr) {
    const float sxy = dsrc[(size_t)b * ds + off_sxy];
    g_beta[2 * b] = sxy;
    g_beta[2 * b + 1] = sxy + p[3 * b] * Q[3 * b] + p[3 * b + 1] * Q[3 * b + 1] + p[3 * b + 2] * Q[3 * b + 2];
  }
  for (int e = tid; e < nb * F; e += nthr) {
    const int b = e / F, c = e - b * F;
    const float* row = dsrc + (size_t)b * ds;
    float v = 0.f;
    for (int k = 0; k < 3; ++k) v += p[3 * b + k] * row[off_r + k * F + c];
    g_ms[e] = row[off_sA + c] + beta[2 * b + 1] * v;
  }
  for (int c = tid; c < F; c += nthr) {
    float v = 0.f;
    for (int b = 0; b < nb; ++b) v += dsrc[(size_t)b * ds + off_sB + c];
    g_mg[c] = v;
  }
}
