// Depthwise 3x3 / 5x5 / 7x7 stencils of the NAS block with LANE = CHANNEL and v_dot2c_f32_bf16 (bf16 mode, gfx950).
// Reference op: the three Conv_sep depthwise convolutions of Split_Block.forward_body, models/wdsr_b.py:375-402,482-496.
//
// What bounds the VALU kernels of nas_block.h (lane = pixel, 8 channels per lane) is not arithmetic but LDS reads of the
// stencil weights: they are uniform over the wave there, every tap costs two 16-byte broadcast reads next to one data read,
// and a broadcast read costs the LDS as much as a data read.  (A dot2 formulation with the same lane mapping had 3.4x fewer
// VALU instructions and ran exactly as long.)  Here a lane owns ONE channel: its 83 stencil weights are 49 packed bf16
// pairs that sit in its registers for the whole launch, and the LDS only delivers pixels:
//   * the halo'd tile is pair-interleaved, [pixel pair][channel][2 pixels], twice: pairs starting at even pixels (XE)
//     and at odd pixels (XO).  Lanes 0..31 = channels on even output columns (XE), lanes 32..63 = the same channels on odd
//     columns (XO): a wave-wide 4-byte read is two runs of 32 consecutive dwords;
//   * one dot2 multiplies two horizontally adjacent pixels of the channel with two adjacent taps; the 3x3 and 5x5 windows
//     lie inside the 7x7 one and use the same pixel pairs with zero-padded weight pairs: 49 dot2 per output and channel;
//   * a lane slides along its row two columns at a time: of the 7 x 4 pairs of its window 7 x 3 stay in registers, only the
//     7 new pairs are read -- 7 LDS dwords per 83 multiply-adds.
// The weights are rounded to bf16 (as the pointwise weights on the matrix cores are); fp32 parity mode keeps nas_block.h.
#pragma once
#include "nas_block.h"

template <int F> struct NasLcCfg {
  typedef NasCfg<F> C;
  static constexpr int NPAIR = C::NP3 / 2;                             // pixel pairs of the halo-3 tile (PW is even)
  static constexpr int XP_DW = (NPAIR + 4) * F;                        // dwords of one pair-interleaved copy (+ slack pairs)
  static constexpr int W7P = 0, W5P = 7 * 4, W3P = W5P + 5 * 3, NWP = W3P + 3 * 2;   // packed weight rows of 32 dwords
  static constexpr int NUNIT = C::TH * 2, UW = C::TW / 2;              // a unit = one output row x half the columns
  static_assert(C::PW % 2 == 0 && C::NP3 % 2 == 0 && UW % 2 == 0, "pairs must not straddle rows; units start on even columns");
};

typedef __attribute__((ext_vector_type(2))) __bf16 nas_bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned nas_u32x4;

SR_DEV unsigned nas_pack2(float lo, float hi) {
  __bf16 a, b;
  cvt_pair<__bf16>(a, b, lo, hi);
  nas_bf16x2 v = {a, b};
  return __builtin_bit_cast(unsigned, v);
}
SR_DEV float nas_dot2(unsigned x, unsigned w, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(nas_bf16x2, x), __builtin_bit_cast(nas_bf16x2, w), acc, false);
}

// packed stencil weights WP[row][32 channels]: row = (stencil, window row, window column pair (0,1) (2,3) (4,5) (6,7));
// `flip`: the data-gradient form (window tap (ty, tx) weighs with w[ks-1-ty][ks-1-tx])
template <int F, int NT>
SR_DEV void nas_pack_weights(unsigned* WP, const float* __restrict__ dwp, int tid, bool flip) {
  typedef NasLcCfg<F> D;
  typedef NasCfg<F> C;
  constexpr int TOTAL = D::NWP * 32, IT = (TOTAL + NT - 1) / NT;
  float v[IT][2];                                      // every load of a thread in flight before the first is packed
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int i = tid + it * NT;
    const int row = i >> 5, ch = i & 31;
    int ks, r, t, wbase;
    if (row < D::W5P) { ks = 7; r = row / 4; t = row % 4; wbase = C::W7; }
    else if (row < D::W3P) { ks = 5; r = (row - D::W5P) / 3; t = (row - D::W5P) % 3; wbase = C::W5; }
    else { ks = 3; r = (row - D::W3P) / 2; t = (row - D::W3P) % 2 + 1; wbase = C::W3; }
    const int off = 3 - ks / 2;                        // the stencil's first window column
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int tx = 2 * t + h - off;                  // stencil column of window column 2 t + h
      v[it][h] = 0.f;
      if (i < TOTAL && tx >= 0 && tx < ks) {
        const int rr = flip ? ks - 1 - r : r, cc = flip ? ks - 1 - tx : tx;
        v[it][h] = dwp[wbase + (rr * ks + cc) * 32 + ch];
      }
    }
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int i = tid + it * NT;
    if (i < TOTAL) WP[i] = nas_pack2(v[it][0], v[it][1]);
  }
}

// pair-interleaved halo-3 tile of `src` (scaled per channel by scale[c], nullptr: 1): XE[q] = pixels (2q, 2q+1),
// XO[q] = pixels (2q+1, 2q+2), each [pair][F channels] dwords.  All global loads of a thread are issued before the first is used.
template <int F, int NTHREADS>
SR_DEV void nas_stage_pairs(unsigned* XE, unsigned* XO, const __bf16* __restrict__ src, const float* __restrict__ scale, int H, int W,
                            int ty0, int tx0, int tid) {
  typedef NasLcCfg<F> D;
  typedef NasCfg<F> C;
  constexpr int TOTAL = (D::NPAIR + 4) * C::FC, ITER = (TOTAL + NTHREADS - 1) / NTHREADS;
  bf16x8 f[ITER][3];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = tid + it * NTHREADS;
    const int q = idx / C::FC, c = idx - q * C::FC;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int hp = 2 * q + k;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[it][k][j] = (__bf16)0.f;
      if (idx < TOTAL && hp < C::NP3) {
        const int hy = hp / C::PW, hx = hp - hy * C::PW;
        const int Y = ty0 - 3 + hy, X = tx0 - 3 + hx;
        if (Y >= 0 && Y < H && X >= 0 && X < W) f[it][k] = *reinterpret_cast<const bf16x8*>(src + ((size_t)Y * W + X) * F + c * 8);
      }
    }
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = tid + it * NTHREADS;
    if (idx >= TOTAL) continue;
    const int q = idx / C::FC, c = idx - q * C::FC;
    nas_u32x4 e[2], o[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float sc = scale ? scale[c * 8 + j] : 1.f;
      const float v0 = (float)f[it][0][j] * sc, v1 = (float)f[it][1][j] * sc, v2 = (float)f[it][2][j] * sc;
      e[j >> 2][j & 3] = nas_pack2(v0, v1);
      o[j >> 2][j & 3] = nas_pack2(v1, v2);
    }
    nas_u32x4* pe = reinterpret_cast<nas_u32x4*>(XE + (size_t)q * F + c * 8);
    nas_u32x4* po = reinterpret_cast<nas_u32x4*>(XO + (size_t)q * F + c * 8);
    pe[0] = e[0]; pe[1] = e[1];
    po[0] = o[0]; po[1] = o[1];
  }
}

// a lane's packed weights, in registers for the whole launch
struct NasLcW {
  unsigned w7[7][4], w5[5][3], w3[3][2];
  template <int F> SR_DEV void load(const unsigned* WP, int ch) {
    typedef NasLcCfg<F> D;
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) w7[r][t] = WP[(D::W7P + r * 4 + t) * 32 + ch];
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
      for (int t = 0; t < 3; ++t) w5[r][t] = WP[(D::W5P + r * 3 + t) * 32 + ch];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int t = 0; t < 2; ++t) w3[r][t] = WP[(D::W3P + r * 2 + t) * 32 + ch];
  }
};

// the three stencils of one output pixel of this lane's channel from its 7 x 4 window of pixel pairs
SR_DEV void nas_lc_point(float& z3, float& z5, float& z7, const unsigned (&win)[7][4], const NasLcW& w) {
#pragma unroll
  for (int ty = 0; ty < 7; ++ty) {
#pragma unroll
    for (int t = 0; t < 4; ++t) z7 = nas_dot2(win[ty][t], w.w7[ty][t], z7);
    if (ty >= 1 && ty <= 5) {
#pragma unroll
      for (int t = 0; t < 3; ++t) z5 = nas_dot2(win[ty][t], w.w5[ty - 1][t], z5);
    }
    if (ty >= 2 && ty <= 4) {
#pragma unroll
      for (int t = 1; t < 3; ++t) z3 = nas_dot2(win[ty][t], w.w3[ty - 2][t - 1], z3);
    }
  }
}

// depthwise forward: V_k = relu(dw_k(m1 * yin) + bd_k), k = 3, 5, 7.  grid = (tiles, N), 8 waves; a wave walks units
// (output row, column half) = wave, wave + 8, wave + 16 and slides along the unit two columns per step.
template <int F>
__global__ __launch_bounds__(512) void nas_dw_fwd_lc_kernel(const __bf16* __restrict__ yin, __bf16* __restrict__ V,
                                                            const float* __restrict__ dwp, int H, int W, int tiles_x, long vstride) {
  typedef NasCfg<F> C;
  typedef NasLcCfg<F> D;
  __shared__ __attribute__((aligned(16))) unsigned XE[D::XP_DW];
  __shared__ __attribute__((aligned(16))) unsigned XO[D::XP_DW];
  __shared__ __attribute__((aligned(16))) unsigned WP[D::NWP * 32];
  const int tid = threadIdx.x, lane = tid & 63, ch = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * F;
  nas_pack_weights<F, 512>(WP, dwp, tid, false);
  nas_stage_pairs<F, 512>(XE, XO, yin + img, dwp + C::M1, H, W, ty0, tx0, tid);
  __syncthreads();
  NasLcW w;
  w.template load<F>(WP, ch);
  const float b3 = dwp[C::BD + ch], b5 = dwp[C::BD + 32 + ch], b7 = dwp[C::BD + 64 + ch];
  const unsigned* X = half ? XO : XE;
  const bool chan = ch < F;
#pragma unroll 1
  for (int u = wave; u < D::NUNIT; u += 8) {
    const int oy = u >> 1, ox0 = (u & 1) * D::UW + half;                // this lane's first column (its parity = half)
    const unsigned* base = X + (size_t)((oy * C::PW + ox0) >> 1) * F + (chan ? ch : 0);
    unsigned win[7][4];
#pragma unroll
    for (int ty = 0; ty < 7; ++ty)
#pragma unroll
      for (int t = 0; t < 4; ++t) win[ty][t] = base[(ty * (C::PW / 2) + t) * F];
#pragma unroll
    for (int s = 0; s < D::UW / 2; ++s) {
      float z3 = b3, z5 = b5, z7 = b7;
      nas_lc_point(z3, z5, z7, win, w);
      const int Y = ty0 + oy, Xc = tx0 + ox0 + 2 * s;
      if (chan && Y < H && Xc < W) {
        const size_t o = img + ((size_t)Y * W + Xc) * F + ch;
        V[o] = (__bf16)fmaxf(z3, 0.f);
        V[vstride + o] = (__bf16)fmaxf(z5, 0.f);
        V[2 * vstride + o] = (__bf16)fmaxf(z7, 0.f);
      }
      if (s + 1 < D::UW / 2) {                         // slide two columns: three pairs per row stay, one is read
#pragma unroll
        for (int ty = 0; ty < 7; ++ty) {
          win[ty][0] = win[ty][1];
          win[ty][1] = win[ty][2];
          win[ty][2] = win[ty][3];
          win[ty][3] = base[(ty * (C::PW / 2) + s + 4) * F];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The block's forward in ONE launch (bf16): the depthwise part as nas_dw_fwd_lc_kernel, whose three outputs V_k go to an LDS
// tile [k][pixel][32] instead of 2-byte global stores; after one barrier the tile is (a) written to V with 16-byte stores
// (the backward needs it) and (b) the B operand of the pointwise part (nas_pw_fwd_kernel's arithmetic, reading LDS it did
// not have to stage).  Saves the second launch, its 14 MB read of V and its staging latency; results are bit-identical to the
// two-kernel route.  The pointwise part has one 32-pixel tile per wave.  grid = (tiles, N).
// ---------------------------------------------------------------------------------------------
// twelve waves (round 3): the 24 row units of the stencil phase are two rounds instead of three (nine waves before; 140 VGPRs fit three
// waves per SIMD); the pointwise part uses the first nine
constexpr int NAS_BLOCK_FWD_THREADS = 768;
template <int F>
__global__ __launch_bounds__(NAS_BLOCK_FWD_THREADS) void nas_block_fwd_kernel(const __bf16* __restrict__ yin, __bf16* __restrict__ V,
                                                            __bf16* __restrict__ y, const float* __restrict__ dwp,
                                                            const __bf16* __restrict__ frags, const float* __restrict__ tabs,
                                                            const float* __restrict__ scal, int H, int W, int tiles_x, long vstride) {
  typedef __bf16 T;
  typedef NasCfg<F> C;
  typedef NasLcCfg<F> D;
  typedef typename FragOf<T>::type FragT;
  typedef typename FragOf<T>::half_type HalfT;
  constexpr int NTHREADS = NAS_BLOCK_FWD_THREADS;
  __shared__ __attribute__((aligned(16))) unsigned XE[D::XP_DW];
  __shared__ __attribute__((aligned(16))) unsigned XO[D::XP_DW];
  __shared__ __attribute__((aligned(16))) unsigned WP[D::NWP * 32];
  __shared__ __attribute__((aligned(16))) T VT[3 * C::VT_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, ch = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * F;
  nas_pack_weights<F, NTHREADS>(WP, dwp, tid, false);
  nas_stage_pairs<F, NTHREADS>(XE, XO, yin + img, dwp + C::M1, H, W, ty0, tx0, tid);
  __syncthreads();
  {
    NasLcW w;
    w.template load<F>(WP, ch);
    const float b3 = dwp[C::BD + ch], b5 = dwp[C::BD + 32 + ch], b7 = dwp[C::BD + 64 + ch];
    const unsigned* X = half ? XO : XE;
    const bool chan = ch < F;
#pragma unroll 1
    for (int u = wave; u < D::NUNIT; u += NTHREADS / 64) {
      const int oy = u >> 1, ox0 = (u & 1) * D::UW + half;                // this lane's first column (its parity = half)
      const unsigned* base = X + (size_t)((oy * C::PW + ox0) >> 1) * F + (chan ? ch : 0);
      unsigned win[7][4];
#pragma unroll
      for (int ty = 0; ty < 7; ++ty)
#pragma unroll
        for (int t = 0; t < 4; ++t) win[ty][t] = base[(ty * (C::PW / 2) + t) * F];
#pragma unroll
      for (int s = 0; s < D::UW / 2; ++s) {
        float z3 = b3, z5 = b5, z7 = b7;
        nas_lc_point(z3, z5, z7, win, w);
        T* v = VT + (oy * C::TW + ox0 + 2 * s) * 32 + ch;               // channels >= F: the zero rows the MFMA wants
        v[0] = chan ? (T)fmaxf(z3, 0.f) : (T)0.f;
        v[C::VT_ELEMS] = chan ? (T)fmaxf(z5, 0.f) : (T)0.f;
        v[2 * C::VT_ELEMS] = chan ? (T)fmaxf(z7, 0.f) : (T)0.f;
        if (s + 1 < D::UW / 2) {                         // slide two columns: three pairs per row stay, one is read
#pragma unroll
          for (int ty = 0; ty < 7; ++ty) {
            win[ty][0] = win[ty][1];
            win[ty][1] = win[ty][2];
            win[ty][2] = win[ty][3];
            win[ty][3] = base[(ty * (C::PW / 2) + s + 4) * F];
          }
        }
      }
    }
  }
  __syncthreads();
  // (a) V_k to global: (branch, pixel, 8-channel chunk) items, 16 bytes each
  for (int idx = tid; idx < 3 * C::NPXC * C::FC; idx += NTHREADS) {
    const int k = idx / (C::NPXC * C::FC), q = idx - k * (C::NPXC * C::FC), pc = q / C::FC, c = q - pc * C::FC;
    const int Y = ty0 + pc / C::TW, X = tx0 + pc % C::TW;
    if (Y < H && X < W)
      *reinterpret_cast<FragT*>(V + k * vstride + img + ((size_t)Y * W + X) * F + c * 8) =
          *reinterpret_cast<const FragT*>(VT + k * C::VT_ELEMS + pc * 32 + c * 8);
  }
  // (b) pointwise + mix: y = mg*yin + beta2 * ms * sum_k p_k relu(pw_k V_k + bp_k): one 32-pixel tile per wave
  if (wave >= C::NPT_O) return;                       // (wave-uniform; no barrier follows)
  const int r = lane & 31, hh = half;
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
      acc = mma16<T>(load_wfrag<T>(frags, 2 * k + s, lane), lds_chunk<T>(VT + k * C::VT_ELEMS, pc * 32 + (2 * s + hh) * 8), acc);
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
// depthwise backward, lane = channel: g_br = sum_k dw_k^T(GZ_k) (flipped stencils), g_x = gy + ms g_br, g_yin = mg g_x;
// slab tail: dbd[3][32] (pixel sums of GZ_k) | sA[c] = sum g_br mg yin | sB[c] = sum g_x yin   (the dW part of the slab is
// nas_dw_wgrad_kernel's).  The three stencils read three DIFFERENT images here, so a tile is three passes (stage GZ_k,
// slide); a lane's 18 partial g_br values (3 units x 6 columns) wait in LDS between the passes, and all its sums
// are plain per-lane running sums -- no cross-lane work until the workgroup's last step.
// grid = (wgs), persistent over tiles.
// ---------------------------------------------------------------------------------------------
template <int KS>
SR_DEV void nas_lc_bwd_unit(float (&g)[6], float& dbsum, const unsigned* base, const unsigned (*wk)[4], int F, int rowpairs) {
  // window rows / pairs of this stencil inside the 7 x 4 window: rows R0 .. R0 + KS - 1, pairs P0 .. P1
  constexpr int R0 = 3 - KS / 2, P0 = KS == 3 ? 1 : 0, P1 = KS == 7 ? 3 : 2, NP = P1 - P0 + 1;
  unsigned win[KS][NP];
#pragma unroll
  for (int ty = 0; ty < KS; ++ty)
#pragma unroll
    for (int t = 0; t < NP; ++t) win[ty][t] = base[((R0 + ty) * rowpairs + P0 + t) * F];
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    float z = g[s];
#pragma unroll
    for (int ty = 0; ty < KS; ++ty)
#pragma unroll
      for (int t = 0; t < NP; ++t) z = nas_dot2(win[ty][t], wk[ty][t], z);
    g[s] = z;
    // the centre pixel (window row 3, column 3 = high half of pair 1) of this output: its sum over pixels is the bias gradient
    dbsum += __builtin_bit_cast(float, win[KS / 2][1 - P0] & 0xffff0000u);
    if (s + 1 < 6) {
#pragma unroll
      for (int ty = 0; ty < KS; ++ty) {
#pragma unroll
        for (int t = 0; t + 1 < NP; ++t) win[ty][t] = win[ty][t + 1];
        win[ty][NP - 1] = base[((R0 + ty) * rowpairs + P0 + NP + s) * F];
      }
    }
  }
}

// (twelve waves, two row units each: round 3; eight waves of three units before)
constexpr int NAS_DW_BWD_THREADS = 768;
template <int F>
__global__ __launch_bounds__(NAS_DW_BWD_THREADS) void nas_dw_bwd_lc_kernel(const __bf16* __restrict__ yin, const __bf16* __restrict__ GZ,
                                                            const __bf16* __restrict__ gy, __bf16* __restrict__ gyin,
                                                            const float* __restrict__ dwp, float* __restrict__ partial, int N, int H,
                                                            int W, int tiles_x, int tiles_per_img, long vstride) {
  typedef NasCfg<F> C;
  typedef NasLcCfg<F> D;
  constexpr int NT = NAS_DW_BWD_THREADS, NWV = NT / 64, UPW = D::NUNIT / NWV;
  static_assert(D::UW / 2 == 6 && D::NUNIT == NWV * UPW, "units of six columns, the same number per wave");
  __shared__ __attribute__((aligned(16))) unsigned XE[D::XP_DW];
  __shared__ __attribute__((aligned(16))) unsigned XO[D::XP_DW];
  __shared__ __attribute__((aligned(16))) unsigned WP[D::NWP * 32];
  __shared__ float GB[UPW * 6 * NT];
  const int tid = threadIdx.x, lane = tid & 63, ch = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  SR_STAMP_DECL;
  SR_STAMP();
  nas_pack_weights<F, NT>(WP, dwp, tid, true);        // flipped: the data gradient
  if ((int)blockIdx.x < N * tiles_per_img) {           // the first tile's first image is staged under the same barrier
    const int t0 = blockIdx.x, n0 = t0 / tiles_per_img, tile0 = t0 - n0 * tiles_per_img;
    nas_stage_pairs<F, NT>(XE, XO, GZ + (size_t)n0 * H * W * F, nullptr, H, W, (tile0 / tiles_x) * C::TH, (tile0 % tiles_x) * C::TW, tid);
  }
  __syncthreads();
  // the stencil of a pass is read from the packed table at the start of the pass (one stencil live at a time: with all 83 weight
  // pairs resident the kernel needs 185 VGPRs, two waves per SIMD; so it fits three)
  const float mgc = dwp[C::MG + ch], msc = dwp[C::MS + ch];
  const unsigned* X = half ? XO : XE;
  const bool chan = ch < F;
  float db[3] = {0.f, 0.f, 0.f}, sA = 0.f, sB = 0.f;

  for (int t = blockIdx.x; t < N * tiles_per_img; t += gridDim.x) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
    const size_t img = (size_t)n * H * W * F;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      SR_STAMP();
      if (!(k == 0 && t == (int)blockIdx.x)) {
        __syncthreads();                               // the previous pass is through with the pair images
        nas_stage_pairs<F, NT>(XE, XO, GZ + k * vstride + img, nullptr, H, W, ty0, tx0, tid);
        __syncthreads();
      }
      SR_STAMP();
      const int KSZ = 3 + 2 * k, KNP = k == 0 ? 2 : (k == 1 ? 3 : 4), KOFF = k == 0 ? D::W3P : (k == 1 ? D::W5P : D::W7P);   // (k: unrolled)
      unsigned wk[7][4];                               // rows of 4: the unit routine indexes [ty][t]
#pragma unroll
      for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (r < KSZ && t < KNP) wk[r][t] = WP[(KOFF + r * KNP + t) * 32 + ch];
#pragma unroll 1
      for (int i = 0; i < UPW; ++i) {
        const int u = wave + NWV * i;
        const int oy = u >> 1, ox0 = (u & 1) * D::UW + half;
        const unsigned* base = X + (size_t)((oy * C::PW + ox0) >> 1) * F + (chan ? ch : 0);
        // this lane's six partial g_br values of the unit wait in LDS between the passes ([unit][column][thread]: conflict-free)
        float g[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) g[s] = k == 0 ? 0.f : GB[(i * 6 + s) * NT + tid];
        // last pass: this unit's yin / gy values are fetched before the 7x7 window work, not after it
        __bf16 yq[6], gq[6];
        if (k == 2) {
#pragma unroll
          for (int s = 0; s < 6; ++s) {
            const int Y = ty0 + oy, Xc = tx0 + ox0 + 2 * s;
            const bool ok = chan && Y < H && Xc < W;
            const size_t o = img + ((size_t)(ok ? Y : 0) * W + (ok ? Xc : 0)) * F + (ok ? ch : 0);
            yq[s] = yin[o];
            gq[s] = gy[o];
          }
        }
        if (k == 0) nas_lc_bwd_unit<3>(g, db[0], base, wk, F, C::PW / 2);
        else if (k == 1) nas_lc_bwd_unit<5>(g, db[1], base, wk, F, C::PW / 2);
        else nas_lc_bwd_unit<7>(g, db[2], base, wk, F, C::PW / 2);
        if (k < 2) {
#pragma unroll
          for (int s = 0; s < 6; ++s) GB[(i * 6 + s) * NT + tid] = g[s];
        } else {                                       // last pass: the epilogue of the data gradient for these six pixels
#pragma unroll
          for (int s = 0; s < 6; ++s) {
            const int Y = ty0 + oy, Xc = tx0 + ox0 + 2 * s;
            if (chan && Y < H && Xc < W) {
              const size_t o = img + ((size_t)Y * W + Xc) * F + ch;
              const float yv = (float)yq[s], gv = (float)gq[s];
              const float gx = gv + msc * g[s];
              gyin[o] = (__bf16)(mgc * gx);
              sA += g[s] * mgc * yv;
              sB += gx * yv;
            }
          }
        }
      }
    }
  }
  SR_STAMP();
  // ---- workgroup reduction of the per-lane sums: [wave][5][32] in LDS (the pair images are free now) ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(XE);
  const float v[5] = {db[0] + __shfl_xor(db[0], 32), db[1] + __shfl_xor(db[1], 32), db[2] + __shfl_xor(db[2], 32),
                      sA + __shfl_xor(sA, 32), sB + __shfl_xor(sB, 32)};
  if (half == 0) {
#pragma unroll
    for (int j = 0; j < 5; ++j) red[(wave * 5 + j) * 32 + ch] = v[j];
  }
  __syncthreads();
  if (tid < 160) {
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < NWV; ++wv) s += red[wv * 160 + tid];
    partial[(size_t)blockIdx.x * C::DWB_SLAB + 83 * 32 + tid] = s;       // dbd[3][32] | sA[32] | sB[32]
  }
  SR_STAMP();
}
