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
  constexpr bool HOIST = (sizeof(T) == 2);   // bf16 fragments fit in registers; fp32 ones do not
  const T* const wblob0 = wblob;
  for (int pt = wave; pt < C::NPT_H; pt += 4) {
    wblob = weights_for_tile<HOIST>(wblob0);
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
    wblob = weights_for_tile<HOIST>(wblob0);
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

// =============================================================================================
// backward
// =============================================================================================
template <typename C> struct BwdCfg {
  static constexpr int KS3B = (9 * C::FC + 1) / 2;
  static constexpr int KSI = (C::FC + 1) / 2;
  static constexpr int W3T_OFF = C::NFRAG_FWD, W2T_OFF = W3T_OFF + KS3B, W1T_OFF = W2T_OFF + 2 * C::NET,
                       ID_OFF = W1T_OFF + C::KS2, W2N_OFF = ID_OFF + KSI, NFRAG = W2N_OFF + 2 * C::NET;
  static constexpr int B1N_OFF = C::CINIT_FWD;                      // natural-order b1 (only when !FOLD_B1)
  static constexpr int CINIT = C::CINIT_FWD + (C::FOLD_B1 ? 0 : C::NET * 32);
  // LDS images (elements)
  static constexpr int DY_ELEMS = (C::NPXH_PAD + 2) * C::F;         // dy with halo, +2 zero rows of slack
  static constexpr int NPXC = C::TH * C::TW;
  static constexpr int XC_ELEMS = (NPXC + 1) * C::KX;               // core x tile (+1 slack row)
  static constexpr int SCR_ELEMS = 33 * 32;                         // per-wave [32 px][32 ch] scratch
  static constexpr int SLAB_A = 2 * C::NET * 1024 + C::NET * 32 + 32;
  static constexpr int SLAB_B = 9 * 1024;
};

template <typename T, typename C>
SR_DEV void stage_dy_halo(T* DYs, const T* __restrict__ din, int H, int W, int ty0, int tx0, int tid) {
  typedef typename FragOf<T>::type FragT;
  for (int idx = tid; idx < (C::NPXH_PAD + 2) * C::FC; idx += 256) {
    const int hp = idx / C::FC, c = idx - hp * C::FC;
    FragT v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (T)0.f;
    if (hp < C::NPXH) {
      const int hy = hp / C::HW, hx = hp - hy * C::HW;
      const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
      if (Y >= 0 && Y < H && X >= 0 && X < W)
        v = *reinterpret_cast<const FragT*>(din + ((size_t)Y * W + X) * C::F + c * 8);
    }
    *reinterpret_cast<FragT*>(DYs + hp * C::F + c * 8) = v;
  }
}

template <typename T, typename C>
SR_DEV void stage_x_core(T* XC, const T* __restrict__ xin, int H, int W, int ty0, int tx0, int tid) {
  typedef typename FragOf<T>::type FragT;
  constexpr int CHX = C::KX / 8, NPXC = C::TH * C::TW;
  for (int idx = tid; idx < (NPXC + 1) * CHX; idx += 256) {
    const int pc = idx / CHX, c = idx - pc * CHX;
    FragT v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (T)0.f;
    if (c < C::FC) {
      if (pc < NPXC) {
        const int Y = ty0 + pc / C::TW, X = tx0 + pc % C::TW;
        if (Y < H && X < W) v = *reinterpret_cast<const FragT*>(xin + ((size_t)Y * W + X) * C::F + c * 8);
      }
    } else if (C::FOLD_B1 && c == C::FC) {
      v[0] = (T)1.f;
    }
    *reinterpret_cast<FragT*>(XC + pc * C::KX + c * 8) = v;
  }
}

// dt^T[l, px] = sum_{u,f} W3[f,l,8-u] dy[px + u - 1, f]   (rows l in regs, pixels on lanes)
template <typename T, typename C>
SR_DEV f32x16 dt_tile(const T* DYs, const T* __restrict__ wblob, int hbase, int lane) {
  typedef BwdCfg<C> B;
  const int hh = lane >> 5;
  f32x16 acc = zero16();
#pragma unroll
  for (int s = 0; s < B::KS3B; ++s) {
    const int q = 2 * s + hh;
    int off = hbase * C::F;
    if (q < 9 * C::FC) {
      const int u = q / C::FC, c = q - u * C::FC;
      off = (hbase + (u / 3) * C::HW + (u % 3)) * C::F + c * 8;
    }
    acc = mma16<T>(load_wfrag<T>(wblob, B::W3T_OFF + s, lane), lds_chunk<T>(DYs, off), acc);
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// backward-data: dx = dy + W1^T [ 1(h>0) * W2^T conv3x3^T(dy; W3) ],  h recomputed from x.
// grid = (tiles, N), block 256
// ---------------------------------------------------------------------------------------------
template <typename T, int F, int E, int L>
__global__ __launch_bounds__(256) void wdsr_block_bwd_data_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, const T* __restrict__ wblob,
    const float* __restrict__ cinit, int H, int W, int tiles_x) {
  typedef BlockCfg<F, E, L> C;
  typedef BwdCfg<C> B;
  typedef typename FragOf<T>::type FragT;
  typedef typename FragOf<T>::half_type HalfT;
  __shared__ __attribute__((aligned(16))) T smem[B::DY_ELEMS + B::XC_ELEMS];
  T* const DYs = smem;
  T* const XC = smem + B::DY_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * F;
  stage_dy_halo<T, C>(DYs, dy + img, H, W, ty0, tx0, tid);
  stage_x_core<T, C>(XC, x + img, H, W, ty0, tx0, tid);
  __syncthreads();

  constexpr bool HOIST = (sizeof(T) == 2) && (F <= 24);
  const T* const wblob0 = wblob;
  for (int ot = wave; ot < C::NPT_O; ot += 4) {
    wblob = weights_for_tile<HOIST>(wblob0);
    const int oy = (ot / (C::TW / 8)) * 4 + (r >> 3), ox = (ot % (C::TW / 8)) * 8 + (r & 7);
    const int hbase = oy * C::HW + ox, pc = oy * C::TW + ox;
    const f32x16 dtacc = dt_tile<T, C>(DYs, wblob, hbase, lane);
    const FragT dtb0 = acc_to_frag<T, 0>(dtacc), dtb1 = acc_to_frag<T, 1>(dtacc);
    FragT xb[C::KS1];
#pragma unroll
    for (int s = 0; s < C::KS1; ++s) xb[s] = lds_chunk<T>(XC, pc * C::KX + (2 * s + hh) * 8);
    f32x16 dxacc = zero16();
#pragma unroll
    for (int et = 0; et < C::NET; ++et) {
      f32x16 hacc = C::FOLD_B1 ? zero16() : load_cinit(cinit + 32 + et * 32, hh);
#pragma unroll
      for (int s = 0; s < C::KS1; ++s)
        hacc = mma16<T>(load_wfrag<T>(wblob, C::W1_OFF + et * C::KS1 + s, lane), xb[s], hacc);
      f32x16 dh = zero16();
      dh = mma16<T>(load_wfrag<T>(wblob, B::W2T_OFF + 2 * et, lane), dtb0, dh);
      dh = mma16<T>(load_wfrag<T>(wblob, B::W2T_OFF + 2 * et + 1, lane), dtb1, dh);
#pragma unroll
      for (int i = 0; i < 16; ++i) dh[i] = hacc[i] > 0.f ? dh[i] : 0.f;
      if (2 * et < C::KS2)
        dxacc = mma16<T>(load_wfrag<T>(wblob, B::W1T_OFF + 2 * et, lane), acc_to_frag<T, 0>(dh), dxacc);
      if (2 * et + 1 < C::KS2)
        dxacc = mma16<T>(load_wfrag<T>(wblob, B::W1T_OFF + 2 * et + 1, lane), acc_to_frag<T, 1>(dh), dxacc);
    }
#pragma unroll
    for (int s = 0; s < B::KSI; ++s) {
      int c = 2 * s + hh;
      if (c >= C::FC) c = 0;
      dxacc = mma16<T>(load_wfrag<T>(wblob, B::ID_OFF + s, lane),
                       lds_chunk<T>(DYs, (hbase + C::HW + 1) * C::F + c * 8), dxacc);
    }
    const int Y = ty0 + oy, X = tx0 + ox;
    if (Y < H && X < W) {
      T* o = dx + img + ((size_t)Y * W + X) * F;
#pragma unroll
      for (int g = 0; g < C::FC; ++g) *reinterpret_cast<HalfT*>(o + g * 8 + hh * 4) = acc_group<T>(dxacc, g);
    }
  }
}

// write a 32-row accumulator tile (rows = channels, lanes = pixels) as [px r][32 ch] into a wave scratch
template <typename T>
SR_DEV void scratch_store(T* scr, const f32x16& acc, bool valid, int r, int hh) {
  typedef typename FragOf<T>::half_type HalfT;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    HalfT v = acc_group<T>(acc, g);
    if (!valid) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (T)0.f;
    }
    *reinterpret_cast<HalfT*>(scr + r * 32 + g * 8 + hh * 4) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradients of conv1/conv2 (+ b1, b2): pixels are the contraction index, so tiles are
// formed "pixels in rows" by swapping MFMA operands, and the pixel-major operands (x^T, dt^T) come
// from transposed LDS reads.  Each workgroup walks tiles t = blockIdx.x, += gridDim.x of layer
// blockIdx.y and writes ONE partial slab (layout: packing.block_grad_tables slab A).
// ---------------------------------------------------------------------------------------------
template <typename T, int F, int E, int L>
__global__ __launch_bounds__(256, 1) void wdsr_block_wgrad12_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ wblob,
    const float* __restrict__ cinit, float* __restrict__ partial, int N, int H, int W, int tiles_x,
    int tiles_per_img, long x_ls, long dy_ls, long w_ls, long c_ls) {
  typedef BlockCfg<F, E, L> C;
  typedef BwdCfg<C> B;
  typedef typename FragOf<T>::type FragT;
  constexpr int STAGE_BYTES = (B::DY_ELEMS + B::XC_ELEMS + 4 * B::SCR_ELEMS) * (int)sizeof(T);
  constexpr int SLAB_BYTES = B::SLAB_A * 4;
  constexpr int LDS_BYTES = STAGE_BYTES > SLAB_BYTES ? STAGE_BYTES : SLAB_BYTES;
  __shared__ __attribute__((aligned(16))) char smem_raw[LDS_BYTES];
  T* const DYs = reinterpret_cast<T*>(smem_raw);
  T* const XC = DYs + B::DY_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  T* const scr = XC + B::XC_ELEMS + wave * B::SCR_ELEMS;

  const int layer = blockIdx.y;
  x += (size_t)layer * x_ls; dy += (size_t)layer * dy_ls; wblob += (size_t)layer * w_ls; cinit += (size_t)layer * c_ls;
  const T* const wblob0 = wblob;

  f32x16 dW1T[C::NET], dW2[C::NET];
  float db1[C::NET];
#pragma unroll
  for (int et = 0; et < C::NET; ++et) { dW1T[et] = zero16(); dW2[et] = zero16(); db1[et] = 0.f; }
  f32x16 db2acc = zero16();

  for (int t = blockIdx.x; t < N * tiles_per_img; t += gridDim.x) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
    const size_t img = (size_t)n * H * W * F;
    __syncthreads();
    stage_dy_halo<T, C>(DYs, dy + img, H, W, ty0, tx0, tid);
    stage_x_core<T, C>(XC, x + img, H, W, ty0, tx0, tid);
    __syncthreads();
    for (int ot = wave; ot < C::NPT_O; ot += 4) {
      wblob = weights_for_tile<false>(wblob0);
      const int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
      const int oy = toy + (r >> 3), ox = tox + (r & 7);
      const int hbase = oy * C::HW + ox, pc = oy * C::TW + ox;
      const bool valid = (ty0 + oy < H) && (tx0 + ox < W);
      f32x16 dtacc = dt_tile<T, C>(DYs, wblob, hbase, lane);
      if (!valid) dtacc = zero16();
#pragma unroll
      for (int i = 0; i < 16; ++i) db2acc[i] += dtacc[i];
      scratch_store<T>(scr, dtacc, true, r, hh);
      FragT dtA[2], dtT[2], xA[C::KS1], xT[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        dtA[s] = lds_chunk<T>(scr, r * 32 + (2 * s + hh) * 8);
        dtT[s] = tr_frag<T>(scr, s, lane, [](int p) { return p * 32; });
        xT[s] = tr_frag<T>(XC, s, lane, [=](int p) { return ((toy + (p >> 3)) * C::TW + tox + (p & 7)) * C::KX; });
      }
#pragma unroll
      for (int s = 0; s < C::KS1; ++s) xA[s] = lds_chunk<T>(XC, pc * C::KX + (2 * s + hh) * 8);
#pragma unroll
      for (int et = 0; et < C::NET; ++et) {
        f32x16 h2;
        if (C::FOLD_B1) {
          h2 = zero16();
        } else {
          const float b = cinit[B::B1N_OFF + et * 32 + r];
#pragma unroll
          for (int i = 0; i < 16; ++i) h2[i] = b;
        }
#pragma unroll
        for (int s = 0; s < C::KS1; ++s)
          h2 = mma16<T>(xA[s], load_wfrag<T>(wblob, C::W1_OFF + et * C::KS1 + s, lane), h2);
        f32x16 dh2 = zero16();
#pragma unroll
        for (int s = 0; s < 2; ++s) dh2 = mma16<T>(dtA[s], load_wfrag<T>(wblob, B::W2N_OFF + 2 * et + s, lane), dh2);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          dh2[i] = h2[i] > 0.f ? dh2[i] : 0.f;
          h2[i] = fmaxf(h2[i], 0.f);
          sum += dh2[i];
        }
        db1[et] += sum;
        dW1T[et] = mma16<T>(xT[0], acc_to_frag<T, 0>(dh2), dW1T[et]);
        dW1T[et] = mma16<T>(xT[1], acc_to_frag<T, 1>(dh2), dW1T[et]);
        dW2[et] = mma16<T>(dtT[0], acc_to_frag<T, 0>(h2), dW2[et]);
        dW2[et] = mma16<T>(dtT[1], acc_to_frag<T, 1>(h2), dW2[et]);
      }
    }
  }
  // ---- reduce the 4 waves through an LDS slab, then one coalesced store per workgroup ----
  __syncthreads();
  float* slab = reinterpret_cast<float*>(smem_raw);
  for (int i = tid; i < B::SLAB_A; i += 256) slab[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int et = 0; et < C::NET; ++et) {
    slab_add_tile(slab, et, dW1T[et], lane);
    slab_add_tile(slab, C::NET + et, dW2[et], lane);
    atomicAdd(slab + 2 * C::NET * 1024 + et * 32 + r, db1[et]);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i)
    atomicAdd(slab + 2 * C::NET * 1024 + C::NET * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh, db2acc[i]);
  __syncthreads();
  float* out = partial + ((size_t)layer * gridDim.x + blockIdx.x) * B::SLAB_A;
  for (int i = tid; i < B::SLAB_A; i += 256) out[i] = slab[i];
}

// ---------------------------------------------------------------------------------------------
// weight gradient of the 3x3 conv (+ b3 through t's ones channel): recompute t on the core pixels,
// dW3^T[u][l, f] = sum_px t[px, l] dy[px + u - 1, f]  for the 9 read offsets u (tap = 8 - u).
// Same walk / slab protocol as wgrad12 (slab B = 9 accumulator tiles).
// ---------------------------------------------------------------------------------------------
template <typename T, int F, int E, int L>
__global__ __launch_bounds__(256, 1) void wdsr_block_wgrad3_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ wblob,
    const float* __restrict__ cinit, float* __restrict__ partial, int N, int H, int W, int tiles_x,
    int tiles_per_img, long x_ls, long dy_ls, long w_ls, long c_ls) {
  typedef BlockCfg<F, E, L> C;
  typedef BwdCfg<C> B;
  typedef typename FragOf<T>::type FragT;
  constexpr int STAGE_BYTES = (B::DY_ELEMS + B::XC_ELEMS + 4 * B::SCR_ELEMS) * (int)sizeof(T);
  constexpr int SLAB_BYTES = B::SLAB_B * 4;
  constexpr int LDS_BYTES = STAGE_BYTES > SLAB_BYTES ? STAGE_BYTES : SLAB_BYTES;
  __shared__ __attribute__((aligned(16))) char smem_raw[LDS_BYTES];
  T* const DYs = reinterpret_cast<T*>(smem_raw);
  T* const XC = DYs + B::DY_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  T* const scr = XC + B::XC_ELEMS + wave * B::SCR_ELEMS;

  const int layer = blockIdx.y;
  x += (size_t)layer * x_ls; dy += (size_t)layer * dy_ls; wblob += (size_t)layer * w_ls; cinit += (size_t)layer * c_ls;
  const T* const wblob0 = wblob;

  f32x16 dW3T[9];
#pragma unroll
  for (int u = 0; u < 9; ++u) dW3T[u] = zero16();

  for (int t = blockIdx.x; t < N * tiles_per_img; t += gridDim.x) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
    const size_t img = (size_t)n * H * W * F;
    __syncthreads();
    stage_dy_halo<T, C>(DYs, dy + img, H, W, ty0, tx0, tid);
    stage_x_core<T, C>(XC, x + img, H, W, ty0, tx0, tid);
    __syncthreads();
    for (int ot = wave; ot < C::NPT_O; ot += 4) {
      wblob = weights_for_tile<false>(wblob0);
      const int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
      const int oy = toy + (r >> 3), ox = tox + (r & 7);
      const int pc = oy * C::TW + ox;
      const bool valid = (ty0 + oy < H) && (tx0 + ox < W);
      FragT xb[C::KS1];
#pragma unroll
      for (int s = 0; s < C::KS1; ++s) xb[s] = lds_chunk<T>(XC, pc * C::KX + (2 * s + hh) * 8);
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
      scratch_store<T>(scr, tacc, valid, r, hh);
      FragT tT[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) tT[s] = tr_frag<T>(scr, s, lane, [](int p) { return p * 32; });
#pragma unroll
      for (int u = 0; u < 9; ++u) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          FragT d = tr_frag<T>(DYs, s, lane, [=](int p) {
            return ((toy + (p >> 3) + u / 3) * C::HW + tox + (p & 7) + u % 3) * C::F;
          });
          dW3T[u] = mma16<T>(tT[s], d, dW3T[u]);
        }
      }
    }
  }
  __syncthreads();
  float* slab = reinterpret_cast<float*>(smem_raw);
  for (int i = tid; i < B::SLAB_B; i += 256) slab[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 9; ++u) slab_add_tile(slab, u, dW3T[u], lane);
  __syncthreads();
  float* out = partial + ((size_t)layer * gridDim.x + blockIdx.x) * B::SLAB_B;
  for (int i = tid; i < B::SLAB_B; i += 256) out[i] = slab[i];
}
