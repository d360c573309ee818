// Head conv (3x3, 3->F) and fused tail (3x3, F->3r^2) + skip (5x5, 3->3r^2) + PixelShuffle(r) + mean,
// forward and backward (gfx950).  Reference ops replaced: BASIC_MODEL.forward, models/basic_wdsr_b.py:85-93
// (x - mean; head :32-42; tail :55-64; skip :66-78; shuf :80-83; + mean :92) and their autograd backward.
//
// The LR image is NCHW fp32 in HBM (the reference's layout) and is staged in LDS as [pixel][4]:
// 3 colours minus the mean (zero outside the image = the conv's zero padding of x - mean) plus a ones
// channel that carries the bias.  One 8-element fragment chunk = two horizontally adjacent pixels.
// The HR output is written straight from the accumulator: with out^T tiles (rows = conv channel
// c*r*r + i*r + j, lanes = LR pixel) the 4 registers of a group are the r = 4 sub-columns j of one HR row,
// so PixelShuffle is pure addressing (no intermediate buffer, no cross-lane traffic).
#pragma once
#include "wdsr_block.h"

template <int F_, int R_>
struct EndsCfg {
  static constexpr int F = F_, R = R_;
  static constexpr int CO = 3 * R * R, NT = (CO + 31) / 32, COP = (CO + 7) / 8 * 8, CC = COP / 8, FC = F / 8;
  static constexpr int KST = (9 * FC + 15 + 1) / 2, KSTB = (9 * CC + 1) / 2;
  static constexpr int TH = 12, TW = 24, HW = TW + 2, HH = TH + 2, NPXH = HW * HH, NPXH_PAD = (NPXH + 31) / 32 * 32;
  static constexpr int NPT_O = (TH / 4) * (TW / 8), NPXC = TH * TW;
  static constexpr int FT_ELEMS = (NPXH_PAD + 2) * F;           // feature (or dy) tile with 1-px halo
  static constexpr int DC_ELEMS = (NPXH_PAD + 2) * COP;         // dconv tile with 1-px halo
  static constexpr int DCC_ELEMS = (NPXC + 2) * COP;            // dconv core tile
  static constexpr int DYC_ELEMS = (NPXC + 2) * F;              // dy core tile (head wgrad)
  template <int P> struct Img {                                 // LR image tile with P-px halo
    static constexpr int IW = TW + 2 * P, IH = TH + 2 * P, NPI = IW * IH, ELEMS = (NPI + 8) * 4;
  };
  static constexpr int TAIL_TILES = 14 * NT;                    // 9 taps + 5 skip rows, NT row tiles each
};

// 8 elements from an LDS image at an offset that is only 4-element aligned
template <typename T> SR_DEV typename FragOf<T>::type lds_chunk_half(const T* img, int off) {
  typedef typename FragOf<T>::half_type HalfT;
  const HalfT lo = *reinterpret_cast<const HalfT*>(img + off);
  const HalfT hi = *reinterpret_cast<const HalfT*>(img + off + 4);
  typename FragOf<T>::type f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

// Staging helpers.  Every helper issues the global loads of up to SR_STAGE_BATCH iterations before the first
// LDS store of the batch, so a tile costs a couple of HBM round trips instead of one per iteration.
#define SR_STAGE_BATCH 4

template <typename T, typename E, int P, int NTHREADS = 256>
SR_DEV void stage_img(T* XI, const float* __restrict__ ximg, float mean, int H, int W, int ty0, int tx0, int tid) {
  typedef typename FragOf<T>::half_type HalfT;
  typedef typename E::template Img<P> I;
  const size_t plane = (size_t)H * W;
  constexpr int TOTAL = I::NPI + 8, ITER = (TOTAL + NTHREADS - 1) / NTHREADS;
#pragma unroll
  for (int b0 = 0; b0 < ITER; b0 += SR_STAGE_BATCH) {
    float v[SR_STAGE_BATCH][3];
#pragma unroll
    for (int b = 0; b < SR_STAGE_BATCH; ++b) {
      const int ip = tid + (b0 + b) * NTHREADS;
      v[b][0] = 0.f; v[b][1] = 0.f; v[b][2] = 0.f;
      if (b0 + b < ITER && ip < I::NPI) {
        const int iy = ip / I::IW, ix = ip - iy * I::IW;
        const int Y = ty0 - P + iy, X = tx0 - P + ix;
        if (Y >= 0 && Y < H && X >= 0 && X < W) {
          const size_t o = (size_t)Y * W + X;
          v[b][0] = ximg[o] - mean;
          v[b][1] = ximg[plane + o] - mean;
          v[b][2] = ximg[2 * plane + o] - mean;
        }
      }
    }
#pragma unroll
    for (int b = 0; b < SR_STAGE_BATCH; ++b) {
      const int ip = tid + (b0 + b) * NTHREADS;
      if (b0 + b < ITER && ip < TOTAL) {
        HalfT h;
        h[0] = (T)v[b][0]; h[1] = (T)v[b][1]; h[2] = (T)v[b][2]; h[3] = (T)1.f;
        *reinterpret_cast<HalfT*>(XI + ip * 4) = h;
      }
    }
  }
}

// rows p < NLIVE of a [NROWS][CH] LDS image = pixels (y0 + p / RW, x0 + p % RW) of an NHWC tensor, zero elsewhere
template <typename T, int CH, int RW, int NROWS, int NLIVE, int NTHREADS>
SR_DEV void stage_rows(T* dst, const T* __restrict__ src, int H, int W, int y0, int x0, int tid) {
  typedef typename FragOf<T>::type FragT;
  constexpr int NC = CH / 8, TOTAL = NROWS * NC, ITER = (TOTAL + NTHREADS - 1) / NTHREADS;
#pragma unroll
  for (int b0 = 0; b0 < ITER; b0 += SR_STAGE_BATCH) {
    FragT v[SR_STAGE_BATCH];
#pragma unroll
    for (int b = 0; b < SR_STAGE_BATCH; ++b) {
      const int idx = tid + (b0 + b) * NTHREADS;
      const int p = idx / NC, c = idx - p * NC;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[b][j] = (T)0.f;
      if (b0 + b < ITER && p < NLIVE) {
        const int py = p / RW, px = p - py * RW;
        const int Y = y0 + py, X = x0 + px;
        if (Y >= 0 && Y < H && X >= 0 && X < W) v[b] = *reinterpret_cast<const FragT*>(src + ((size_t)Y * W + X) * CH + c * 8);
      }
    }
#pragma unroll
    for (int b = 0; b < SR_STAGE_BATCH; ++b) {
      const int idx = tid + (b0 + b) * NTHREADS;
      if (b0 + b < ITER && idx < TOTAL) *reinterpret_cast<FragT*>(dst + idx * 8) = v[b];
    }
  }
}

// feature / gradient tile with a 1-pixel halo: [NPXH_PAD + 2][CH] (CH multiple of 8), zero outside
template <typename T, typename E, int CH, int NTHREADS = 256>
SR_DEV void stage_halo(T* dst, const T* __restrict__ src, int H, int W, int ty0, int tx0, int tid) {
  stage_rows<T, CH, E::HW, E::NPXH_PAD + 2, E::NPXH, NTHREADS>(dst, src, H, W, ty0 - 1, tx0 - 1, tid);
}

template <typename T, typename E, int CH, int NTHREADS = 256>
SR_DEV void stage_core(T* dst, const T* __restrict__ src, int H, int W, int ty0, int tx0, int tid) {
  stage_rows<T, CH, E::TW, E::NPXC + 2, E::NPXC, NTHREADS>(dst, src, H, W, ty0, tx0, tid);
}

// un-shuffle the HR gradient (NCHW fp32, N x 3 x RH x RW) into a dconv tile [px][COP]:
// channel c*R*R + i*R + j of LR pixel (Y, X) = dout[c][Y*R + i][X*R + j].  HALO = 1: tile with halo,
// HALO = 0: core tile.
// LOSS folds the trainers' loss into this read (pretrain.py:73-77 L1, train_video_superresolution.py:43-53
// Charbonnier): `dout` is then the network OUTPUT sr, `hr` the target, and the gradient is formed on the fly,
//   LOSS 1:  g = sign(sr - hr) * gscale           (torch's l1_loss backward; gscale = upstream / numel)
//   LOSS 2:  g = (sr - hr) / sqrt((sr - hr)^2 + 1e-12) * gscale
// while `lsum` collects this thread's share of sum |sr - hr| (resp. sum sqrt(d^2 + 1e-12)) over the CORE pixels
// (each HR pixel is a core pixel of exactly one tile): the 14 MB HR gradient never exists in HBM.
struct LossIn {
  const float* hr;
  float gscale;
};
template <int LOSS> SR_DEV float loss_grad(float q, float h, float gscale, float& term) {
  const float d = q - h;
  if constexpr (LOSS == 1) {
    term = fabsf(d);
    return d == 0.f ? 0.f : __builtin_copysignf(gscale, d);
  } else {
    const float s = sqrtf(d * d + 1e-12f);
    term = s;
    return d / s * gscale;
  }
}
template <typename T, typename E, int HALO, int NTHREADS = 256, int LOSS = 0>
SR_DEV void stage_dconv(T* DC, const float* __restrict__ dout, int H, int W, int ty0, int tx0, int tid, LossIn li = LossIn{nullptr, 0.f},
                        float* lsum = nullptr) {
  constexpr int R = E::R;
  constexpr int NPX = HALO ? (E::NPXH_PAD + 2) : (E::NPXC + 2);
  constexpr int NLIVE = HALO ? E::NPXH : E::NPXC;
  constexpr int TWW = HALO ? E::HW : E::TW;
  const unsigned hrw = (unsigned)W * R, plane = (unsigned)H * R * hrw;   // one image: 32-bit offsets from a uniform base
  constexpr int ROWS = 3 * R;                       // (colour, sub-row) pairs per pixel
  constexpr int BATCH = LOSS ? (R == 2 ? 1 : 2) : SR_STAGE_BATCH;   // loads in flight per thread (the loss fold reads two tensors)
  constexpr int TOTAL = NPX * ROWS, ITER = (TOTAL + NTHREADS - 1) / NTHREADS;
#pragma unroll
  for (int b0 = 0; b0 < ITER; b0 += BATCH) {
    float v[BATCH][R];
#pragma unroll
    for (int b = 0; b < BATCH; ++b) {
      const int idx = tid + (b0 + b) * NTHREADS;
      const int p = idx / ROWS, cr = idx - p * ROWS;
      const int c = cr / R, si = cr - c * R;
#pragma unroll
      for (int j = 0; j < R; ++j) v[b][j] = 0.f;
      if (b0 + b < ITER && p < NLIVE) {
        const int py = p / TWW, px = p - py * TWW;
        const int Y = ty0 - HALO + py, X = tx0 - HALO + px;
        if (Y >= 0 && Y < H && X >= 0 && X < W) {
          const unsigned off = c * plane + ((unsigned)Y * R + si) * hrw + (unsigned)X * R;
          const float* s = dout + off;
          if constexpr (R == 4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(s);
            v[b][0] = q[0]; v[b][1] = q[1]; v[b][2] = q[2]; v[b][3] = q[3];
          } else {
#pragma unroll
            for (int j = 0; j < R; ++j) v[b][j] = s[j];
          }
          if constexpr (LOSS != 0) {
            float h[R];
            if constexpr (R == 4) {
              const f32x4 q = *reinterpret_cast<const f32x4*>(li.hr + off);
              h[0] = q[0]; h[1] = q[1]; h[2] = q[2]; h[3] = q[3];
            } else {
#pragma unroll
              for (int j = 0; j < R; ++j) h[j] = li.hr[off + j];
            }
            const bool core = !HALO || (py >= 1 && py <= E::TH && px >= 1 && px <= E::TW);
#pragma unroll
            for (int j = 0; j < R; ++j) {
              float term;
              v[b][j] = loss_grad<LOSS>(v[b][j], h[j], li.gscale, term);
              if (core && lsum) *lsum += term;
            }
          }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < BATCH; ++b) {
      const int idx = tid + (b0 + b) * NTHREADS;
      if (b0 + b < ITER && idx < TOTAL) {
        const int p = idx / ROWS, cr = idx - p * ROWS;
        const int c = cr / R, si = cr - c * R;
        T* d = DC + p * E::COP + c * R * R + si * R;
        if constexpr (R == 4) {
          typename FragOf<T>::half_type hv;
          hv[0] = (T)v[b][0]; hv[1] = (T)v[b][1]; hv[2] = (T)v[b][2]; hv[3] = (T)v[b][3];
          *reinterpret_cast<typename FragOf<T>::half_type*>(d) = hv;
        } else {
#pragma unroll
          for (int j = 0; j < R; ++j) d[j] = (T)v[b][j];
        }
      }
    }
  }
  if constexpr (E::COP > E::CO) {                    // zero the padding channels
    for (int idx = tid; idx < NPX * (E::COP - E::CO); idx += NTHREADS) {
      const int p = idx / (E::COP - E::CO), k = idx - p * (E::COP - E::CO);
      DC[p * E::COP + E::CO + k] = (T)0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// head forward: y[n, Y, X, :] = conv3x3(x - mean) + b   (NCHW fp32 image -> NHWC T features)
// ---------------------------------------------------------------------------------------------
template <typename T, int F>
__global__ __launch_bounds__(256) void sr_head_fwd_kernel(const float* __restrict__ ximg, T* __restrict__ y,
                                                          const T* __restrict__ wblob, float mean, int H, int W,
                                                          int tiles_x) {
  typedef EndsCfg<F, 4> E;
  typedef typename E::template Img<1> I;
  typedef typename FragOf<T>::half_type HalfT;
  __shared__ __attribute__((aligned(16))) T XI[I::ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * E::TH, tx0 = (tile % tiles_x) * E::TW;
  stage_img<T, E, 1>(XI, ximg + (size_t)n * 3 * H * W, mean, H, W, ty0, tx0, tid);
  __syncthreads();
  for (int ot = wave; ot < E::NPT_O; ot += 4) {
    const int oy = (ot / (E::TW / 8)) * 4 + (r >> 3), ox = (ot % (E::TW / 8)) * 8 + (r & 7);
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int q = 2 * s + hh, ky = q >> 1, m = q & 1;
      acc = mma16<T>(load_wfrag<T>(wblob, s, lane), lds_chunk_half<T>(XI, ((oy + ky) * I::IW + ox + 2 * m) * 4), acc);
    }
    const int Y = ty0 + oy, X = tx0 + ox;
    if (Y < H && X < W) {
      T* yo = y + (((size_t)n * H + Y) * W + X) * F;
#pragma unroll
      for (int g = 0; g < E::FC; ++g) stream_store(reinterpret_cast<HalfT*>(yo + g * 8 + hh * 4), acc_group<T>(acc, g));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// tail forward: out = PixelShuffle_R( conv3x3(feat; Wt) + conv5x5(x - mean; Ws) + bt + bs ) + mean
// (the three constants ride on the image's ones channel at the skip's centre tap)
// ---------------------------------------------------------------------------------------------
// eight waves (round 3): the tile's nine pixel tiles are two rounds instead of three, two waves per SIMD (four waves before: one per SIMD)
constexpr int SR_TAIL_FWD_THREADS = 512;
template <typename T, int F, int R>
__global__ __launch_bounds__(SR_TAIL_FWD_THREADS) void sr_tail_fwd_kernel(const T* __restrict__ feat, const float* __restrict__ ximg,
                                                          float* __restrict__ out, const T* __restrict__ wblob,
                                                          float mean, int H, int W, int tiles_x) {
  typedef EndsCfg<F, R> E;
  typedef typename E::template Img<2> I;
  typedef typename FragOf<T>::type FragT;
  __shared__ __attribute__((aligned(16))) T smem[E::FT_ELEMS + I::ELEMS];
  T* const FT = smem;
  constexpr int XI0 = E::FT_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * E::TH, tx0 = (tile % tiles_x) * E::TW;
  stage_halo<T, E, F, SR_TAIL_FWD_THREADS>(FT, feat + (size_t)n * H * W * F, H, W, ty0, tx0, tid);
  stage_img<T, E, 2, SR_TAIL_FWD_THREADS>(smem + XI0, ximg + (size_t)n * 3 * H * W, mean, H, W, ty0, tx0, tid);
  __syncthreads();
  constexpr bool HOIST = (sizeof(T) == 2) && (E::NT * E::KST <= 42);
  const T* const wblob0 = wblob;
  for (int ot = wave; ot < E::NPT_O; ot += SR_TAIL_FWD_THREADS / 64) {
    wblob = weights_for_tile<HOIST>(wblob0);
    const int oy = (ot / (E::TW / 8)) * 4 + (r >> 3), ox = (ot % (E::TW / 8)) * 8 + (r & 7);
    const int hbase = oy * E::HW + ox;
    f32x16 acc[E::NT];
#pragma unroll
    for (int ti = 0; ti < E::NT; ++ti) acc[ti] = zero16();
#pragma unroll
    for (int s = 0; s < E::KST; ++s) {
      const int q = 2 * s + hh;
      FragT b;
      if (2 * s + 1 < 9 * E::FC) {                          // both lane halves read feature chunks
        const int tap = q / E::FC, c = q - tap * E::FC;
        b = lds_chunk<T>(smem, (hbase + (tap / 3) * E::HW + (tap % 3)) * F + c * 8);
      } else {
        int off;
        if (q < 9 * E::FC) {
          const int tap = q / E::FC, c = q - tap * E::FC;
          off = (hbase + (tap / 3) * E::HW + (tap % 3)) * F + c * 8;
        } else {
          int qs = q - 9 * E::FC;
          if (qs >= 15) qs = 0;                             // zero weights there
          const int ky = qs / 3, m = qs - ky * 3;
          off = XI0 + ((oy + ky) * I::IW + ox + 2 * m) * 4;
        }
        b = lds_chunk_half<T>(smem, off);
      }
#pragma unroll
      for (int ti = 0; ti < E::NT; ++ti) acc[ti] = mma16<T>(load_wfrag<T>(wblob, ti * E::KST + s, lane), b, acc[ti]);
    }
    const int Y = ty0 + oy, X = tx0 + ox;
    if (Y < H && X < W) {
      const size_t hrw = (size_t)W * R, plane = (size_t)H * R * hrw;
      float* o = out + (size_t)n * 3 * plane;
#pragma unroll
      for (int ti = 0; ti < E::NT; ++ti) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch0 = 32 * ti + 8 * g + 4 * hh;         // rows ch0 .. ch0+3 live in regs 4g..4g+3
          if constexpr (R == 4) {
            if (ch0 < E::CO) {
              const int c = ch0 >> 4, si = (ch0 >> 2) & 3;
              f32x4 v;
              v[0] = acc[ti][4 * g]; v[1] = acc[ti][4 * g + 1]; v[2] = acc[ti][4 * g + 2]; v[3] = acc[ti][4 * g + 3];
              *reinterpret_cast<f32x4*>(o + c * plane + ((size_t)Y * 4 + si) * hrw + (size_t)X * 4) = v;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ch = ch0 + j;
              if (ch < E::CO) {
                const int c = ch / (R * R), rem = ch - c * R * R, si = rem / R, sj = rem - si * R;
                o[c * plane + ((size_t)Y * R + si) * hrw + (size_t)X * R + sj] = acc[ti][4 * g + j];
              }
            }
          }
        }
      }
    }
  }
}

// sum `v` over the workgroup (plain LDS stores, `red` = NTHREADS / 64 floats) and let thread 0 write it
template <int NTHREADS> SR_DEV void wg_sum_store(float v, float* red, float* out, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int i = 0; i < NTHREADS / 64; ++i) t += red[i];
    *out = t;
  }
}

// ---------------------------------------------------------------------------------------------
// tail backward-data: dfeat[px, f] = sum_{u, ch} Wt[ch, f, 8-u] dconv[px + u - 1, ch]
// ---------------------------------------------------------------------------------------------
template <typename T, int F, int R, int LOSS = 0>
__global__ __launch_bounds__(256) void sr_tail_bwd_data_kernel(const float* __restrict__ dout, T* __restrict__ dfeat,
                                                               const T* __restrict__ wblob, int H, int W,
                                                               int tiles_x, LossIn li) {
  typedef EndsCfg<F, R> E;
  typedef typename FragOf<T>::half_type HalfT;
  __shared__ __attribute__((aligned(16))) T DC[E::DC_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * E::TH, tx0 = (tile % tiles_x) * E::TW;
  li.hr += LOSS ? (size_t)n * 3 * H * R * W * R : 0;
  stage_dconv<T, E, 1, 256, LOSS>(DC, dout + (size_t)n * 3 * H * R * W * R, H, W, ty0, tx0, tid, li, nullptr);
  __syncthreads();
  const T* wt = wblob + (size_t)E::NT * E::KST * 512;        // backward-data section follows the forward one
  constexpr bool HOIST = (sizeof(T) == 2);
  const T* const wt0 = wt;
  for (int ot = wave; ot < E::NPT_O; ot += 4) {
    wt = weights_for_tile<HOIST>(wt0);
    const int oy = (ot / (E::TW / 8)) * 4 + (r >> 3), ox = (ot % (E::TW / 8)) * 8 + (r & 7);
    const int hbase = oy * E::HW + ox;
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < E::KSTB; ++s) {
      const int q = 2 * s + hh;
      int off = hbase * E::COP;
      if (q < 9 * E::CC) {
        const int u = q / E::CC, c = q - u * E::CC;
        off = (hbase + (u / 3) * E::HW + (u % 3)) * E::COP + c * 8;
      }
      acc = mma16<T>(load_wfrag<T>(wt, s, lane), lds_chunk<T>(DC, off), acc);
    }
    const int Y = ty0 + oy, X = tx0 + ox;
    if (Y < H && X < W) {
      T* o = dfeat + (((size_t)n * H + Y) * W + X) * F;
#pragma unroll
      for (int g = 0; g < E::FC; ++g) stream_store(reinterpret_cast<HalfT*>(o + g * 8 + hh * 4), acc_group<T>(acc, g));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// tail weight gradients.  14 waves: wave g < 9 owns tap g of the 3x3 tail conv, wave 9 + ky owns skip row ky;
// each keeps its NT accumulator tiles [conv channel rows, input columns] in registers over every pixel
// tile the workgroup walks (no cross-wave reduction), pixels contracted through transposed LDS reads.
// Slab per workgroup: [TAIL_TILES][16][64]; layout in packing.ends_grad_tables.
// ---------------------------------------------------------------------------------------------

template <typename T, int F, int R, int LOSS = 0>
__global__ __launch_bounds__(896) void sr_tail_wgrad_kernel(const float* __restrict__ dout, const T* __restrict__ feat,
                                                            const float* __restrict__ ximg, float mean,
                                                            float* __restrict__ partial, int N, int H, int W,
                                                            int tiles_x, int tiles_per_img, LossIn li, float* __restrict__ loss_part) {
  typedef EndsCfg<F, R> E;
  typedef typename E::template Img<2> I;
  typedef typename FragOf<T>::type FragT;
  constexpr int NT = E::NT, NTHREADS = 896;
  __shared__ __attribute__((aligned(16))) T smem[E::DCC_ELEMS + E::FT_ELEMS + I::ELEMS];
  T* const DC = smem;
  T* const FT = DC + E::DCC_ELEMS;
  T* const XI = FT + E::FT_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool is_tap = wave < 9;
  const int ty = is_tap ? wave / 3 : wave - 9, tx = is_tap ? wave % 3 : 0;

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = zero16();
  float lsum = 0.f;

  SR_STAMP_DECL;
  SR_STAMP();
  for (int t = blockIdx.x; t < N * tiles_per_img; t += gridDim.x) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * E::TH, tx0 = (tile % tiles_x) * E::TW;
    __syncthreads();
    SR_STAMP();
    LossIn lin = li;
    lin.hr += LOSS ? (size_t)n * 3 * H * R * W * R : 0;
    stage_dconv<T, E, 0, NTHREADS, LOSS>(DC, dout + (size_t)n * 3 * H * R * W * R, H, W, ty0, tx0, tid, lin, &lsum);
    SR_STAMP();
    stage_halo<T, E, F, NTHREADS>(FT, feat + (size_t)n * H * W * F, H, W, ty0, tx0, tid);
    stage_img<T, E, 2, NTHREADS>(XI, ximg + (size_t)n * 3 * H * W, mean, H, W, ty0, tx0, tid);
    SR_STAMP();
    __syncthreads();
    SR_STAMP();
    constexpr int UNR = sizeof(T) == 2 ? 3 : 1;
#pragma unroll UNR
    for (int ot = 0; ot < E::NPT_O; ++ot) {
      const int toy = (ot / (E::TW / 8)) * 4, tox = (ot % (E::TW / 8)) * 8;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        FragT b;
        if (is_tap) b = tr_frag<T>(FT, s, lane, [=](int p) { return ((toy + (p >> 3) + ty) * E::HW + tox + (p & 7) + tx) * F; });
        else b = tr_frag<T>(XI, s, lane, [=](int p) { return ((toy + (p >> 3) + ty) * I::IW + tox + (p & 7)) * 4; });
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
          const FragT a = tr_frag<T>(DC, s, lane, [=](int p) { return ((toy + (p >> 3)) * E::TW + tox + (p & 7)) * E::COP + 32 * ti; });
          acc[ti] = mma16<T>(a, b, acc[ti]);
        }
      }
    }
  }
  SR_STAMP();
  // global tile index: taps (ty*3 + tx)*NT + ti ; skip rows 9*NT + ky*NT + ti
  float* out = partial + (size_t)blockIdx.x * E::TAIL_TILES * 1024;
  const int gbase = is_tap ? (ty * 3 + tx) * NT : (9 + ty) * NT;
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int i = 0; i < 16; ++i) out[((gbase + ti) * 16 + i) * 64 + lane] = acc[ti][i];
  if constexpr (LOSS != 0) wg_sum_store<NTHREADS>(lsum, reinterpret_cast<float*>(smem), loss_part + blockIdx.x, tid);
  SR_STAMP();
}

// ---------------------------------------------------------------------------------------------
// head weight gradient: tiles [ky] = [f rows, (kx, ci) columns], dy0 = gradient w.r.t. the head output.
// 9 waves: wave (ky, grp) owns tile ky over the pixel tiles grp, grp+3, grp+6.
// ---------------------------------------------------------------------------------------------
template <typename T, int F>
__global__ __launch_bounds__(576) void sr_head_wgrad_kernel(const T* __restrict__ dy0, const float* __restrict__ ximg,
                                                            float mean, float* __restrict__ partial, int N, int H,
                                                            int W, int tiles_x, int tiles_per_img) {
  typedef EndsCfg<F, 4> E;
  typedef typename E::template Img<1> I;
  typedef typename FragOf<T>::type FragT;
  constexpr int NTHREADS = 576;
  constexpr int STAGE_BYTES = (E::DYC_ELEMS + I::ELEMS) * (int)sizeof(T);
  constexpr int SLAB_BYTES = 9 * 1024 * 4;
  constexpr int LDS_BYTES = STAGE_BYTES > SLAB_BYTES ? STAGE_BYTES : SLAB_BYTES;
  __shared__ __attribute__((aligned(16))) char smem_raw[LDS_BYTES];
  T* const DY = reinterpret_cast<T*>(smem_raw);
  T* const XI = DY + E::DYC_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ky = wave / 3, grp = wave - ky * 3;
  f32x16 acc = zero16();
  for (int t = blockIdx.x; t < N * tiles_per_img; t += gridDim.x) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * E::TH, tx0 = (tile % tiles_x) * E::TW;
    __syncthreads();
    stage_core<T, E, F, NTHREADS>(DY, dy0 + (size_t)n * H * W * F, H, W, ty0, tx0, tid);
    stage_img<T, E, 1, NTHREADS>(XI, ximg + (size_t)n * 3 * H * W, mean, H, W, ty0, tx0, tid);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ot = grp + 3 * k;
      const int toy = (ot / (E::TW / 8)) * 4, tox = (ot % (E::TW / 8)) * 8;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const FragT a = tr_frag<T>(DY, s, lane, [=](int p) { return ((toy + (p >> 3)) * E::TW + tox + (p & 7)) * F; });
        const FragT b = tr_frag<T>(XI, s, lane, [=](int p) { return ((toy + (p >> 3) + ky) * I::IW + tox + (p & 7)) * 4; });
        acc = mma16<T>(a, b, acc);
      }
    }
  }
  __syncthreads();
  float* slab = reinterpret_cast<float*>(smem_raw);           // [ky][grp][1024]: plain stores, no LDS atomics
  slab_store_tile(slab, wave, acc, lane);
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * 3 * 1024;
  for (int i = tid; i < 3 * 1024; i += NTHREADS) {
    const int k = i >> 10, w = i & 1023;
    out[i] = slab[(3 * k) * 1024 + w] + slab[(3 * k + 1) * 1024 + w] + slab[(3 * k + 2) * 1024 + w];
  }
}

// ---------------------------------------------------------------------------------------------
// tail backward, data and weight gradients in one launch (bf16): both stage the same un-shuffled HR gradient
// tile (14 MB of fp32 per batch), so the fused kernel reads it once and drops a dependent launch.  Per tile
// waves 0..8 first compute dfeat for one 32-pixel tile each (weights staged once per workgroup in LDS), then all
// 14 waves accumulate their weight-gradient tiles as sr_tail_wgrad_kernel does (no barrier in between: both
// only read LDS).  Bit-identical to sr_tail_bwd_data_kernel + sr_tail_wgrad_kernel with the same grid.
// ---------------------------------------------------------------------------------------------
template <typename T, int F, int R, int LOSS = 0>
__global__ __launch_bounds__(896) void sr_tail_bwd_kernel(const float* __restrict__ dout, const T* __restrict__ feat,
                                                          const float* __restrict__ ximg, float mean,
                                                          const T* __restrict__ wblob, T* __restrict__ dfeat,
                                                          float* __restrict__ partial, int N, int H, int W, int tiles_x,
                                                          int tiles_per_img, LossIn li, float* __restrict__ loss_part) {
  typedef EndsCfg<F, R> E;
  typedef typename E::template Img<2> I;
  typedef typename FragOf<T>::type FragT;
  typedef typename FragOf<T>::half_type HalfT;
  static_assert(sizeof(T) == 2, "bf16 only");
  constexpr int NT = E::NT, NTHREADS = 896;
  __shared__ __attribute__((aligned(16))) T smem[E::DC_ELEMS + E::FT_ELEMS + I::ELEMS + E::KSTB * 512];
  T* const DC = smem;                       // dconv with halo [NPXH_PAD + 2][COP]
  T* const FT = DC + E::DC_ELEMS;
  T* const XI = FT + E::FT_ELEMS;
  T* const WL = XI + I::ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const bool is_tap = wave < 9;
  const int ty = is_tap ? wave / 3 : wave - 9, tx = is_tap ? wave % 3 : 0;
  stage_weights<T, NTHREADS>(WL, wblob + (size_t)E::NT * E::KST * 512, E::KSTB, tid);   // backward-data section
  WSrc<T, true> wsrc;
  wsrc.p = WL;

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = zero16();
  __shared__ float lsum_lds[LOSS ? NTHREADS : 1];      // this thread's loss partial lives in LDS: no VGPR to spare here
  if constexpr (LOSS != 0) lsum_lds[tid] = 0.f;

  for (int t = blockIdx.x; t < N * tiles_per_img; t += gridDim.x) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * E::TH, tx0 = (tile % tiles_x) * E::TW;
    __syncthreads();
    LossIn lin = li;
    lin.hr += LOSS ? (size_t)n * 3 * H * R * W * R : 0;
    {
      float lsum = 0.f;
      stage_dconv<T, E, 1, NTHREADS, LOSS>(DC, dout + (size_t)n * 3 * H * R * W * R, H, W, ty0, tx0, tid, lin, &lsum);
      if constexpr (LOSS != 0) lsum_lds[tid] += lsum;
    }
    stage_halo<T, E, F, NTHREADS>(FT, feat + (size_t)n * H * W * F, H, W, ty0, tx0, tid);
    stage_img<T, E, 2, NTHREADS>(XI, ximg + (size_t)n * 3 * H * W, mean, H, W, ty0, tx0, tid);
    __syncthreads();
    if (wave < E::NPT_O) {                  // dfeat[px, f] = sum_{u, ch} Wt[ch, f, 8-u] dconv[px + u - 1, ch]
      const int ot = wave;
      const int oy = (ot / (E::TW / 8)) * 4 + (r >> 3), ox = (ot % (E::TW / 8)) * 8 + (r & 7);
      const int hbase = oy * E::HW + ox;
      f32x16 d = zero16();
      constexpr int UNRD = (NT > 1 && F > 24) ? 3 : E::KSTB;      // x4 / 32 units: keep the operand prefetch short
#pragma unroll UNRD
      for (int s = 0; s < E::KSTB; ++s) {
        const int q = 2 * s + hh;
        int off = hbase * E::COP;
        if (q < 9 * E::CC) {
          const int u = q / E::CC, c = q - u * E::CC;
          off = (hbase + (u / 3) * E::HW + (u % 3)) * E::COP + c * 8;
        }
        d = mma16<T>(wsrc.get(s, lane), lds_chunk<T>(DC, off), d);
      }
      const int Y = ty0 + oy, X = tx0 + ox;
      if (Y < H && X < W) {
        T* o = dfeat + (((size_t)n * H + Y) * W + X) * F;
#pragma unroll
        for (int g = 0; g < E::FC; ++g) stream_store(reinterpret_cast<HalfT*>(o + g * 8 + hh * 4), acc_group<T>(d, g));
      }
    }
    constexpr int UNR = (NT > 1 || (LOSS != 0 && R == 2)) ? 1 : 3;   // x4: two accumulator tiles per wave, 128-register budget
#pragma unroll UNR
    for (int ot = 0; ot < E::NPT_O; ++ot) {
      const int toy = (ot / (E::TW / 8)) * 4, tox = (ot % (E::TW / 8)) * 8;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        FragT b;
        if (is_tap) b = tr_frag<T>(FT, s, lane, [=](int p) { return ((toy + (p >> 3) + ty) * E::HW + tox + (p & 7) + tx) * F; });
        else b = tr_frag<T>(XI, s, lane, [=](int p) { return ((toy + (p >> 3) + ty) * I::IW + tox + (p & 7)) * 4; });
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
          const FragT a = tr_frag<T>(DC, s, lane, [=](int p) { return ((toy + (p >> 3) + 1) * E::HW + tox + (p & 7) + 1) * E::COP + 32 * ti; });
          acc[ti] = mma16<T>(a, b, acc[ti]);
        }
      }
    }
  }
  float* out = partial + (size_t)blockIdx.x * E::TAIL_TILES * 1024;
  const int gbase = is_tap ? (ty * 3 + tx) * NT : (9 + ty) * NT;
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int i = 0; i < 16; ++i) out[((gbase + ti) * 16 + i) * 64 + lane] = acc[ti][i];
  if constexpr (LOSS != 0) wg_sum_store<NTHREADS>(lsum_lds[tid], reinterpret_cast<float*>(smem), loss_part + blockIdx.x, tid);
}
