// 3x3 convolution kernels of the BasicVSR propagation trunk (gfx950).  Reference ops replaced:
// ConvResidualBlocks / ResidualBlockNoBN, models/basicvsr_arch.py:108-147 (identical copies in
// basicvsr_arch_origin.py:98-152, mvvsr_arch.py:112-166): conv3x3(F+3 -> F) + LeakyReLU(0.1), then
// num_block x [x + conv2(relu(conv1(x)))], plain convs with bias.
//
// Activations are NHWC with CI in {24, 32} input channels (the 27-channel concat is stored zero-padded
// to 32) and 24 output channels.  The bias rides on a ones channel (index ONES, a spare slot of the
// 32-wide LDS row) at the centre tap.  Activation: 0 none, 1 ReLU, 2 LeakyReLU(0.1).
#pragma once
#include <type_traits>
#include "wdsr_block.h"
#include "wdsr_fwd_rs.h"
#include "flow_warp.h"

struct C3Cfg {
  static constexpr int CO = 24, COC = 3;                        // output channels / 8-channel chunks
  // 16 x 16 tiles: a 64 x 64 frame is 16 tiles, so a BasicVSR frame step of 8 clips x 2 directions is 256 workgroups = ONE round
  // on 256 CUs (12 x 24 tiles made it 288: a second round for 32 of them)
  static constexpr int TH = 16, TW = 16, HW = TW + 2, HH = TH + 2, NPXH = HW * HH, NPXH_PAD = (NPXH + 31) / 32 * 32;
  static constexpr int NPT_O = (TH / 4) * (TW / 8), NPXC = TH * TW;
  static constexpr int KSF = 18;                                // forward k-steps: 9 taps x 4 chunks of the 32-wide row
  static constexpr int KSB = 14;                                // backward-data k-steps: 9 offsets x 3 chunks of dz
  static constexpr int XH_ELEMS = (NPXH_PAD + 2) * 32;          // input tile with halo, 32 channels per pixel
  static constexpr int XC_ELEMS = (NPXC + 1) * 32;              // core input tile
  static constexpr int DZ_ELEMS = (NPXH_PAD + 2) * CO;          // masked output-gradient tile with halo
};

// Two trunks in ONE launch (the two time directions of a BasicVSR frame step are independent): images [0, n_dir) of the batch use the
// weights at the blob pointer, images [n_dir, ..) those `w_ds` elements behind it (the other trunk's blob, same layout); n_dir = 0: one
// trunk.  The weight-gradient launches split their workgroups the same way (first half of the grid: first trunk's images).
struct C3Dir { long w_ds; int n_dir; };
SR_DEV long c3_dir_off(const C3Dir& d, int n) { return (d.n_dir > 0 && n >= d.n_dir) ? d.w_ds : 0; }

// a region of an NHWC bf16 image (24 channels per pixel) -> LDS rows by LDS-DMA, zero outside the image: CH = 3 chunks per row (48-byte
// rows) or CH = 4 with the fourth chunk = [1, 0, ..] (64-byte rows whose channel 24 is the ones channel that carries the bias).  NROWS
// rows are written, the first NLIVE of them are region pixels (row-major, RW wide, origin (y0, x0)); the destination holds whole 1 KB
// pieces (c3_dma_elems).  Round 3: these tiles went global -> registers -> LDS before (a dependent round trip in front of the first MFMA).
constexpr int c3_dma_elems(int nrows, int ch) { return (nrows * ch + 63) / 64 * 64 * 8; }
template <int CH, int RW, int NLIVE, int NROWS, int NWAVES>
SR_DEV void c3_dma_region(__bf16* dst, const __bf16* __restrict__ img, int H, int W, int y0, int x0, int tid) {
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr int NP = (NROWS * CH + 63) / 64;
  const int lane = tid & 63, wave = tid >> 6;
  const char* const ones = reinterpret_cast<const char*>(g_sr_const_chunks);
  char* const d = reinterpret_cast<char*>(dst);
  for (int p = wave; p < NP; p += NWAVES) {
    const int idx = p * 64 + lane;
    const int pix = idx / CH, c = idx - pix * CH;
    const char* src = ones + 16;                        // zeros
    if (CH == 4 && c == 3) src = ones;
    else if (pix < NLIVE) {
      const int py = pix / RW, px = pix - py * RW;
      const int Y = y0 + py, X = x0 + px;
      if (Y >= 0 && Y < H && X >= 0 && X < W) src = reinterpret_cast<const char*>(img + ((size_t)Y * W + X) * C3Cfg::CO + c * 8);
    }
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(d + p * 1024), 16, 0, 0);
  }
}

// B operand of k-step s of a 3x3 conv whose input sits in LDS as 48-BYTE rows (the 24 real channels: a conflict-free stride for the
// 16-byte reads of 32 consecutive pixels, where 64-byte rows are 4-way conflicted): chunk q = 2 s + hh of the window in (tap, chunk)
// order, four chunks per tap of which the fourth is the ONES chunk -- one constant in LDS (element offset `ones_off` from `base`;
// the bias is the centre tap's weight on it).  `win` = element offset of the window's top-left pixel row, `rw` = region width.
template <typename T> SR_DEV typename FragOf<T>::type c3_frag24(const T* base, int win, int rw, int ones_off, int s, int hh) {
  const int tap = s >> 1, c0 = (2 * s) & 3;                // (s: compile-time after unrolling; chunks c0 and c0 + 1 of tap s / 2)
  const int row = win + ((tap / 3) * rw + (tap % 3)) * C3Cfg::CO;
  const int off = c0 == 0 ? row + hh * 8 : (hh ? ones_off : row + 16);
  return lds_chunk<T>(base, off);
}

template <int ACT> SR_DEV float c3_act(float v) {
  if (ACT == 1) return fmaxf(v, 0.f);
  if (ACT == 2) return v > 0.f ? v : 0.1f * v;
  return v;
}
template <int ACT> SR_DEV float c3_dact(float a) {               // derivative from the SAVED post-activation value
  if (ACT == 1) return a > 0.f ? 1.f : 0.f;
  if (ACT == 2) return a > 0.f ? 1.f : 0.1f;
  return 1.f;
}

// x tile [rows][32]: CI channels from HBM (zero outside the image), zeros up to 32, ones channel at ONES
template <typename T, int CI, int ONES, bool HALO, int NTHREADS>
SR_DEV void c3_stage_x(T* Xs, const T* __restrict__ xin, int H, int W, int ty0, int tx0, int tid) {
  typedef typename FragOf<T>::type FragT;
  constexpr int ROWS = HALO ? C3Cfg::NPXH_PAD + 2 : C3Cfg::NPXC + 1;
  constexpr int LIVE = HALO ? C3Cfg::NPXH : C3Cfg::NPXC;
  constexpr int TWW = HALO ? C3Cfg::HW : C3Cfg::TW;
  constexpr int TOTAL = ROWS * 4, ITER = (TOTAL + NTHREADS - 1) / NTHREADS;
  FragT v[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = tid + it * NTHREADS;
    const int p = idx >> 2, c = idx & 3;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[it][j] = (T)0.f;
    if (idx < TOTAL && p < LIVE && c * 8 < CI) {
      const int py = p / TWW, px = p - py * TWW;
      const int Y = ty0 - (HALO ? 1 : 0) + py, X = tx0 - (HALO ? 1 : 0) + px;
      if (Y >= 0 && Y < H && X >= 0 && X < W) v[it] = *reinterpret_cast<const FragT*>(xin + ((size_t)Y * W + X) * CI + c * 8);
    }
    if (c == ONES / 8) v[it][ONES % 8] = (T)1.f;
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = tid + it * NTHREADS;
    if (idx < TOTAL) *reinterpret_cast<FragT*>(Xs + idx * 8) = v[it];
  }
}

// ---------------------------------------------------------------------------------------------
// The first conv's input gathered on the fly (f1): the reference builds it per frame and direction as
//   feat_prop = flow_warp(feat_prop, flow.permute(0, 2, 3, 1)); feat_prop = torch.cat([x_i, feat_prop], dim=1)
// (models/basicvsr_arch.py:74-76,85-87 / mvvsr_arch.py:79-81,90-92).  Here the 3 frame channels and the 24 bilinear
// samples of the recurrent state land straight in the conv's LDS tile [rows][32] = frame 0..2 | state 3..26 | ones 27:
// no warped tensor, no concat, no NCHW -> NHWC copy.  The blend is flow_warp_fwd_kernel's, term by term.
// ---------------------------------------------------------------------------------------------
template <typename T> struct C3WarpSrc {
  const float* frame;      // [N][3][H][W] fp32, batch stride frame_bs elements
  const T* state;          // [N][H][W][24] (the previous call's output in the hot layout); nullptr = zero state
  const float* flow;       // [N][2][H][W] fp32: x then y displacement in pixels, batch stride flow_bs; nullptr = no warp
  long frame_bs, flow_bs;
  T* x0_save;              // forward: the gathered 32-channel input [N][H][W][32], kept for the weight gradient; or nullptr
};

template <typename T> SR_DEV void c3_warp_chunk(float (&o)[8], const T* __restrict__ st, int W, const WarpTaps& t, int chunk) {
  typedef typename FragOf<T>::type FragT;
  const float w00 = (1.f - t.wx) * (1.f - t.wy), w01 = t.wx * (1.f - t.wy), w10 = (1.f - t.wx) * t.wy, w11 = t.wx * t.wy;
  const T* p00 = st + ((size_t)t.y0 * W + t.x0) * C3Cfg::CO + chunk * 8;
  FragT a, b, c, d;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (T)0.f; b[j] = (T)0.f; c[j] = (T)0.f; d[j] = (T)0.f; }
  if (t.vy0 && t.vx0) a = *reinterpret_cast<const FragT*>(p00);
  if (t.vy0 && t.vx1) b = *reinterpret_cast<const FragT*>(p00 + C3Cfg::CO);
  if (t.vy1 && t.vx0) c = *reinterpret_cast<const FragT*>(p00 + (size_t)W * C3Cfg::CO);
  if (t.vy1 && t.vx1) d = *reinterpret_cast<const FragT*>(p00 + (size_t)(W + 1) * C3Cfg::CO);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = 0.f;
    if (t.vy0 && t.vx0) v += w00 * (float)a[j];
    if (t.vy0 && t.vx1) v += w01 * (float)b[j];
    if (t.vy1 && t.vx0) v += w10 * (float)c[j];
    if (t.vy1 && t.vx1) v += w11 * (float)d[j];
    o[j] = v;
  }
}

SR_DEV WarpTaps c3_taps_of(const float* __restrict__ flow, long flow_bs, int n, int Y, int X, int H, int W) {
  if (!flow) return warp_taps((float)X, (float)Y, H, W);
  const float* f = flow + (size_t)n * flow_bs + (size_t)Y * W + X;
  return warp_taps(warp_pos((float)X + f[0], W), warp_pos((float)Y + f[(size_t)H * W], H), H, W);
}

template <typename T, bool HALO, int NTHREADS>
SR_DEV void c3_stage_x_warp(T* Xs, const C3WarpSrc<T>& s, int n, int H, int W, int ty0, int tx0, int tid) {
  typedef typename FragOf<T>::type FragT;
  constexpr int ROWS = HALO ? C3Cfg::NPXH_PAD + 2 : C3Cfg::NPXC + 1;
  constexpr int LIVE = HALO ? C3Cfg::NPXH : C3Cfg::NPXC;
  constexpr int TWW = HALO ? C3Cfg::HW : C3Cfg::TW;
  constexpr int TOTAL = ROWS * 4;
  const T* st = s.state ? s.state + (size_t)n * H * W * C3Cfg::CO : nullptr;
#pragma unroll 2
  for (int idx = tid; idx < TOTAL; idx += NTHREADS) {
    const int p = idx >> 2, c = idx & 3;
    FragT v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (T)0.f;
    if (p < LIVE) {
      const int py = p / TWW, px = p - py * TWW;
      const int Y = ty0 - (HALO ? 1 : 0) + py, X = tx0 - (HALO ? 1 : 0) + px;
      if (Y >= 0 && Y < H && X >= 0 && X < W) {
        if (c == 0) {
          const float* fr = s.frame + (size_t)n * s.frame_bs + (size_t)Y * W + X;
#pragma unroll
          for (int j = 0; j < 3; ++j) v[j] = (T)fr[(size_t)j * H * W];
        }
        if (st) {                                   // LDS channel 8c + j holds state channel 8c + j - 3
          const WarpTaps t = c3_taps_of(s.flow, s.flow_bs, n, Y, X, H, W);
          float lo[8], hi[8];
          if (c >= 1) {
            c3_warp_chunk<T>(lo, st, W, t, c - 1);
#pragma unroll
            for (int j = 0; j < 3; ++j) v[j] = (T)lo[5 + j];
          }
          if (c <= 2) {
            c3_warp_chunk<T>(hi, st, W, t, c);
#pragma unroll
            for (int j = 3; j < 8; ++j) v[j] = (T)hi[j - 3];
          }
        }
      }
    }
    if (c == 3) v[3] = (T)1.f;                     // the ones channel (27) that carries the bias
    *reinterpret_cast<FragT*>(Xs + idx * 8) = v;
  }
}

// dz tile with halo [NPXH_PAD + 2][24]: dz = dA * act'(A) (zero outside the image)
template <typename T, int ACT, int NTHREADS>
SR_DEV void c3_stage_dz(T* DZ, const T* __restrict__ dA, const T* __restrict__ A, int H, int W, int ty0, int tx0, int tid) {
  typedef typename FragOf<T>::type FragT;
  constexpr int TOTAL = (C3Cfg::NPXH_PAD + 2) * C3Cfg::COC, ITER = (TOTAL + NTHREADS - 1) / NTHREADS;
  FragT g[ITER], a[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = tid + it * NTHREADS;
    const int hp = idx / C3Cfg::COC, c = idx - hp * C3Cfg::COC;
#pragma unroll
    for (int j = 0; j < 8; ++j) { g[it][j] = (T)0.f; a[it][j] = (T)1.f; }
    if (idx < TOTAL && hp < C3Cfg::NPXH) {
      const int hy = hp / C3Cfg::HW, hx = hp - hy * C3Cfg::HW;
      const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
      if (Y >= 0 && Y < H && X >= 0 && X < W) {
        const size_t o = ((size_t)Y * W + X) * C3Cfg::CO + c * 8;
        g[it] = *reinterpret_cast<const FragT*>(dA + o);
        if (ACT != 0) a[it] = *reinterpret_cast<const FragT*>(A + o);
      }
    }
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = tid + it * NTHREADS;
    if (idx < TOTAL) {
      FragT z;
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = (T)((float)g[it][j] * c3_dact<ACT>((float)a[it][j]));
      *reinterpret_cast<FragT*>(DZ + idx * 8) = z;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward: y = act(conv3x3(x) + b) [+ res].  grid = (tiles, N); one wave per 32-pixel output tile.
// ---------------------------------------------------------------------------------------------
template <typename T, int CI, int ONES, int ACT, bool ADD, bool WARP = false>
__global__ __launch_bounds__((64 * C3Cfg::NPT_O)) void c3_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                                     T* __restrict__ y, const T* __restrict__ wblob_,
                                                                     int H, int W, int tiles_x, C3WarpSrc<T> warp, C3Dir dir) {
  static_assert(!WARP || (CI == 32 && ONES == 27), "the gathered input is the 27-channel concat");
  const T* const wblob = wblob_ + c3_dir_off(dir, blockIdx.y);
  typedef C3Cfg C;
  typedef typename FragOf<T>::half_type HalfT;
  constexpr int NTHREADS = 64 * C::NPT_O;
  constexpr bool WLDS = (sizeof(T) == 2);
  __shared__ __attribute__((aligned(16))) T smem[C::XH_ELEMS + (WLDS ? C::KSF * 512 : 8)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  WSrc<T, WLDS> wsrc;
  if constexpr (WLDS) {
    stage_weights<T, NTHREADS>(smem + C::XH_ELEMS, wblob, C::KSF, tid);
    wsrc.p = smem + C::XH_ELEMS;
  } else {
    wsrc.p0 = wblob;
  }
  if constexpr (WARP) c3_stage_x_warp<T, true, NTHREADS>(smem, warp, n, H, W, ty0, tx0, tid);
  else c3_stage_x<T, CI, ONES, true, NTHREADS>(smem, x + (size_t)n * H * W * CI, H, W, ty0, tx0, tid);
  __syncthreads();
  if constexpr (WARP) {
    if (warp.x0_save) {                              // the core of the gathered tile, as the unfused path would have built it
      typedef typename FragOf<T>::type FragT;
      for (int idx = tid; idx < C::NPXC * 4; idx += NTHREADS) {
        const int p = idx >> 2, c = idx & 3, py = p / C::TW, px = p - py * C::TW;
        const int Y = ty0 + py, X = tx0 + px;
        if (Y < H && X < W) {
          FragT v = *reinterpret_cast<const FragT*>(smem + ((py + 1) * C::HW + px + 1) * 32 + c * 8);
          if (c == 3) v[3] = (T)0.f;                 // the ones channel is the staging's, not the tensor's
          *reinterpret_cast<FragT*>(warp.x0_save + (((size_t)n * H + Y) * W + X) * 32 + c * 8) = v;
        }
      }
    }
  }
  wsrc.tile();
  const int ot = wave;
  const int oy = (ot / (C::TW / 8)) * 4 + (r >> 3), ox = (ot % (C::TW / 8)) * 8 + (r & 7);
  const int hbase = oy * C::HW + ox;
  f32x16 acc = zero16();
#pragma unroll
  for (int s = 0; s < C::KSF; ++s) {
    const int q = 2 * s + hh, tap = q >> 2, c = q & 3;
    acc = mma16<T>(wsrc.get(s, lane), lds_chunk<T>(smem, (hbase + (tap / 3) * C::HW + (tap % 3)) * 32 + c * 8), acc);
  }
  const int Y = ty0 + oy, X = tx0 + ox;
  if (Y < H && X < W) {
    const size_t o = (((size_t)n * H + Y) * W + X) * C::CO;
#pragma unroll
    for (int g = 0; g < C::COC; ++g) {
      HalfT v;
      HalfT rv;
      if (ADD) rv = *reinterpret_cast<const HalfT*>(res + o + g * 8 + hh * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float f = c3_act<ACT>(acc[4 * g + j]);
        if (ADD) f += (float)rv[j];
        v[j] = (T)f;
      }
      *reinterpret_cast<HalfT*>(y + o + g * 8 + hh * 4) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward-data: dx = conv3x3^T(dA * act'(A)) [+ add], CI output channels (rows of the tile)
// ---------------------------------------------------------------------------------------------
template <typename T, int CI, int ACT, bool ADD>
__global__ __launch_bounds__((64 * C3Cfg::NPT_O)) void c3_bwd_data_kernel(const T* __restrict__ dA, const T* __restrict__ A,
                                                                          const T* __restrict__ add, T* __restrict__ dx,
                                                                          const T* __restrict__ wblob_, int H, int W,
                                                                          int tiles_x, C3Dir dir) {
  const T* const wblob = wblob_ + c3_dir_off(dir, blockIdx.y);
  typedef C3Cfg C;
  typedef typename FragOf<T>::half_type HalfT;
  constexpr int NTHREADS = 64 * C::NPT_O;
  constexpr bool WLDS = (sizeof(T) == 2);
  __shared__ __attribute__((aligned(16))) T smem[C::DZ_ELEMS + (WLDS ? C::KSB * 512 : 8)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * C::CO;
  const T* wb = wblob + (size_t)C::KSF * 512;                   // backward section follows the forward one
  WSrc<T, WLDS> wsrc;
  if constexpr (WLDS) {
    stage_weights<T, NTHREADS>(smem + C::DZ_ELEMS, wb, C::KSB, tid);
    wsrc.p = smem + C::DZ_ELEMS;
  } else {
    wsrc.p0 = wb;
  }
  c3_stage_dz<T, ACT, NTHREADS>(smem, dA + img, A + img, H, W, ty0, tx0, tid);
  __syncthreads();
  wsrc.tile();
  const int ot = wave;
  const int oy = (ot / (C::TW / 8)) * 4 + (r >> 3), ox = (ot % (C::TW / 8)) * 8 + (r & 7);
  const int hbase = oy * C::HW + ox;
  f32x16 acc = zero16();
#pragma unroll
  for (int s = 0; s < C::KSB; ++s) {
    const int q = 2 * s + hh;
    int off = hbase * C::CO;
    if (q < 27) {
      const int u = q / 3, c = q - u * 3;
      off = (hbase + (u / 3) * C::HW + (u % 3)) * C::CO + c * 8;
    }
    acc = mma16<T>(wsrc.get(s, lane), lds_chunk<T>(smem, off), acc);
  }
  const int Y = ty0 + oy, X = tx0 + ox;
  if (Y < H && X < W) {
    const size_t px = ((size_t)n * H + Y) * W + X;
#pragma unroll
    for (int g = 0; g < CI / 8; ++g) {
      HalfT v = acc_group<T>(acc, g);
      if (ADD) {
        const HalfT a = *reinterpret_cast<const HalfT*>(add + px * C::CO + g * 8 + hh * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (T)(acc[4 * g + j] + (float)a[j]);
      }
      *reinterpret_cast<HalfT*>(dx + px * CI + g * 8 + hh * 4) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient: slab = 9 tiles [u][ci rows, co cols], dW[co, ci, tap] = tile[8 - tap]; the ones
// channel row of the centre tile is db.  Wave u owns tile u and walks all pixel tiles of each tile.
// ---------------------------------------------------------------------------------------------
template <typename T, int CI, int ONES, int ACT, bool WARP = false>
__global__ __launch_bounds__((64 * 9)) void c3_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dA,
                                                            const T* __restrict__ A, float* __restrict__ partial, int N,
                                                            int H, int W, int tiles_x, int tiles_per_img, long x_ls,
                                                            long d_ls, long a_ls, long p_ls, C3WarpSrc<T> warp, int n_dir) {
  typedef C3Cfg C;
  constexpr int NTHREADS = 576;
  // blockIdx.y = layer: several convolutions of the same kind in one launch (element strides between layers)
  x += (size_t)blockIdx.y * x_ls; dA += (size_t)blockIdx.y * d_ls; A += (size_t)blockIdx.y * a_ls;
  partial += (size_t)blockIdx.y * p_ls;
  constexpr int STAGE_BYTES = (C::XC_ELEMS + C::DZ_ELEMS) * (int)sizeof(T);
  constexpr int LDS_BYTES = STAGE_BYTES > 9 * 4096 ? STAGE_BYTES : 9 * 4096;
  __shared__ __attribute__((aligned(16))) char smem_raw[LDS_BYTES];
  T* const XC = reinterpret_cast<T*>(smem_raw);
  T* const DZ = XC + C::XC_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int uy = wave / 3, ux = wave - uy * 3;
  f32x16 acc = zero16();
  // n_dir > 0 (two trunks in one launch): the first half of the workgroups sums over images [0, n_dir), the second half over the rest
  const int half = gridDim.x / 2, second = (n_dir > 0 && (int)blockIdx.x >= half) ? 1 : 0;
  const int t_begin = n_dir > 0 ? second * n_dir * tiles_per_img + ((int)blockIdx.x - second * half) : (int)blockIdx.x;
  const int t_end = n_dir > 0 ? (second ? N : n_dir) * tiles_per_img : N * tiles_per_img;
  const int t_step = n_dir > 0 ? half : (int)gridDim.x;
  for (int t = t_begin; t < t_end; t += t_step) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
    __syncthreads();
    if constexpr (WARP) c3_stage_x_warp<T, false, NTHREADS>(XC, warp, n, H, W, ty0, tx0, tid);
    else c3_stage_x<T, CI, ONES, false, NTHREADS>(XC, x + (size_t)n * H * W * CI, H, W, ty0, tx0, tid);
    c3_stage_dz<T, ACT, NTHREADS>(DZ, dA + (size_t)n * H * W * C::CO, A + (size_t)n * H * W * C::CO, H, W, ty0, tx0, tid);
    __syncthreads();
    constexpr int UNR = sizeof(T) == 2 ? 3 : 1;
#pragma unroll UNR
    for (int ot = 0; ot < C::NPT_O; ++ot) {
      const int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
      auto rowx = [=](int p) { return ((toy + (p >> 3)) * C::TW + tox + (p & 7)) * 32; };
      auto rowd = [=](int p) { return ((toy + (p >> 3) + uy) * C::HW + tox + (p & 7) + ux) * C::CO; };
      acc = mma16<T>(tr_frag<T>(XC, 0, lane, rowx), tr_frag<T>(DZ, 0, lane, rowd), acc);
      acc = mma16<T>(tr_frag<T>(XC, 1, lane, rowx), tr_frag<T>(DZ, 1, lane, rowd), acc);
    }
  }
  __syncthreads();
  float* slab = reinterpret_cast<float*>(smem_raw);
#pragma unroll
  for (int i = 0; i < 16; ++i) slab[(wave * 16 + i) * 64 + lane] = acc[i];
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * 9 * 1024;
  for (int i = tid; i < 9 * 1024; i += NTHREADS) out[i] = slab[i];
}

// =============================================================================================
// One ResidualBlockNoBN per launch (bf16): a per-layer launch on a 64x64 clip batch is 128 workgroups of <1 us
// of work under a ~5 us launch floor, so the two convs of a block share a launch.  conv1 + ReLU runs on the
// tile + 1-pixel halo from x on a 2-pixel halo and hands t to conv2 through LDS; t is still written once (core)
// because backward needs it.  Bit-identical to c3_fwd<ReLU> followed by c3_fwd<none, +res>.
// =============================================================================================
struct C3Pair {
  typedef C3Cfg C;
  static constexpr int W2 = C::TW + 4, H2 = C::TH + 4, NP2 = W2 * H2;       // 20 x 20 = 400
  static constexpr int NPT_H = C::NPXH_PAD / 32;                             // 11 pixel tiles over the 18x18 region
  static constexpr int X2_ELEMS = c3_dma_elems(NP2 + 2, 3), T1_ELEMS = (C::NPXH_PAD + 2) * C::CO;   // 48-byte rows (x: whole DMA pieces)
  static constexpr int G2_ELEMS = c3_dma_elems(NP2 + 2, 3), M1_ELEMS = c3_dma_elems(C::NPXH_PAD + 2, 3);   // (whole DMA pieces)
};

template <typename T>
__global__ __launch_bounds__((64 * C3Pair::NPT_H)) void c3_resblock_fwd_kernel(const T* __restrict__ x, T* __restrict__ tmid,
                                                                               T* __restrict__ y, const T* __restrict__ w1_,
                                                                               const T* __restrict__ w2_, int H, int W,
                                                                               int tiles_x, C3Dir dir) {
  const T* const w1 = w1_ + c3_dir_off(dir, blockIdx.y);
  const T* const w2 = w2_ + c3_dir_off(dir, blockIdx.y);
  typedef C3Cfg C;
  typedef C3Pair P;
  typedef typename FragOf<T>::type FragT;
  typedef typename FragOf<T>::half_type HalfT;
  static_assert(sizeof(T) == 2, "bf16 only (LDS budget)");
  constexpr int NTHREADS = 64 * P::NPT_H;
  __shared__ __attribute__((aligned(16))) T smem[P::X2_ELEMS + P::T1_ELEMS + 2 * C::KSF * 512 + 8];
  T* const X2 = smem;
  T* const T1 = X2 + P::X2_ELEMS;
  T* const WL = T1 + P::T1_ELEMS;
  constexpr int ONES_OFF = P::X2_ELEMS + P::T1_ELEMS + 2 * C::KSF * 512;   // the ones chunk [1, 0 x 7]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * C::CO;
  WSrc<T, true> ws1, ws2;
  ws1.p = WL;
  ws2.p = WL + C::KSF * 512;
  stage_weights<T, NTHREADS>(WL, w1, C::KSF, tid);
  stage_weights<T, NTHREADS>(WL + C::KSF * 512, w2, C::KSF, tid);
  // x on the 2-pixel halo: [NP2 + 2][24] (LDS-DMA); the ones chunk
  c3_dma_region<3, P::W2, P::NP2, P::NP2 + 2, P::NPT_H>(reinterpret_cast<__bf16*>(X2), reinterpret_cast<const __bf16*>(x + img), H, W, ty0 - 2, tx0 - 2, tid);
  if (tid < 8) smem[ONES_OFF + tid] = tid == 0 ? (T)1.f : (T)0.f;
  __syncthreads();

  // ---- conv1 + ReLU on the tile + 1-pixel halo ----
  {
    const int hp1 = wave * 32 + r;
    const bool live = hp1 < C::NPXH;
    const int hp1c = live ? hp1 : 0;
    const int hy = hp1c / C::HW, hx = hp1c - hy * C::HW;
    const int base2 = hy * P::W2 + hx;
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < C::KSF; ++s) {
      acc = mma16<T>(ws1.get(s, lane), c3_frag24<T>(smem, base2 * C::CO, P::W2, ONES_OFF, s, hh), acc);
    }
    if (live) {
      const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
      const bool inimg = (Y >= 0 && Y < H && X >= 0 && X < W);
      const bool core = inimg && hy >= 1 && hy <= C::TH && hx >= 1 && hx <= C::TW;
#pragma unroll
      for (int g = 0; g < C::COC; ++g) {
        HalfT v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = inimg ? (T)c3_act<1>(acc[4 * g + j]) : (T)0.f;
        *reinterpret_cast<HalfT*>(T1 + hp1 * C::CO + g * 8 + hh * 4) = v;
        if (core) *reinterpret_cast<HalfT*>(tmid + img + ((size_t)Y * W + X) * C::CO + g * 8 + hh * 4) = v;
      }
    }
  }
  __syncthreads();

  // ---- conv2 + residual on the core (32 consecutive pixels of the row-major core per wave: conflict-free rows) ----
  if (wave < C::NPT_O) {
    const int pc = wave * 32 + r;
    const int oy = pc / C::TW, ox = pc - oy * C::TW;
    const int hbase = oy * C::HW + ox;
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < C::KSF; ++s) acc = mma16<T>(ws2.get(s, lane), c3_frag24<T>(smem, P::X2_ELEMS + hbase * C::CO, C::HW, ONES_OFF, s, hh), acc);
    const int Y = ty0 + oy, X = tx0 + ox;
    if (Y < H && X < W) {
      const size_t o = img + ((size_t)Y * W + X) * C::CO;
      const T* xr = X2 + ((oy + 2) * P::W2 + ox + 2) * C::CO;
#pragma unroll
      for (int g = 0; g < C::COC; ++g) {
        const HalfT rv = *reinterpret_cast<const HalfT*>(xr + g * 8 + hh * 4);
        HalfT v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (T)(acc[4 * g + j] + (float)rv[j]);
        *reinterpret_cast<HalfT*>(y + o + g * 8 + hh * 4) = v;
      }
    }
  }
}

// =============================================================================================
// TWO ResidualBlockNoBN per launch (bf16, round 3): a C4 frame step is 8 + 8 of these ~9 us launches, each one round of 256
// workgroups and almost all of it launch ramp, staging and drain.  Four convs chained through LDS on shrinking regions of ONE
// common grid (the tile + 4-pixel halo, 24 x 24): conv1 on [1, 23)^2 (16 pixel tiles) -> t_a, conv2 + x on [2, 22)^2 (13) -> y_a
// IN PLACE over x, conv3 on [3, 21)^2 (11) -> t_b over t_a, conv4 + y_a on the core [4, 20)^2 (8) -> y_b: 48 pixel tiles of MFMAs
// where two launches run 38, for one staging, one ramp and one drain.  16 waves, one pixel tile per wave and stage.  t_a, y_a, t_b,
// y_b are written for the backward exactly where the per-block launches write them; the same products in the same order and the
// same bf16 rounding points: bit-identical to two c3_resblock_fwd launches.
// Measured at C4: 16.6 us against 2 x 8.7 us -- 5 %, not the 30 % the launch count suggests: these kernels read BOTH MFMA operands
// from LDS (2 KB per MFMA; a wave has one pixel tile per stage, so register-resident weights would move the same bytes), i.e. a
// stage of 16 pixel tiles is 576 KB of LDS reads = 1.9 us at 128 B/clk, four stages 5.7 us: LDS bandwidth, not launches, is what a
// residual block costs here.  The backward-data kernel therefore stays at one block per launch.
// =============================================================================================
struct C3Quad {
  typedef C3Cfg C;
  static constexpr int GW = C::TW + 8, GH = C::TH + 8, NG = GW * GH;         // 24 x 24 = 576
  static constexpr int G_ELEMS = c3_dma_elems(NG + 2, 3);                   // 48-byte rows, whole DMA pieces
  static constexpr int NWAVES = ((GW - 2) * (GH - 2) + 31) / 32;             // conv1's pixel tiles: 16
  static constexpr int LDS_BYTES = (2 * G_ELEMS + 4 * C::KSF * 512 + 8) * 2;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <typename T>
__global__ __launch_bounds__((64 * C3Quad::NWAVES)) void c3_resblock2_fwd_kernel(const T* __restrict__ x, T* __restrict__ tmid_a,
                                                                               T* __restrict__ y_a, T* __restrict__ tmid_b,
                                                                               T* __restrict__ y_b, const T* __restrict__ w_,
                                                                               long o1, long o2, long o3, long o4, int H, int W,
                                                                               int tiles_x, C3Dir dir) {
  typedef C3Cfg C;
  typedef C3Quad Q;
  typedef typename FragOf<T>::type FragT;
  typedef typename FragOf<T>::half_type HalfT;
  static_assert(sizeof(T) == 2, "bf16 only (LDS budget)");
  constexpr int NTHREADS = 64 * Q::NWAVES;
  __shared__ __attribute__((aligned(16))) T smem[2 * Q::G_ELEMS + 4 * C::KSF * 512 + 8];
  constexpr int ONES_OFF = 2 * Q::G_ELEMS + 4 * C::KSF * 512;               // the ones chunk [1, 0 x 7]
  T* const XB = smem;                       // x on the grid, then y_a in place
  T* const TB = XB + Q::G_ELEMS;            // t_a, then t_b
  T* const WL = TB + Q::G_ELEMS;
  const T* const wbase = w_ + c3_dir_off(dir, blockIdx.y);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * C::CO;
  WSrc<T, true> ws[4];
  const long offs[4] = {o1, o2, o3, o4};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ws[k].p = WL + k * C::KSF * 512;
    stage_weights<T, NTHREADS>(WL + k * C::KSF * 512, wbase + offs[k], C::KSF, tid);
  }
  // x on the grid: [NG + 2][24] (LDS-DMA); the ones chunk
  c3_dma_region<3, Q::GW, Q::NG, Q::NG + 2, Q::NWAVES>(reinterpret_cast<__bf16*>(XB), reinterpret_cast<const __bf16*>(x + img), H, W, ty0 - 4, tx0 - 4, tid);
  if (tid < 8) smem[ONES_OFF + tid] = tid == 0 ? (T)1.f : (T)0.f;
  __syncthreads();

  // one conv stage: output region [LO, GW - LO)^2 of the grid, one pixel tile per wave; IN -> acc; the epilogue is the caller's
  auto conv = [&](auto lo_c, const T* IN, const WSrc<T, true>& wk, int& gp, bool& live, bool& inimg, bool& core, int& Y, int& X) {
    constexpr int LO = decltype(lo_c)::value, RW = Q::GW - 2 * LO, NP = RW * (Q::GH - 2 * LO);
    const int hp = wave * 32 + r;
    live = hp < NP;
    const int hpc = live ? hp : 0;
    const int hy = hpc / RW, hx = hpc - hy * RW;
    const int gy = LO + hy, gx = LO + hx;
    gp = gy * Q::GW + gx;
    Y = ty0 - 4 + gy;
    X = tx0 - 4 + gx;
    inimg = Y >= 0 && Y < H && X >= 0 && X < W;
    core = inimg && gy >= 4 && gy < 4 + C::TH && gx >= 4 && gx < 4 + C::TW;
    const int win = (int)(IN - smem) + (gp - Q::GW - 1) * C::CO;
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < C::KSF; ++s) acc = mma16<T>(wk.get(s, lane), c3_frag24<T>(smem, win, Q::GW, ONES_OFF, s, hh), acc);
    return acc;
  };
  auto relu_stage = [&](auto lo_c, const T* IN, T* OUT, const WSrc<T, true>& wk, T* tmid) {
    constexpr int LO = decltype(lo_c)::value, NT = ((Q::GW - 2 * LO) * (Q::GH - 2 * LO) + 31) / 32;
    if (wave >= NT) return;
    int gp, Y, X;
    bool live, inimg, core;
    const f32x16 acc = conv(lo_c, IN, wk, gp, live, inimg, core, Y, X);
    if (live) {
#pragma unroll
      for (int g = 0; g < C::COC; ++g) {
        HalfT v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = inimg ? (T)c3_act<1>(acc[4 * g + j]) : (T)0.f;
        *reinterpret_cast<HalfT*>(OUT + gp * C::CO + g * 8 + hh * 4) = v;
        if (core) *reinterpret_cast<HalfT*>(tmid + img + ((size_t)Y * W + X) * C::CO + g * 8 + hh * 4) = v;
      }
    }
  };
  auto res_stage = [&](auto lo_c, const T* IN, T* RES, const WSrc<T, true>& wk, T* yout, bool to_lds) {
    constexpr int LO = decltype(lo_c)::value, NT = ((Q::GW - 2 * LO) * (Q::GH - 2 * LO) + 31) / 32;
    if (wave >= NT) return;
    int gp, Y, X;
    bool live, inimg, core;
    const f32x16 acc = conv(lo_c, IN, wk, gp, live, inimg, core, Y, X);
    if (live) {
#pragma unroll
      for (int g = 0; g < C::COC; ++g) {
        const HalfT rv = *reinterpret_cast<const HalfT*>(RES + gp * C::CO + g * 8 + hh * 4);
        HalfT v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = inimg ? (T)(acc[4 * g + j] + (float)rv[j]) : (T)0.f;
        if (to_lds) *reinterpret_cast<HalfT*>(RES + gp * C::CO + g * 8 + hh * 4) = v;    // in place: this pixel's own row
        if (core) *reinterpret_cast<HalfT*>(yout + img + ((size_t)Y * W + X) * C::CO + g * 8 + hh * 4) = v;
      }
    }
  };
  relu_stage(std::integral_constant<int, 1>{}, XB, TB, ws[0], tmid_a);
  __syncthreads();
  res_stage(std::integral_constant<int, 2>{}, TB, XB, ws[1], y_a, true);
  __syncthreads();
  relu_stage(std::integral_constant<int, 3>{}, XB, TB, ws[2], tmid_b);
  __syncthreads();
  res_stage(std::integral_constant<int, 4>{}, TB, XB, ws[3], y_b, false);
}

// backward-data of one ResidualBlockNoBN per launch: gt = conv2^T(g) on the tile + 1-pixel halo (written once for
// the conv1 weight gradient), masked by relu'(t), then ga = conv1^T(.) + g on the core.  Bit-identical to
// c3_bwd_data<none> followed by c3_bwd_data<ReLU, +add>.
template <typename T>
__global__ __launch_bounds__((64 * C3Pair::NPT_H)) void c3_resblock_bwd_data_kernel(
    const T* __restrict__ g, const T* __restrict__ tmid, T* __restrict__ gt, T* __restrict__ ga, const T* __restrict__ w1_,
    const T* __restrict__ w2_, int H, int W, int tiles_x, C3Dir dir) {
  const T* const w1 = w1_ + c3_dir_off(dir, blockIdx.y);
  const T* const w2 = w2_ + c3_dir_off(dir, blockIdx.y);
  typedef C3Cfg C;
  typedef C3Pair P;
  typedef typename FragOf<T>::type FragT;
  typedef typename FragOf<T>::half_type HalfT;
  static_assert(sizeof(T) == 2, "bf16 only (LDS budget)");
  constexpr int NTHREADS = 64 * P::NPT_H;
  __shared__ __attribute__((aligned(16))) T smem[P::G2_ELEMS + 2 * P::M1_ELEMS + 2 * C::KSB * 512];
  T* const G2 = smem;                       // g on the 2-pixel halo [NP2 + 2][24]
  T* const M1 = G2 + P::G2_ELEMS;           // t on the 1-pixel halo [NPXH_PAD + 2][24]
  T* const DZ = M1 + P::M1_ELEMS;           // gt * relu'(t)  [NPXH_PAD + 2][24]
  T* const WL = DZ + P::M1_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * C::CO;
  WSrc<T, true> ws1, ws2;
  ws1.p = WL;
  ws2.p = WL + C::KSB * 512;
  stage_weights<T, NTHREADS>(WL, w1 + (size_t)C::KSF * 512, C::KSB, tid);
  stage_weights<T, NTHREADS>(WL + C::KSB * 512, w2 + (size_t)C::KSF * 512, C::KSB, tid);
  {
    // g on the 2-pixel halo, t on the 1-pixel halo: 48-byte rows by LDS-DMA
    c3_dma_region<3, P::W2, P::NP2, P::NP2 + 2, P::NPT_H>(reinterpret_cast<__bf16*>(G2), reinterpret_cast<const __bf16*>(g + img), H, W, ty0 - 2, tx0 - 2, tid);
    c3_dma_region<3, C::HW, C::NPXH, C::NPXH_PAD + 2, P::NPT_H>(reinterpret_cast<__bf16*>(M1), reinterpret_cast<const __bf16*>(tmid + img), H, W, ty0 - 1, tx0 - 1, tid);
    for (int i = tid; i < (C::NPXH_PAD + 2 - C::NPXH) * C::CO; i += NTHREADS) DZ[C::NPXH * C::CO + i] = (T)0.f;   // slack rows
  }
  __syncthreads();

  // ---- gt = conv2^T(g) on the tile + 1-pixel halo; DZ = gt * relu'(t) ----
  {
    const int hp1 = wave * 32 + r;
    const bool live = hp1 < C::NPXH;
    const int hp1c = live ? hp1 : 0;
    const int hy = hp1c / C::HW, hx = hp1c - hy * C::HW;
    const int base2 = hy * P::W2 + hx;
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < C::KSB; ++s) {
      const int q = 2 * s + hh;
      int off = base2 * C::CO;
      if (q < 27) {
        const int u = q / 3, c = q - u * 3;
        off = (base2 + (u / 3) * P::W2 + (u % 3)) * C::CO + c * 8;
      }
      acc = mma16<T>(ws2.get(s, lane), lds_chunk<T>(G2, off), acc);
    }
    if (live) {
      const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
      const bool inimg = (Y >= 0 && Y < H && X >= 0 && X < W);
      const bool core = inimg && hy >= 1 && hy <= C::TH && hx >= 1 && hx <= C::TW;
#pragma unroll
      for (int gi = 0; gi < C::COC; ++gi) {
        const HalfT v = acc_group<T>(acc, gi);
        const HalfT a = *reinterpret_cast<const HalfT*>(M1 + hp1 * C::CO + gi * 8 + hh * 4);
        HalfT z;
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = inimg ? (T)((float)v[j] * c3_dact<1>((float)a[j])) : (T)0.f;
        *reinterpret_cast<HalfT*>(DZ + hp1 * C::CO + gi * 8 + hh * 4) = z;
        if (core) *reinterpret_cast<HalfT*>(gt + img + ((size_t)Y * W + X) * C::CO + gi * 8 + hh * 4) = v;
      }
    }
  }
  __syncthreads();

  // ---- ga = conv1^T(DZ) + g on the core ----
  if (wave < C::NPT_O) {
    const int ot = wave;
    const int oy = (ot / (C::TW / 8)) * 4 + (r >> 3), ox = (ot % (C::TW / 8)) * 8 + (r & 7);
    const int hbase = oy * C::HW + ox;
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < C::KSB; ++s) {
      const int q = 2 * s + hh;
      int off = hbase * C::CO;
      if (q < 27) {
        const int u = q / 3, c = q - u * 3;
        off = (hbase + (u / 3) * C::HW + (u % 3)) * C::CO + c * 8;
      }
      acc = mma16<T>(ws1.get(s, lane), lds_chunk<T>(DZ, off), acc);
    }
    const int Y = ty0 + oy, X = tx0 + ox;
    if (Y < H && X < W) {
      const size_t o = img + ((size_t)Y * W + X) * C::CO;
      const T* gr = G2 + ((oy + 2) * P::W2 + ox + 2) * C::CO;
#pragma unroll
      for (int gi = 0; gi < C::COC; ++gi) {
        const HalfT a = *reinterpret_cast<const HalfT*>(gr + gi * 8 + hh * 4);
        HalfT v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (T)(acc[4 * gi + j] + (float)a[j]);
        *reinterpret_cast<HalfT*>(ga + o + gi * 8 + hh * 4) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// flow_warp backward for the gathered first conv, WITHOUT float atomics.  g = dx0 (the first conv's input gradient,
// [N][H][W][32]: channels 3..26 are the gradient of the warped state).
//   d state[s] = sum over output pixels o that sampled s of w_tap(o) * g[o]      (gather form: each source pixel walks
//                the window |o - s| <= RW, RW = ceil(max |flow|) + 1 read from a device scalar the caller computed
//                once per clip; the summation order is fixed, so the result is deterministic)
//   d flow[o]  = sum_c g[o][c] * d(bilinear)/d(position)                         (per output pixel, optional)
// One thread per source pixel, 16 x 16 of them per workgroup.
// ---------------------------------------------------------------------------------------------
struct C3WarpBwd {
  static constexpr int TS = 16;                      // 16 x 16 source pixels per workgroup
  static constexpr int RW_LDS = 7;                   // windows up to this radius keep the candidates' taps in LDS
  static constexpr int CW = TS + 2 * RW_LDS, NCAND = CW * CW;
};

template <typename T>
__global__ __launch_bounds__(256) void c3_warp_bwd_kernel(const T* __restrict__ g, C3WarpSrc<T> s, const float* __restrict__ flow_bound,
                                                          T* __restrict__ dstate, float* __restrict__ dflow, long dflow_bs,
                                                          int H, int W, int tiles_x) {
  typedef typename FragOf<T>::type FragT;
  typedef C3WarpBwd B;
  __shared__ int2 tap_xy[B::NCAND];                  // (x0, y0) of every candidate output pixel; x0 = INT_MIN: outside the image
  __shared__ float2 tap_w[B::NCAND];                 // (wx, wy)
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * B::TS, tx0 = (tile % tiles_x) * B::TS;
  const int sy = ty0 + (int)threadIdx.x / B::TS, sx = tx0 + (int)threadIdx.x % B::TS;
  int rw = 0;
  if (s.flow) {
    const float b = *flow_bound;
    rw = (b < 0.f ? 0 : (b > 1e6f ? 1000000 : (int)ceilf(b))) + 1;
  }
  const bool in_lds = rw <= B::RW_LDS;                // uniform over the grid
  const int cw = B::TS + 2 * rw;
  if (in_lds) {                                      // every candidate's taps once per workgroup (two IEEE divisions each)
    for (int i = threadIdx.x; i < cw * cw; i += 256) {
      const int oy = ty0 - rw + i / cw, ox = tx0 - rw + i % cw;
      int2 xy = {INT_MIN, 0};
      float2 wq = {0.f, 0.f};
      if (oy >= 0 && oy < H && ox >= 0 && ox < W) {
        const WarpTaps t = c3_taps_of(s.flow, s.flow_bs, n, oy, ox, H, W);
        xy.x = t.x0; xy.y = t.y0; wq.x = t.wx; wq.y = t.wy;
      }
      tap_xy[i] = xy;
      tap_w[i] = wq;
    }
    __syncthreads();
  }
  const bool live = sy < H && sx < W;
  const int p = sy * W + sx;
  const T* gi = g + (size_t)n * H * W * 32;
  float acc[C3Cfg::CO];
#pragma unroll
  for (int j = 0; j < C3Cfg::CO; ++j) acc[j] = 0.f;
  if (live) {
    const int y_lo = sy - rw < 0 ? 0 : sy - rw, y_hi = sy + rw >= H ? H - 1 : sy + rw;
    const int x_lo = sx - rw < 0 ? 0 : sx - rw, x_hi = sx + rw >= W ? W - 1 : sx + rw;
    for (int oy = y_lo; oy <= y_hi; ++oy)
      for (int ox = x_lo; ox <= x_hi; ++ox) {
        int x0, y0;
        float wx, wy;
        if (in_lds) {
          const int i = (oy - (ty0 - rw)) * cw + ox - (tx0 - rw);
          x0 = tap_xy[i].x; y0 = tap_xy[i].y; wx = tap_w[i].x; wy = tap_w[i].y;
        } else {
          const WarpTaps t = c3_taps_of(s.flow, s.flow_bs, n, oy, ox, H, W);
          x0 = t.x0; y0 = t.y0; wx = t.wx; wy = t.wy;
        }
        const int dy = sy - y0, dx = sx - x0;
        if (dy < 0 || dy > 1 || dx < 0 || dx > 1) continue;
        const float wgt = (dx ? wx : 1.f - wx) * (dy ? wy : 1.f - wy);
        // state channel c sits at g channel 3 + c: the 64-byte row as four aligned chunks, shifted by 3
        const T* row = gi + ((size_t)oy * W + ox) * 32;
        FragT q[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = *reinterpret_cast<const FragT*>(row + c * 8);
#pragma unroll
        for (int j = 0; j < C3Cfg::CO; ++j) acc[j] += wgt * (float)q[(3 + j) >> 3][(3 + j) & 7];
      }
#pragma unroll
    for (int c = 0; c < C3Cfg::COC; ++c) {
      FragT o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (T)acc[c * 8 + j];
      *reinterpret_cast<FragT*>(dstate + ((size_t)n * H * W + p) * C3Cfg::CO + c * 8) = o;
    }
  }
  if (!live) return;

  if (dflow) {                                       // this thread's pixel as an OUTPUT pixel
    float gx = 0.f, gy = 0.f;
    if (s.state && s.flow) {
      const WarpTaps t = c3_taps_of(s.flow, s.flow_bs, n, sy, sx, H, W);
      const T* st = s.state + (size_t)n * H * W * C3Cfg::CO + ((size_t)t.y0 * W + t.x0) * C3Cfg::CO;
      const T* grow = gi + (size_t)p * 32;
      for (int c = 0; c < C3Cfg::CO; ++c) {
        const float gg = (float)grow[3 + c];
        const float v00 = (t.vy0 && t.vx0) ? (float)st[c] : 0.f, v01 = (t.vy0 && t.vx1) ? (float)st[C3Cfg::CO + c] : 0.f;
        const float v10 = (t.vy1 && t.vx0) ? (float)st[(size_t)W * C3Cfg::CO + c] : 0.f;
        const float v11 = (t.vy1 && t.vx1) ? (float)st[(size_t)(W + 1) * C3Cfg::CO + c] : 0.f;
        gx += gg * ((v01 - v00) * (1.f - t.wy) + (v11 - v10) * t.wy);
        gy += gg * ((v10 - v00) * (1.f - t.wx) + (v11 - v01) * t.wx);
      }
    }
    float* df = dflow + (size_t)n * dflow_bs + p;
    df[0] = W > 1 ? gx : 0.f;
    df[(size_t)H * W] = H > 1 ? gy : 0.f;
  }
}
