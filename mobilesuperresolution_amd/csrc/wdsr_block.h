// Fused WDSR-B residual block kernels (gfx950).  Reference op being replaced:
// Block.forward, models/basic_wdsr_b.py:108-144 (wn-1x1 F->E, ReLU, wn-1x1 E->L, wn-3x3 L->F, + x)
// and its autograd backward.  Activations are NHWC; the E-wide intermediate never leaves registers,
// the L-wide one never leaves LDS.
#pragma once
#include "sr_common.h"

// a wave inside its MFMA chains goes first at the issue arbiter; the other waves of the SIMD fill the gaps with their
// epilogue VALU work (two-block backward-data kernel: 12.9 -> 12.6 us; results unchanged)
#ifdef SR_BWD_NO_SETPRIO
#define SR_BWD_PRIO(p) do {} while (0)
#else
#define SR_BWD_PRIO(p) __builtin_amdgcn_s_setprio(p)
#endif

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
  // "dense K" form of the 3x3 conv (csrc/wdsr_fwd_rs.h, packing.py W3D): the three taps of a window row are 3 TD contiguous
  // channels of the t image (TD = L rounded up to a multiple of 4: 20 at 24 units, 28 at 32 units), cut into 4-channel chunks
  // (15 / 21) + a spare one = KPR k-steps per window row (4 / 6); lane half hh of k-step q reads the chunks HALF hh + 2 q and
  // HALF hh + 2 q + 1 (16 contiguous bytes); the bias rides on a "ones" chunk in the last row's chunk slot 3 LCD, which is the SECOND
  // chunk of half 1 in k-step QONE; the residual is the accumulator's initial value.  12 / 18 k-steps where KS3 = 15 / 19.
  static constexpr int TD = (L + 3) / 4 * 4;                       // t channels per LDS row of the dense form
  static constexpr int LCD = TD / 4;                               // 4-channel chunks per tap
  static constexpr int KPR = (3 * LCD + 1 + 3) / 4;                // k-steps per window row
  static constexpr int HALF = 2 * KPR;                             // chunk slots per lane half and row
  static constexpr int QONE = (3 * LCD - HALF) / 2;                // the k-step (within the row) whose half-1 second chunk is slot 3 LCD
  static constexpr bool DENSE3 = 3 * LCD >= HALF && (3 * LCD - HALF) % 2 == 1;
  static constexpr int KS3D = DENSE3 ? 3 * KPR : 0;
  static constexpr int LC = L / 4;                                 // (24 units: LC = LCD)
};

// stage the halo'd x tile [NPXH_PAD][KX] into LDS: zero outside the image, ones channel at index F.
// All global loads are issued before the first LDS store (one HBM round trip, not one per chunk).
template <typename T, typename C, int NTHREADS>
SR_DEV void stage_x_halo(T* Xs, const T* __restrict__ xin, int H, int W, int ty0, int tx0, int tid) {
  typedef typename FragOf<T>::type FragT;
  constexpr int CHX = C::KX / 8, TOTAL = C::NPXH_PAD * CHX, ITER = (TOTAL + NTHREADS - 1) / NTHREADS;
  FragT v[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = tid + it * NTHREADS;
    const int hp = idx / CHX, c = idx - hp * CHX;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[it][j] = (T)0.f;
    if (idx < TOTAL) {
      if (c < C::FC) {
        if (hp < C::NPXH) {
          const int hy = hp / C::HW, hx = hp - hy * C::HW;
          const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
          if (Y >= 0 && Y < H && X >= 0 && X < W)
            v[it] = *reinterpret_cast<const FragT*>(xin + ((size_t)Y * W + X) * C::F + c * 8);
        }
      } else if (C::FOLD_B1 && c == C::FC) {
        v[it][0] = (T)1.f;
      }
    }
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = tid + it * NTHREADS;
    if (idx < TOTAL) *reinterpret_cast<FragT*>(Xs + idx * 8) = v[it];
  }
}

// t^T[l, px] = W2 relu(W1 x + b1) + b2 for one 32-pixel tile whose x fragments are in `xb` (rows l in regs,
// pixels on lanes).  PIPE: the conv1 products of e-tile et+1 are issued BEFORE the convert/ReLU of e-tile et,
// so a wave always has independent MFMAs to issue while its VALU turns the previous accumulator into the next
// B operand (the serial form leaves the matrix pipe idle for the MFMA->VALU latency plus ~16 VALU ops per
// e-tile and per wave).  Same products in the same order either way: results are bit-identical.
template <typename T, typename C, typename WS, bool PIPE>
SR_DEV f32x16 t_from_xb(const typename FragOf<T>::type (&xb)[C::KS1], const WS& wsrc, const float* __restrict__ cinit,
                        int lane) {
  const int hh = lane >> 5;
  f32x16 tacc = load_cinit(cinit, hh);
  auto conv1 = [&](int et) {
    f32x16 hacc = C::FOLD_B1 ? zero16() : load_cinit(cinit + 32 + et * 32, hh);
#pragma unroll
    for (int s = 0; s < C::KS1; ++s) hacc = mma16<T>(wsrc.get(C::W1_OFF + et * C::KS1 + s, lane), xb[s], hacc);
    return hacc;
  };
  if constexpr (PIPE) {
    f32x16 hacc = conv1(0);
#pragma unroll
    for (int et = 0; et < C::NET; ++et) {
      f32x16 hnext = hacc;
      if (et + 1 < C::NET) hnext = conv1(et + 1);
      if (2 * et < C::KS2) tacc = mma16<T>(wsrc.get(C::W2_OFF + 2 * et, lane), acc_to_frag_relu<T, 0>(hacc), tacc);
      if (2 * et + 1 < C::KS2) tacc = mma16<T>(wsrc.get(C::W2_OFF + 2 * et + 1, lane), acc_to_frag_relu<T, 1>(hacc), tacc);
      hacc = hnext;
    }
  } else {
#pragma unroll 1
    for (int et = 0; et < C::NET; ++et) {
      const f32x16 hacc = conv1(et);
      if (2 * et < C::KS2) tacc = mma16<T>(wsrc.get(C::W2_OFF + 2 * et, lane), acc_to_frag_relu<T, 0>(hacc), tacc);
      if (2 * et + 1 < C::KS2) tacc = mma16<T>(wsrc.get(C::W2_OFF + 2 * et + 1, lane), acc_to_frag_relu<T, 1>(hacc), tacc);
    }
  }
  return tacc;
}

// ---------------------------------------------------------------------------------------------
// forward:  y = conv3x3(W3, conv1x1(W2, relu(conv1x1(W1, x) + b1)) + b2) + b3 + x
// grid = (tiles_y * tiles_x, N); one wave per 32-pixel tile of the halo'd region (NPT_H waves).
// ---------------------------------------------------------------------------------------------
template <typename T, int F, int E, int L>
__global__ __launch_bounds__((64 * BlockCfg<F, E, L>::NPT_H)) void wdsr_block_fwd_kernel(
    const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ wblob,
    const float* __restrict__ cinit, int H, int W, int tiles_x, T* __restrict__ tsave) {
  typedef BlockCfg<F, E, L> C;
  typedef typename FragOf<T>::type FragT;
  typedef typename FragOf<T>::half_type HalfT;
  constexpr int NTHREADS = 64 * C::NPT_H;
  SR_STAMP_DECL;
  SR_STAMP();
  constexpr bool WLDS = (sizeof(T) == 2);
  constexpr int TS0 = C::NPXH_PAD * C::KX, W0 = C::NPXH_PAD * (C::KX + C::LP);
  // one LDS array (x tile, t tile, packed weights) so that every fragment address is an offset into it
  __shared__ __attribute__((aligned(16))) T smem[W0 + (WLDS ? C::NFRAG_FWD * 512 : 8)];
  T* const Xs = smem;
  T* const Ts = smem + TS0;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const T* xin = x + (size_t)n * H * W * F;

  WSrc<T, WLDS> wsrc;
  if constexpr (WLDS) {
    stage_weights<T, NTHREADS>(smem + W0, wblob, C::NFRAG_FWD, tid);
    wsrc.p = smem + W0;
  } else {
    wsrc.p0 = wblob;
  }
  stage_x_halo<T, C, NTHREADS>(Xs, xin, H, W, ty0, tx0, tid);
  SR_STAMP();
  __syncthreads();
  SR_STAMP();

  // ---- phase A: t = W2 relu(W1 x + b1) + b2 on every halo'd pixel (zero outside the image) ----
  {
    wsrc.tile();
    const int hp = wave * 32 + r;
    FragT xb[C::KS1];
#pragma unroll
    for (int s = 0; s < C::KS1; ++s) xb[s] = lds_chunk<T>(Xs, hp * C::KX + (2 * s + hh) * 8);
    SR_BWD_PRIO(2);
    const f32x16 tacc = t_from_xb<T, C, WSrc<T, WLDS>, (sizeof(T) == 2)>(xb, wsrc, cinit, lane);
    SR_BWD_PRIO(0);
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
      if (tsave && hp < C::NPXH) {               // keep t of the core pixels for sr_wdsr_block_wgrad_saved (tile-local layout)
        const int hy = hp / C::HW, hx = hp - hy * C::HW;
        if (hy >= 1 && hy <= C::TH && hx >= 1 && hx <= C::TW)
          stream_store(reinterpret_cast<HalfT*>(tsave + (((size_t)n * gridDim.x + tile) * (C::TH * C::TW) + (hy - 1) * C::TW + hx - 1) * C::LP +
                                                g * 8 + hh * 4), v);
      }
    }
  }
  SR_STAMP();
  __syncthreads();
  SR_STAMP();

  // ---- phase B: y = sum_taps W3_tap t(shifted) + b3 (ones channel) + x (identity chunks) ----
  if (wave < C::NPT_O) {
    wsrc.tile();
    const int ot = wave;
    const int oy = (ot / (C::TW / 8)) * 4 + (r >> 3), ox = (ot % (C::TW / 8)) * 8 + (r & 7);
    const int hbase = oy * C::HW + ox;
    f32x16 oacc = zero16();
    SR_BWD_PRIO(2);
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
      oacc = mma16<T>(wsrc.get(C::W3_OFF + s, lane), lds_chunk<T>(smem, off), oacc);
    }
    SR_BWD_PRIO(0);
    const int Y = ty0 + oy, X = tx0 + ox;
    if (Y < H && X < W) {
      T* yo = y + (((size_t)n * H + Y) * W + X) * F;
#pragma unroll
      for (int g = 0; g < C::FC; ++g) stream_store(reinterpret_cast<HalfT*>(yo + g * 8 + hh * 4), acc_group<T>(oacc, g));
    }
  }
  SR_STAMP();
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

// register-staged form of the same tiles: load() issues every global load, store() writes LDS later, so a
// persistent kernel can fetch tile t+1 from HBM while it computes on tile t (double-buffered LDS).
template <typename T, typename C, int NTHREADS> struct BwdTileRegs {
  typedef typename FragOf<T>::type FragT;
  static constexpr int CHX = C::KX / 8, NPXC = C::TH * C::TW;
  static constexpr int TD = (C::NPXH_PAD + 2) * C::FC, TX = (NPXC + 1) * CHX;
  static constexpr int ID = (TD + NTHREADS - 1) / NTHREADS, IX = (TX + NTHREADS - 1) / NTHREADS;
  FragT vd[ID], vx[IX];
  SR_DEV void load(const T* __restrict__ din, const T* __restrict__ xin, int H, int W, int ty0, int tx0, int tid) {
#pragma unroll
    for (int it = 0; it < ID; ++it) {
      const int idx = tid + it * NTHREADS;
      const int hp = idx / C::FC, c = idx - hp * C::FC;
#pragma unroll
      for (int j = 0; j < 8; ++j) vd[it][j] = (T)0.f;
      if (idx < TD && hp < C::NPXH) {
        const int hy = hp / C::HW, hx = hp - hy * C::HW;
        const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
        if (Y >= 0 && Y < H && X >= 0 && X < W) vd[it] = *reinterpret_cast<const FragT*>(din + ((size_t)Y * W + X) * C::F + c * 8);
      }
    }
#pragma unroll
    for (int it = 0; it < IX; ++it) {
      const int idx = tid + it * NTHREADS;
      const int pc = idx / CHX, c = idx - pc * CHX;
#pragma unroll
      for (int j = 0; j < 8; ++j) vx[it][j] = (T)0.f;
      if (idx < TX) {
        if (c < C::FC) {
          if (pc < NPXC) {
            const int Y = ty0 + pc / C::TW, X = tx0 + pc % C::TW;
            if (Y < H && X < W) vx[it] = *reinterpret_cast<const FragT*>(xin + ((size_t)Y * W + X) * C::F + c * 8);
          }
        } else if (C::FOLD_B1 && c == C::FC) {
          vx[it][0] = (T)1.f;
        }
      }
    }
  }
  SR_DEV void store(T* DYs, T* XC, int tid) const {
#pragma unroll
    for (int it = 0; it < ID; ++it) {
      const int idx = tid + it * NTHREADS;
      if (idx < TD) *reinterpret_cast<FragT*>(DYs + idx * 8) = vd[it];
    }
#pragma unroll
    for (int it = 0; it < IX; ++it) {
      const int idx = tid + it * NTHREADS;
      if (idx < TX) *reinterpret_cast<FragT*>(XC + idx * 8) = vx[it];
    }
  }
};

// backward tiles: dy with a 1-pixel halo [NPXH_PAD + 2][F] and the core x tile [NPXC + 1][KX]; every
// global load of both is issued before the first LDS store.
template <typename T, typename C, int NTHREADS>
SR_DEV void stage_bwd_tiles(T* DYs, T* XC, const T* __restrict__ din, const T* __restrict__ xin, int H, int W,
                            int ty0, int tx0, int tid) {
  typedef typename FragOf<T>::type FragT;
  constexpr int CHX = C::KX / 8, NPXC = C::TH * C::TW;
  constexpr int TD = (C::NPXH_PAD + 2) * C::FC, TX = (NPXC + 1) * CHX;
  constexpr int ID = (TD + NTHREADS - 1) / NTHREADS, IX = (TX + NTHREADS - 1) / NTHREADS;
  FragT vd[ID], vx[IX];
#pragma unroll
  for (int it = 0; it < ID; ++it) {
    const int idx = tid + it * NTHREADS;
    const int hp = idx / C::FC, c = idx - hp * C::FC;
#pragma unroll
    for (int j = 0; j < 8; ++j) vd[it][j] = (T)0.f;
    if (idx < TD && hp < C::NPXH) {
      const int hy = hp / C::HW, hx = hp - hy * C::HW;
      const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
      if (Y >= 0 && Y < H && X >= 0 && X < W)
        vd[it] = *reinterpret_cast<const FragT*>(din + ((size_t)Y * W + X) * C::F + c * 8);
    }
  }
  if constexpr (sizeof(T) == 4) {   // fp32 parity mode: retire the dy registers before loading x (register budget)
#pragma unroll
    for (int it = 0; it < ID; ++it) {
      const int idx = tid + it * NTHREADS;
      if (idx < TD) *reinterpret_cast<FragT*>(DYs + idx * 8) = vd[it];
    }
  }
#pragma unroll
  for (int it = 0; it < IX; ++it) {
    const int idx = tid + it * NTHREADS;
    const int pc = idx / CHX, c = idx - pc * CHX;
#pragma unroll
    for (int j = 0; j < 8; ++j) vx[it][j] = (T)0.f;
    if (idx < TX) {
      if (c < C::FC) {
        if (pc < NPXC) {
          const int Y = ty0 + pc / C::TW, X = tx0 + pc % C::TW;
          if (Y < H && X < W) vx[it] = *reinterpret_cast<const FragT*>(xin + ((size_t)Y * W + X) * C::F + c * 8);
        }
      } else if (C::FOLD_B1 && c == C::FC) {
        vx[it][0] = (T)1.f;
      }
    }
  }
  if constexpr (sizeof(T) != 4) {
#pragma unroll
    for (int it = 0; it < ID; ++it) {
      const int idx = tid + it * NTHREADS;
      if (idx < TD) *reinterpret_cast<FragT*>(DYs + idx * 8) = vd[it];
    }
  }
#pragma unroll
  for (int it = 0; it < IX; ++it) {
    const int idx = tid + it * NTHREADS;
    if (idx < TX) *reinterpret_cast<FragT*>(XC + idx * 8) = vx[it];
  }
}

// dt^T[l, px] = sum_{u,f} W3[f,l,8-u] dy[px + u - 1, f]   (rows l in regs, pixels on lanes)
template <typename T, typename C, typename WS, int UNROLL = 64>
SR_DEV f32x16 dt_tile(const T* DYs, const WS& wsrc, int w3t_base, int hbase, int lane) {
  typedef BwdCfg<C> B;
  const int hh = lane >> 5;
  f32x16 acc = zero16();
#pragma unroll UNROLL
  for (int s = 0; s < B::KS3B; ++s) {
    const int q = 2 * s + hh;
    int off = hbase * C::F;
    if (q < 9 * C::FC) {
      const int u = q / C::FC, c = q - u * C::FC;
      off = (hbase + (u / 3) * C::HW + (u % 3)) * C::F + c * 8;
    }
    acc = mma16<T>(wsrc.get(w3t_base + s, lane), lds_chunk<T>(DYs, off), acc);
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// backward-data: dx = dy + W1^T [ 1(h>0) * W2^T conv3x3^T(dy; W3) ],  h recomputed from x.
// grid = (tiles, N); one wave per 32-pixel output tile (NPT_O waves).
// ---------------------------------------------------------------------------------------------
template <typename T, int F, int E, int L>
__global__ __launch_bounds__((64 * BlockCfg<F, E, L>::NPT_O)) void wdsr_block_bwd_data_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, const T* __restrict__ wblob,
    const float* __restrict__ cinit, int H, int W, int tiles_x, T* __restrict__ dtsave) {
  typedef BlockCfg<F, E, L> C;
  typedef BwdCfg<C> B;
  typedef typename FragOf<T>::type FragT;
  typedef typename FragOf<T>::half_type HalfT;
  constexpr int NTHREADS = 64 * C::NPT_O;
  constexpr bool WLDS = (sizeof(T) == 2);
  // LDS weight image: W1 (fragments 0 ..), then the contiguous blob range W3T | W2T | W1T | ID
  constexpr int NW1 = C::NET * C::KS1, NREST = B::W2N_OFF - B::W3T_OFF;
  constexpr int LW3T = WLDS ? NW1 : B::W3T_OFF, LW2T = LW3T + B::KS3B, LW1T = LW2T + 2 * C::NET, LID = LW1T + C::KS2;
  constexpr int W0 = B::DY_ELEMS + B::XC_ELEMS;
  __shared__ __attribute__((aligned(16))) T smem[W0 + (WLDS ? (NW1 + NREST) * 512 : 8)];
  T* const DYs = smem;
  T* const XC = smem + B::DY_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * F;
  WSrc<T, WLDS> wsrc;
  if constexpr (WLDS) {
    stage_weights<T, NTHREADS>(smem + W0, wblob, NW1, tid);
    stage_weights<T, NTHREADS>(smem + W0 + NW1 * 512, wblob + (size_t)B::W3T_OFF * 512, NREST, tid);
    wsrc.p = smem + W0;
  } else {
    wsrc.p0 = wblob;
  }
  stage_bwd_tiles<T, C, NTHREADS>(DYs, XC, dy + img, x + img, H, W, ty0, tx0, tid);
  __syncthreads();

  {
    wsrc.tile();
    const int ot = wave;
    const int oy = (ot / (C::TW / 8)) * 4 + (r >> 3), ox = (ot % (C::TW / 8)) * 8 + (r & 7);
    const int hbase = oy * C::HW + ox, pc = oy * C::TW + ox;
    SR_BWD_PRIO(2);
    const f32x16 dtacc = dt_tile<T, C>(DYs, wsrc, LW3T, hbase, lane);
    if (dtsave) {                                  // keep dt of the core pixels (zero outside the image)
      const bool inimg = (ty0 + oy < H) && (tx0 + ox < W);
      T* o = dtsave + (((size_t)n * gridDim.x + tile) * B::NPXC + pc) * C::LP;
#pragma unroll
      for (int g = 0; g < C::CPT; ++g) {
        HalfT v = acc_group<T>(dtacc, g);
        if (!inimg) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (T)0.f;
        }
        stream_store(reinterpret_cast<HalfT*>(o + g * 8 + hh * 4), v);
      }
    }
    const FragT dtb0 = acc_to_frag<T, 0>(dtacc), dtb1 = acc_to_frag<T, 1>(dtacc);
    FragT xb[C::KS1];
#pragma unroll
    for (int s = 0; s < C::KS1; ++s) xb[s] = lds_chunk<T>(XC, pc * C::KX + (2 * s + hh) * 8);
    f32x16 dxacc = zero16();
#pragma unroll
    for (int et = 0; et < C::NET; ++et) {
      f32x16 hacc = C::FOLD_B1 ? zero16() : load_cinit(cinit + 32 + et * 32, hh);
#pragma unroll
      for (int s = 0; s < C::KS1; ++s) hacc = mma16<T>(wsrc.get(C::W1_OFF + et * C::KS1 + s, lane), xb[s], hacc);
      f32x16 dh = zero16();
      dh = mma16<T>(wsrc.get(LW2T + 2 * et, lane), dtb0, dh);
      dh = mma16<T>(wsrc.get(LW2T + 2 * et + 1, lane), dtb1, dh);
#pragma unroll
      for (int i = 0; i < 16; ++i) dh[i] = hacc[i] > 0.f ? dh[i] : 0.f;
      if (2 * et < C::KS2) dxacc = mma16<T>(wsrc.get(LW1T + 2 * et, lane), acc_to_frag<T, 0>(dh), dxacc);
      if (2 * et + 1 < C::KS2) dxacc = mma16<T>(wsrc.get(LW1T + 2 * et + 1, lane), acc_to_frag<T, 1>(dh), dxacc);
    }
#pragma unroll
    for (int s = 0; s < B::KSI; ++s) {
      int c = 2 * s + hh;
      if (c >= C::FC) c = 0;
      dxacc = mma16<T>(wsrc.get(LID + s, lane), lds_chunk<T>(DYs, (hbase + C::HW + 1) * C::F + c * 8), dxacc);
    }
    SR_BWD_PRIO(0);
    const int Y = ty0 + oy, X = tx0 + ox;
    if (Y < H && X < W) {
      T* o = dx + img + ((size_t)Y * W + X) * F;
#pragma unroll
      for (int g = 0; g < C::FC; ++g) stream_store(reinterpret_cast<HalfT*>(o + g * 8 + hh * 4), acc_group<T>(dxacc, g));
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

// t^T[l, px] = W2 relu(W1 x + b1) + b2 for one 32-pixel tile (rows l in regs, pixels on lanes)
template <typename T, typename C, typename WS, int UNROLL>
SR_DEV f32x16 t_tile(const typename FragOf<T>::type (&xb)[C::KS1], const WS& wsrc, const float* __restrict__ cinit,
                     int lane) {
  return t_from_xb<T, C, WS, (UNROLL > 1)>(xb, wsrc, cinit, lane);
}

// ---------------------------------------------------------------------------------------------
// weight gradients of the block.  Pixels are the contraction index, so accumulator tiles are formed
// "pixels in rows" by swapping MFMA operands and the pixel-major operands (x^T, dt^T, t^T, shifted dy)
// come from transposed LDS reads.  The ~300 accumulator registers one wave would need are split over
// the workgroup, and over two kernels so that each stays far below the 168-register / 3-waves-per-SIMD
// budget (ROCm 7.2 hipcc must never spill here, see sr_common.h):
//   ROLE 0 (2 NET waves): phase 1  waves 0..8: dt = conv3x3^T(dy) of one 32-pixel tile each -> LDS image
//                         phase 2  wave (et, half): dW1^T[et], dW2[et], db1 over every second pixel tile
//   ROLE 1 (9 waves)    : phase 1  t = W2 relu(W1 x + b1) + b2 of one pixel tile each -> LDS image
//                         phase 2  wave u: tap u of dW3^T over all pixel tiles (b3 via t's ones channel)
// Each workgroup walks tiles t = blockIdx.x, += gridDim.x of layer blockIdx.y and writes ONE partial
// slab (layout: packing.block_grad_tables, slab A for ROLE 0, slab B for ROLE 1).
// ---------------------------------------------------------------------------------------------
template <int F, int E, int L, int ROLE> struct WgradCfg {
  typedef BlockCfg<F, E, L> C;
  static constexpr int NWAVES = ROLE == 0 ? 2 * C::NET : 9;          // ROLE 0 needs >= 9 (phase 1)
};

template <typename T, int F, int E, int L, int ROLE>
__global__ __launch_bounds__((64 * WgradCfg<F, E, L, ROLE>::NWAVES)) void wdsr_block_wgrad_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ wblob,
    const float* __restrict__ cinit, float* __restrict__ partial, int N, int H, int W, int tiles_x,
    int tiles_per_img, long x_ls, long dy_ls, long w_ls, long c_ls) {
  typedef BlockCfg<F, E, L> C;
  typedef BwdCfg<C> B;
  typedef WgradCfg<F, E, L, ROLE> G;
  SR_STAMP_DECL;
  SR_STAMP();
  typedef typename FragOf<T>::type FragT;
  constexpr int NTHREADS = 64 * G::NWAVES;
  constexpr bool WLDS = (sizeof(T) == 2);
  constexpr int IMG_ELEMS = (B::NPXC + 1) * 32;                       // dt or t image: [core px][32 ch]
  // LDS weight image.  ROLE 0: W1 | W3T | W2N.  ROLE 1: W1 | W2 (the first fragments of the blob).
  constexpr int NW1 = C::NET * C::KS1;
  constexpr int LW3T = WLDS ? NW1 : B::W3T_OFF, LW2N = WLDS ? NW1 + B::KS3B : B::W2N_OFF;
  constexpr int NWL = !WLDS ? 0 : (ROLE == 0 ? NW1 + B::KS3B + 2 * C::NET : C::W3_OFF);
  constexpr int SLAB = ROLE == 0 ? B::SLAB_A : B::SLAB_B;
  constexpr int TILE_ELEMS = B::DY_ELEMS + B::XC_ELEMS;               // one (dy halo, x core) buffer
  constexpr int CIN_BYTES = ((B::CINIT + 3) / 4) * 16;                // C-init tables, copied to LDS
  // double-buffer the tile staging when two buffers fit beside the image and the weights (bf16 only)
  constexpr bool DBUF = WLDS && !(ROLE == 0 && F > 24) &&   // (F = 32, ROLE 0: the prefetch registers do not fit the 168 budget)
                        ((2 * TILE_ELEMS + IMG_ELEMS + NWL * 512) * (int)sizeof(T) + CIN_BYTES + 256 <= 160 * 1024);
  constexpr int STAGE_ELEMS = (DBUF ? 2 : 1) * TILE_ELEMS + IMG_ELEMS;
  constexpr int STAGE_BYTES = (STAGE_ELEMS + NWL * 512) * (int)sizeof(T) + CIN_BYTES;
  constexpr int RED_BYTES = ROLE == 0 ? (2 * SLAB + 9 * 32) * 4 : 0;   // ROLE 0: two half-slabs + per-wave db2 rows
  constexpr int LDS_BYTES = STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES;
  constexpr bool DB2_REGS = (sizeof(T) == 2);       // fp32 parity mode keeps db2 in LDS (register budget)
  __shared__ __attribute__((aligned(16))) char smem_raw[LDS_BYTES + 128];
  T* const BUF = reinterpret_cast<T*>(smem_raw);                       // [DBUF ? 2 : 1][TILE_ELEMS]
  T* const IMG = BUF + (DBUF ? 2 : 1) * TILE_ELEMS;
  float* const cls = reinterpret_cast<float*>(smem_raw + (STAGE_ELEMS + NWL * 512) * sizeof(T));
  float* const db2lds = reinterpret_cast<float*>(smem_raw + LDS_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  if (tid < 32) db2lds[tid] = 0.f;

  const int layer = blockIdx.y;
  x += (size_t)layer * x_ls; dy += (size_t)layer * dy_ls; wblob += (size_t)layer * w_ls; cinit += (size_t)layer * c_ls;
  for (int i = tid; i < B::CINIT; i += NTHREADS) cls[i] = cinit[i];     // no global load inside the tile loop
  WSrc<T, WLDS> wsrc;
  if constexpr (WLDS) {
    T* wl = BUF + STAGE_ELEMS;
    if constexpr (ROLE == 0) {
      stage_weights<T, NTHREADS>(wl, wblob, NW1, tid);
      stage_weights<T, NTHREADS>(wl + NW1 * 512, wblob + (size_t)B::W3T_OFF * 512, B::KS3B, tid);
      stage_weights<T, NTHREADS>(wl + (NW1 + B::KS3B) * 512, wblob + (size_t)B::W2N_OFF * 512, 2 * C::NET, tid);
    } else {
      stage_weights<T, NTHREADS>(wl, wblob, C::W3_OFF, tid);
    }
    wsrc.p = wl;
  } else {
    wsrc.p0 = wblob;
  }
  // the image's slack row is read (never written) by transposed loads: keep it finite
  if (tid < 32) IMG[B::NPXC * 32 + tid] = (T)0.f;

  f32x16 accA = zero16(), accB = zero16();          // ROLE 0: dW1^T[et], dW2[et].  ROLE 1: accA = dW3^T[tap = wave]
  f32x16 db2acc = zero16();                         // ROLE 0, phase-1 waves
  float db1 = 0.f;
  const int et = wave >> 1, half = wave & 1;

  const int total = N * tiles_per_img;
  BwdTileRegs<T, C, NTHREADS> regs;
  auto fetch = [&](int t) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const size_t img = (size_t)n * H * W * F;
    int tid2 = tid;                                    // recompute the per-thread tile indices every time:
    asm volatile("" : "+v"(tid2));                     // hoisted, they are spilled around the tile loop
    regs.load(dy + img, x + img, H, W, (tile / tiles_x) * C::TH, (tile % tiles_x) * C::TW, tid2);
  };
  int cur = 0;
  if ((int)blockIdx.x < total) fetch(blockIdx.x);
  __syncthreads();
  if ((int)blockIdx.x < total) regs.store(BUF, BUF + B::DY_ELEMS, tid);
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
    const bool has_next = t + (int)gridDim.x < total;
    T* const DYs = BUF + cur * TILE_ELEMS;
    T* const XC = DYs + B::DY_ELEMS;
    SR_STAMP();
    __syncthreads();                                   // tile t staged; IMG free
    SR_STAMP();
    if (DBUF && has_next) fetch(t + gridDim.x);        // HBM loads of tile t+1 fly during the compute below
    SR_STAMP();

    // ---- phase 1: dt (ROLE 0) or t (ROLE 1) of one pixel tile per wave -> LDS image (zero outside the image) ----
    if (wave < C::NPT_O) {
      wsrc.tile();
      const int ot = wave;
      const int oy = (ot / (C::TW / 8)) * 4 + (r >> 3), ox = (ot % (C::TW / 8)) * 8 + (r & 7);
      const int hbase = oy * C::HW + ox, pc = oy * C::TW + ox;
      const bool valid = (ty0 + oy < H) && (tx0 + ox < W);
      if constexpr (ROLE == 0) {
        f32x16 dtacc = dt_tile<T, C, WSrc<T, WLDS>, (sizeof(T) == 2 ? 4 : 1)>(DYs, wsrc, LW3T, hbase, lane);
        if (!valid) dtacc = zero16();
        if constexpr (DB2_REGS) {
#pragma unroll
          for (int i = 0; i < 16; ++i) db2acc[i] += dtacc[i];
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float v = half_sum(dtacc[i]);
            if (r == 0) atomicAdd(db2lds + (i & 3) + 8 * (i >> 2) + 4 * hh, v);
          }
        }
        scratch_store<T>(IMG + (pc - r) * 32, dtacc, true, r, hh);
      } else {
        FragT xb[C::KS1];
#pragma unroll
        for (int s = 0; s < C::KS1; ++s) xb[s] = lds_chunk<T>(XC, pc * C::KX + (2 * s + hh) * 8);
        const f32x16 tacc = t_tile<T, C, WSrc<T, WLDS>, (sizeof(T) == 2 ? 64 : 1)>(xb, wsrc, cls, lane);
        scratch_store<T>(IMG + (pc - r) * 32, tacc, valid, r, hh);
      }
    }
    SR_STAMP();
    __syncthreads();
    SR_STAMP();

    // ---- phase 2 ----
    if constexpr (ROLE == 0) {
      wsrc.tile();
#pragma unroll 1
      for (int ot = half; ot < C::NPT_O; ot += 2) {
        const int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
        const int pc = (toy + (r >> 3)) * C::TW + tox + (r & 7);
        auto rowx = [=](int p) { return ((toy + (p >> 3)) * C::TW + tox + (p & 7)) * C::KX; };
        auto rowi = [=](int p) { return ((toy + (p >> 3)) * C::TW + tox + (p & 7)) * 32; };
        f32x16 h2;
        if (C::FOLD_B1) {
          h2 = zero16();
        } else {
          const float b = cls[B::B1N_OFF + et * 32 + r];
#pragma unroll
          for (int i = 0; i < 16; ++i) h2[i] = b;
        }
#pragma unroll
        for (int s = 0; s < C::KS1; ++s)
          h2 = mma16<T>(lds_chunk<T>(XC, pc * C::KX + (2 * s + hh) * 8), wsrc.get(C::W1_OFF + et * C::KS1 + s, lane), h2);
        f32x16 dh2 = zero16();
#pragma unroll
        for (int s = 0; s < 2; ++s)
          dh2 = mma16<T>(lds_chunk<T>(IMG, pc * 32 + (2 * s + hh) * 8), wsrc.get(LW2N + 2 * et + s, lane), dh2);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          dh2[i] = h2[i] > 0.f ? dh2[i] : 0.f;
          sum += dh2[i];
        }
        db1 += sum;
        accA = mma16<T>(tr_frag<T>(XC, 0, lane, rowx), acc_to_frag<T, 0>(dh2), accA);
        accA = mma16<T>(tr_frag<T>(XC, 1, lane, rowx), acc_to_frag<T, 1>(dh2), accA);
        accB = mma16<T>(tr_frag<T>(IMG, 0, lane, rowi), acc_to_frag_relu<T, 0>(h2), accB);
        accB = mma16<T>(tr_frag<T>(IMG, 1, lane, rowi), acc_to_frag_relu<T, 1>(h2), accB);
      }
    } else {
      const int uy = wave / 3, ux = wave - uy * 3;    // tap u = wave
      constexpr int UNR2 = sizeof(T) == 2 ? 3 : 1;
#pragma unroll UNR2
      for (int ot = 0; ot < C::NPT_O; ++ot) {
        const int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
        auto rowi = [=](int p) { return ((toy + (p >> 3)) * C::TW + tox + (p & 7)) * 32; };
        auto rowd = [=](int p) { return ((toy + (p >> 3) + uy) * C::HW + tox + (p & 7) + ux) * C::F; };
        accA = mma16<T>(tr_frag<T>(IMG, 0, lane, rowi), tr_frag<T>(DYs, 0, lane, rowd), accA);
        accA = mma16<T>(tr_frag<T>(IMG, 1, lane, rowi), tr_frag<T>(DYs, 1, lane, rowd), accA);
      }
    }
    SR_STAMP();
    if (has_next) {
      if constexpr (DBUF) {
        cur ^= 1;
        regs.store(BUF + cur * TILE_ELEMS, BUF + cur * TILE_ELEMS + B::DY_ELEMS, tid);
      } else {
        __syncthreads();                               // everyone done with the single buffer
        fetch(t + gridDim.x);
        regs.store(BUF, BUF + B::DY_ELEMS, tid);
      }
    }
  }

  // ---- combine the two waves of each e-tile through plain LDS stores (no LDS atomics), one coalesced
  //      store per workgroup.  ROLE 1: every tile has a single owner wave -> straight to HBM. ----
  float* out = partial + ((size_t)layer * gridDim.x + blockIdx.x) * SLAB;
  if constexpr (ROLE == 0) {
    float db2keep = 0.f;
    if constexpr (!DB2_REGS) { if (tid < 32) db2keep = db2lds[tid]; }
    __syncthreads();
    SR_STAMP();
    float* red = reinterpret_cast<float*>(smem_raw);
    float* mine = red + half * SLAB;
    float* db2w = red + 2 * SLAB;
    slab_store_tile(mine, et, accA, lane);
    slab_store_tile(mine, C::NET + et, accB, lane);
    const float d1 = db1 + __shfl_xor(db1, 32);
    if (hh == 0) mine[2 * C::NET * 1024 + et * 32 + r] = d1;
    if constexpr (DB2_REGS) {
      if (wave < C::NPT_O) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = half_sum(db2acc[i]);
          if (r == 0) db2w[wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh] = v;
        }
      }
    }
    __syncthreads();
    SR_STAMP();
    constexpr int NSUM = 2 * C::NET * 1024 + C::NET * 32;
    for (int i = tid; i < NSUM; i += NTHREADS) out[i] = red[i] + red[SLAB + i];
    if (tid < 32) {
      float v = db2keep;
      if constexpr (DB2_REGS) {
#pragma unroll
        for (int w = 0; w < C::NPT_O; ++w) v += db2w[w * 32 + tid];
      }
      out[NSUM + tid] = v;
    }
    SR_STAMP();
  } else {
    SR_STAMP();
    SR_STAMP();
    slab_store_tile(out, wave, accA, lane);
    SR_STAMP();
  }
  SR_STAMP();
}

// =============================================================================================
// two residual blocks per launch (bf16, F = 24): at batch 32 a block launch is bound by its fixed costs
// (launch/drain ~2.7 us, staging ~1 us) rather than by bytes or flops, so block A is computed on the
// tile + 1-pixel halo (from x on a 2-pixel halo) and handed to block B through LDS.  The forward pair lives
// in wdsr_fwd_rs.h (register-resident weights); the geometry and the 3x3 / conv1-conv2 helpers below are shared
// with the backward pair kernel.
// =============================================================================================
template <int F_, int E_, int L_> struct Pair {
  typedef BlockCfg<F_, E_, L_> C;
  static constexpr int W2 = C::TW + 4, H2 = C::TH + 4, NP2 = W2 * H2;          // 28 x 16 = 448 = 14 * 32
  static constexpr int NWAVES = NP2 / 32;
  static constexpr int XA_ELEMS = NP2 * C::KX, T_ELEMS = NP2 * C::LP, XB_ELEMS = C::NPXH_PAD * C::KX;
  static constexpr int W_ELEMS = C::NFRAG_FWD * 512;
  static constexpr int LDS_ELEMS = XA_ELEMS + T_ELEMS + XB_ELEMS + 2 * W_ELEMS;
  static_assert(NP2 % 32 == 0, "16x28 region must be whole 32-pixel tiles");
};

// t^T = W2 relu(W1 x + b1) + b2 for the 32 pixels `xrow` points at (one pixel per lane pair)
template <typename T, typename C, typename WS>
SR_DEV f32x16 t_from_x(const T* Ximg, int xrow, const WS& wsrc, const float* __restrict__ cinit, int lane) {
  typedef typename FragOf<T>::type FragT;
  const int hh = lane >> 5;
  FragT xb[C::KS1];
#pragma unroll
  for (int s = 0; s < C::KS1; ++s) xb[s] = lds_chunk<T>(Ximg, xrow * C::KX + (2 * s + hh) * 8);
  return t_from_xb<T, C, WS, true>(xb, wsrc, cinit, lane);
}

// 3x3 + bias + residual for one pixel per lane pair: taps at Timg[(trow + ty*tstride + tx)], residual from Ximg[xrow]
template <typename T, typename C, typename WS>
SR_DEV f32x16 y_from_t(const T* Timg, int trow, int tstride, const T* Ximg, int xrow, const WS& wsrc, int lane) {
  const int hh = lane >> 5;
  f32x16 oacc = zero16();
#pragma unroll
  for (int s = 0; s < C::KS3; ++s) {
    const int q = 2 * s + hh;
    const T* src;
    if (q < 9 * C::CPT) {
      const int tap = q / C::CPT, c = q - tap * C::CPT;
      src = Timg + (trow + (tap / 3) * tstride + (tap % 3)) * C::LP + c * 8;
    } else {
      int c = q - 9 * C::CPT;
      if (c >= C::FC) c = 0;
      src = Ximg + xrow * C::KX + c * 8;
    }
    oacc = mma16<T>(wsrc.get(C::W3_OFF + s, lane), *reinterpret_cast<const typename FragOf<T>::type*>(src), oacc);
  }
  return oacc;
}

// =============================================================================================
// backward-data of two consecutive blocks per launch (bf16, F = 24; same reasoning as the forward pair):
// dyB (gradient at block B's output) -> dxB = dyA on the tile + 1-pixel halo (LDS, and HBM for the core:
// the weight-gradient kernels read it) -> dxA on the core.  Bit-identical to two single-block launches.
//   phase 1 (12 waves): dxB on the 14x26 region from dyB on the 16x28 region and xB (= block A's output)
//   phase 2 ( 9 waves): dxA on the 12x24 core from the LDS dxB image and xA
// =============================================================================================
template <typename T, int NTHREADS, int RW, int NROWS, int NLIVE, int CH, int FCH, bool ONES> struct RegionRegs {
  typedef typename FragOf<T>::type FragT;
  static constexpr int TOTAL = NROWS * CH, ITER = (TOTAL + NTHREADS - 1) / NTHREADS;
  FragT v[ITER];
  // rows p < NLIVE are pixels (y0 + p / RW, x0 + p % RW) of the NHWC image `src` (FCH chunks of 8 channels);
  // everything else, and pixels outside the image, is zero; chunk FCH carries the ones channel if ONES
  SR_DEV void load(const T* __restrict__ src, int H, int W, int y0, int x0, int tid) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * NTHREADS;
      const int p = idx / CH, c = idx - p * CH;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[it][j] = (T)0.f;
      if (idx < TOTAL) {
        if (c < FCH) {
          if (p < NLIVE) {
            const int py = p / RW, px = p - py * RW;
            const int Y = y0 + py, X = x0 + px;
            if (Y >= 0 && Y < H && X >= 0 && X < W) v[it] = *reinterpret_cast<const FragT*>(src + ((size_t)Y * W + X) * (FCH * 8) + c * 8);
          }
        } else if (ONES && c == FCH) {
          v[it][0] = (T)1.f;
        }
      }
    }
  }
  SR_DEV void store(T* dst, int tid) const {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * NTHREADS;
      if (idx < TOTAL) *reinterpret_cast<FragT*>(dst + idx * 8) = v[it];
    }
  }
};

// dx^T for 32 pixels: dy + W1^T [ 1(h>0) * W2^T dt ], h recomputed from the x row; `dyoff` = element offset
// of each lane's own pixel in the dy image (identity term)
template <typename T, typename C, typename WS>
SR_DEV f32x16 dx_from_dt(const f32x16& dtacc, const T* Ximg, int xrow, const T* DYimg, int dyoff, const WS& wsrc,
                         int lw2t, int lw1t, int lid, const float* __restrict__ cinit, int lane) {
  typedef BwdCfg<C> B;
  typedef typename FragOf<T>::type FragT;
  const int hh = lane >> 5;
  const FragT dtb0 = acc_to_frag<T, 0>(dtacc), dtb1 = acc_to_frag<T, 1>(dtacc);
  FragT xb[C::KS1];
#pragma unroll
  for (int s = 0; s < C::KS1; ++s) xb[s] = lds_chunk<T>(Ximg, xrow * C::KX + (2 * s + hh) * 8);
  f32x16 dxacc = zero16();
#pragma unroll
  for (int et = 0; et < C::NET; ++et) {
    f32x16 hacc = C::FOLD_B1 ? zero16() : load_cinit(cinit + 32 + et * 32, hh);
#pragma unroll
    for (int s = 0; s < C::KS1; ++s) hacc = mma16<T>(wsrc.get(C::W1_OFF + et * C::KS1 + s, lane), xb[s], hacc);
    f32x16 dh = zero16();
    dh = mma16<T>(wsrc.get(lw2t + 2 * et, lane), dtb0, dh);
    dh = mma16<T>(wsrc.get(lw2t + 2 * et + 1, lane), dtb1, dh);
#pragma unroll
    for (int i = 0; i < 16; ++i) dh[i] = hacc[i] > 0.f ? dh[i] : 0.f;
    if (2 * et < C::KS2) dxacc = mma16<T>(wsrc.get(lw1t + 2 * et, lane), acc_to_frag<T, 0>(dh), dxacc);
    if (2 * et + 1 < C::KS2) dxacc = mma16<T>(wsrc.get(lw1t + 2 * et + 1, lane), acc_to_frag<T, 1>(dh), dxacc);
  }
#pragma unroll
  for (int s = 0; s < B::KSI; ++s) {
    int c = 2 * s + hh;
    if (c >= C::FC) c = 0;
    dxacc = mma16<T>(wsrc.get(lid + s, lane), lds_chunk<T>(DYimg, dyoff + c * 8), dxacc);
  }
  return dxacc;
}

// dt^T for 32 pixels, taps at DY[(hbase + ty * stride + tx)]
template <typename T, typename C, typename WS>
SR_DEV f32x16 dt_from_dy(const T* DYs, int hbase, int stride, const WS& wsrc, int w3t_base, int lane) {
  typedef BwdCfg<C> B;
  const int hh = lane >> 5;
  f32x16 acc = zero16();
#pragma unroll
  for (int s = 0; s < B::KS3B; ++s) {
    const int q = 2 * s + hh;
    int off = hbase * C::F;
    if (q < 9 * C::FC) {
      const int u = q / C::FC, c = q - u * C::FC;
      off = (hbase + (u / 3) * stride + (u % 3)) * C::F + c * 8;
    }
    acc = mma16<T>(wsrc.get(w3t_base + s, lane), lds_chunk<T>(DYs, off), acc);
  }
  return acc;
}

template <typename T, int F, int E, int L>
__global__ __launch_bounds__((64 * BlockCfg<F, E, L>::NPT_H)) void wdsr_block2_bwd_data_kernel(
    const T* __restrict__ xa, const T* __restrict__ xb, const T* __restrict__ dyb, T* __restrict__ dxb,
    T* __restrict__ dxa, const T* __restrict__ wa, const T* __restrict__ wb, const float* __restrict__ cia,
    const float* __restrict__ cib, T* __restrict__ dta, T* __restrict__ dtb, int H, int W, int tiles_x) {
  typedef BlockCfg<F, E, L> C;
  typedef BwdCfg<C> B;
  typedef Pair<F, E, L> P;
  typedef typename FragOf<T>::half_type HalfT;
  constexpr int NTHREADS = 64 * C::NPT_H;
  constexpr int NW1 = C::NET * C::KS1, NREST = B::W2N_OFF - B::W3T_OFF, NWL = NW1 + NREST;
  constexpr int LW3T = NW1, LW2T = LW3T + B::KS3B, LW1T = LW2T + 2 * C::NET, LID = LW1T + C::KS2;
  constexpr int DY2_ELEMS = (P::NP2 + 2) * C::F, XB_ELEMS = C::NPXH_PAD * C::KX;
  static_assert(B::XC_ELEMS <= DY2_ELEMS, "xA reuses the dyB image");
  __shared__ __attribute__((aligned(16))) T smem[DY2_ELEMS + XB_ELEMS + B::DY_ELEMS + 2 * NWL * 512];
  T* const DY2 = smem;
  T* const XBs = DY2 + DY2_ELEMS;
  T* const DY1 = XBs + XB_ELEMS;
  T* const XA = DY2;                      // phase 2 only: xA waits in registers during phase 1 (LDS budget)
  T* const WL = DY1 + B::DY_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * F;
  const size_t tile_g = (size_t)n * gridDim.x + tile;            // tile index of the saved dt images

  SR_STAMP_DECL;
  SR_STAMP();
  WSrc<T, true> wsa, wsb;
  wsa.p = WL;
  wsb.p = WL + NWL * 512;
  stage_weights<T, NTHREADS>(WL, wa, NW1, tid);
  stage_weights<T, NTHREADS>(WL + NW1 * 512, wa + (size_t)B::W3T_OFF * 512, NREST, tid);
  stage_weights<T, NTHREADS>(WL + NWL * 512, wb, NW1, tid);
  stage_weights<T, NTHREADS>(WL + (NWL + NW1) * 512, wb + (size_t)B::W3T_OFF * 512, NREST, tid);
  RegionRegs<T, NTHREADS, C::TW, B::NPXC + 1, B::NPXC, C::KX / 8, C::FC, C::FOLD_B1> ra;
  {
    RegionRegs<T, NTHREADS, P::W2, P::NP2 + 2, P::NP2, C::FC, C::FC, false> rd;
    RegionRegs<T, NTHREADS, C::HW, C::NPXH_PAD, C::NPXH, C::KX / 8, C::FC, C::FOLD_B1> rb;
    rd.load(dyb + img, H, W, ty0 - 2, tx0 - 2, tid);
    rb.load(xb + img, H, W, ty0 - 1, tx0 - 1, tid);
    ra.load(xa + img, H, W, ty0, tx0, tid);
    rd.store(DY2, tid);
    rb.store(XBs, tid);
    // slack rows of the dxB image (read by the padded taps of phase 2, never written by phase 1)
    for (int i = tid; i < (C::NPXH_PAD + 2 - C::NPXH) * C::F; i += NTHREADS) DY1[C::NPXH * C::F + i] = (T)0.f;
  }
  SR_STAMP();
  __syncthreads();
  SR_STAMP();

  // ---- phase 1: dxB on the 14x26 region ----
  {
    const int hp1 = wave * 32 + r;
    const bool live = hp1 < C::NPXH;
    const int hp1c = live ? hp1 : 0;
    const int hy = hp1c / C::HW, hx = hp1c - hy * C::HW;
    SR_BWD_PRIO(2);
    const f32x16 dtacc = dt_from_dy<T, C>(DY2, hy * P::W2 + hx, P::W2, wsb, LW3T, lane);
    const f32x16 dxacc = dx_from_dt<T, C>(dtacc, XBs, hp1c, DY2, ((hy + 1) * P::W2 + hx + 1) * C::F, wsb, LW2T, LW1T,
                                           LID, cib, lane);
    SR_BWD_PRIO(0);
    if (live) {
      const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
      const bool inimg = (Y >= 0 && Y < H && X >= 0 && X < W);
      if (dtb && hy >= 1 && hy <= C::TH && hx >= 1 && hx <= C::TW) {
        T* o = dtb + (tile_g * B::NPXC + (hy - 1) * C::TW + hx - 1) * C::LP;
#pragma unroll
        for (int g = 0; g < C::CPT; ++g) {
          HalfT v = acc_group<T>(dtacc, g);
          if (!inimg) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (T)0.f;
          }
          stream_store(reinterpret_cast<HalfT*>(o + g * 8 + hh * 4), v);
        }
      }
#pragma unroll
      for (int g = 0; g < C::FC; ++g) {
        HalfT v = acc_group<T>(dxacc, g);
        if (!inimg) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (T)0.f;
        }
        *reinterpret_cast<HalfT*>(DY1 + hp1 * C::F + g * 8 + hh * 4) = v;
      }
      if (inimg && hy >= 1 && hy <= C::TH && hx >= 1 && hx <= C::TW) {
        T* o = dxb + img + ((size_t)Y * W + X) * F;
#pragma unroll
        for (int g = 0; g < C::FC; ++g) stream_store(reinterpret_cast<HalfT*>(o + g * 8 + hh * 4), acc_group<T>(dxacc, g));
      }
    }
  }
  SR_STAMP();
  __syncthreads();
  ra.store(XA, tid);
  __syncthreads();
  SR_STAMP();

  // ---- phase 2: dxA on the core ----
  if (wave < C::NPT_O) {
    const int ot = wave;
    const int oy = (ot / (C::TW / 8)) * 4 + (r >> 3), ox = (ot % (C::TW / 8)) * 8 + (r & 7);
    const int hbase = oy * C::HW + ox, pc = oy * C::TW + ox;
    SR_BWD_PRIO(2);
    const f32x16 dtacc = dt_from_dy<T, C>(DY1, hbase, C::HW, wsa, LW3T, lane);
    const f32x16 dxacc = dx_from_dt<T, C>(dtacc, XA, pc, DY1, (hbase + C::HW + 1) * C::F, wsa, LW2T, LW1T, LID, cia, lane);
    SR_BWD_PRIO(0);
    const int Y = ty0 + oy, X = tx0 + ox;
    if (dta) {
      T* o = dta + (tile_g * B::NPXC + pc) * C::LP;
#pragma unroll
      for (int g = 0; g < C::CPT; ++g) {
        HalfT v = acc_group<T>(dtacc, g);
        if (!(Y < H && X < W)) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (T)0.f;
        }
        stream_store(reinterpret_cast<HalfT*>(o + g * 8 + hh * 4), v);
      }
    }
    if (Y < H && X < W) {
      T* o = dxa + img + ((size_t)Y * W + X) * F;
#pragma unroll
      for (int g = 0; g < C::FC; ++g) stream_store(reinterpret_cast<HalfT*>(o + g * 8 + hh * 4), acc_group<T>(dxacc, g));
    }
  }
  SR_STAMP();
}

// =============================================================================================
// weight gradients from SAVED intermediates (bf16, F = 24, networks run through the two-block kernels):
// the forward pair kernel keeps t (the 3x3 conv's input) and the backward pair kernel keeps dt (the gradient
// at the 3x3 conv's input) of every core pixel, tile-local [tile][288][LP].  HBM has ~85 % headroom in this
// path while the recompute phases and their barriers were half of the weight-gradient time, so
//   ROLE 0: x core tile + dt image -> dW1, dW2, db1, db2     (no dy halo, no transposed 3x3, one barrier/tile)
//   ROLE 1: t image + dy halo tile -> dW3 (+ b3 via t's ones channel)   (no weights at all)
// Slab layouts are those of wdsr_block_wgrad_kernel.
// =============================================================================================
template <int F, int E, int L, int ROLE> struct WgradSavedCfg {
  // ROLE 0: waves per e-tile (each takes every NSPLIT-th pixel tile); 32 units: 6 e-tiles, two partial slabs fit LDS
  static constexpr int NSPLIT = BlockCfg<F, E, L>::NET <= 5 ? 3 : 2;
  static constexpr int NWAVES = ROLE == 0 ? NSPLIT * BlockCfg<F, E, L>::NET : 9;
};

template <typename T, int F, int E, int L, int ROLE>
__global__ __launch_bounds__((64 * WgradSavedCfg<F, E, L, ROLE>::NWAVES)) void wdsr_block_wgrad_saved_kernel(
    const T* __restrict__ act, const T* __restrict__ side, const T* __restrict__ wblob, const float* __restrict__ cinit,
    float* __restrict__ partial, int N, int H, int W, int tiles_x, int tiles_per_img, long act_ls, long side_ls, long w_ls,
    long c_ls) {
  typedef BlockCfg<F, E, L> C;
  typedef BwdCfg<C> B;
  typedef WgradSavedCfg<F, E, L, ROLE> G;
  typedef typename FragOf<T>::type FragT;
  static_assert(sizeof(T) == 2, "saved-image weight gradients: bf16");
  constexpr int NTHREADS = 64 * G::NWAVES;
  constexpr int IMG_ELEMS = (B::NPXC + 1) * 32;                       // dt or t image: [core px][32 ch]
  constexpr int ACT_ELEMS = ROLE == 0 ? B::XC_ELEMS : B::DY_ELEMS;     // x core tile / dy halo tile
  constexpr int TILE_ELEMS = IMG_ELEMS + ACT_ELEMS;
  constexpr int NW1 = C::NET * C::KS1, LW2N = NW1;
  constexpr int NWL = ROLE == 0 ? NW1 + 2 * C::NET : 0;
  constexpr int SLAB = ROLE == 0 ? B::SLAB_A : B::SLAB_B;
  constexpr int NSIDE = B::NPXC * C::CPT, IS = (NSIDE + NTHREADS - 1) / NTHREADS;     // 16-byte chunks of a saved image
  constexpr int STAGE_BYTES = (2 * TILE_ELEMS + NWL * 512) * (int)sizeof(T);
  // ROLE 0: per-thread running sums of the staged dt chunks (-> db2) live in LDS behind both the staging
  // buffers and the epilogue's partial slabs (8 VGPRs the 128-register budget does not have)
  constexpr int P_OFF = STAGE_BYTES > G::NSPLIT * SLAB * 4 ? STAGE_BYTES : G::NSPLIT * SLAB * 4;
  constexpr int LDS_BYTES = ROLE == 0 ? P_OFF + (IS * NTHREADS * 8 + 8 * 32) * 4 : STAGE_BYTES;
  __shared__ __attribute__((aligned(16))) char smem_raw[LDS_BYTES];
  T* const BUF = reinterpret_cast<T*>(smem_raw);                       // [2][IMG | ACT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;

  const int layer = blockIdx.y;
  act += (size_t)layer * act_ls;
  side += (size_t)layer * side_ls;
  WSrc<T, true> wsrc;
  float b1n[ROLE == 0 && !C::FOLD_B1 ? 1 : 1] = {0.f};      // conv1 bias of this lane's e column (32 units: not folded)
  if constexpr (ROLE == 0 && !C::FOLD_B1) b1n[0] = cinit[(size_t)layer * c_ls + B::B1N_OFF + (wave / G::NSPLIT) * 32 + r];
  if constexpr (ROLE == 0) {
    wblob += (size_t)layer * w_ls;
    T* wl = BUF + 2 * TILE_ELEMS;
    stage_weights<T, NTHREADS>(wl, wblob, NW1, tid);
    stage_weights<T, NTHREADS>(wl + NW1 * 512, wblob + (size_t)B::W2N_OFF * 512, 2 * C::NET, tid);
    wsrc.p = wl;
  }
  // channels LP..31 and the slack row of both images are never written by the staging: zero them once
  for (int i = tid; i < 2 * (B::NPXC + 1); i += NTHREADS) {
    T* row = BUF + (i / (B::NPXC + 1)) * TILE_ELEMS + (i % (B::NPXC + 1)) * 32;
    const int c0 = (i % (B::NPXC + 1)) == B::NPXC ? 0 : C::LP;
    for (int c = c0; c < 32; ++c) row[c] = (T)0.f;
  }

  f32x16 accA = zero16(), accB = zero16();          // ROLE 0: dW1^T[et], dW2[et].  ROLE 1: accA = dW3^T[tap = wave]
  float* const P = reinterpret_cast<float*>(smem_raw + (ROLE == 0 ? P_OFF : 0));   // [IS * NTHREADS][8]
  if constexpr (ROLE == 0) {
#pragma unroll
    for (int it = 0; it < IS; ++it)
#pragma unroll
      for (int j = 0; j < 8; ++j) P[(size_t)(tid + it * NTHREADS) * 8 + j] = 0.f;
  }
  float db1 = 0.f;
  const int et = wave / G::NSPLIT, half = wave % G::NSPLIT;

  const int total = N * tiles_per_img;
  FragT vs[IS];
  RegionRegs<T, NTHREADS, (ROLE == 0 ? C::TW : C::HW), (ROLE == 0 ? B::NPXC + 1 : C::NPXH_PAD + 2),
             (ROLE == 0 ? B::NPXC : C::NPXH), (ROLE == 0 ? C::KX / 8 : C::FC), C::FC, ROLE == 0> ra;
  auto fetch = [&](int t) {
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    int tid2 = tid;
    asm volatile("" : "+v"(tid2));                     // keep the per-thread indices out of long-lived registers
    const T* sp = side + (size_t)t * (B::NPXC * C::LP);
#pragma unroll
    for (int it = 0; it < IS; ++it) {
      const int idx = tid2 + it * NTHREADS;
#pragma unroll
      for (int j = 0; j < 8; ++j) vs[it][j] = (T)0.f;
      if (idx < NSIDE) vs[it] = *reinterpret_cast<const FragT*>(sp + (size_t)idx * 8);
    }
    const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
    ra.load(act + (size_t)n * H * W * F, H, W, ROLE == 0 ? ty0 : ty0 - 1, ROLE == 0 ? tx0 : tx0 - 1, tid2);
  };
  auto store = [&](int buf) {
    T* IMGb = BUF + buf * TILE_ELEMS;
#pragma unroll
    for (int it = 0; it < IS; ++it) {
      const int idx = tid + it * NTHREADS;
      if (idx < NSIDE) {
        *reinterpret_cast<FragT*>(IMGb + (idx / C::CPT) * 32 + (idx % C::CPT) * 8) = vs[it];
        if constexpr (ROLE == 0) {
          f32x4* pp = reinterpret_cast<f32x4*>(P + (size_t)idx * 8);
          f32x4 a = pp[0], b = pp[1];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            a[j] += (float)vs[it][j];
            b[j] += (float)vs[it][4 + j];
          }
          pp[0] = a;
          pp[1] = b;
        }
      }
    }
    ra.store(IMGb + IMG_ELEMS, tid);
  };
  int cur = 0;
  if ((int)blockIdx.x < total) {
    fetch(blockIdx.x);
    store(0);
  }
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const bool has_next = t + (int)gridDim.x < total;
    const T* IMG = BUF + cur * TILE_ELEMS;
    const T* ACT = IMG + IMG_ELEMS;
    __syncthreads();                                   // tile t staged; the other buffer is free
    if (has_next) fetch(t + gridDim.x);
    if constexpr (ROLE == 0) {
      const T* XC = ACT;
#pragma unroll 1
      for (int ot = half; ot < C::NPT_O; ot += G::NSPLIT) {
        const int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
        const int pc = (toy + (r >> 3)) * C::TW + tox + (r & 7);
        auto rowx = [=](int p) { return ((toy + (p >> 3)) * C::TW + tox + (p & 7)) * C::KX; };
        auto rowi = [=](int p) { return ((toy + (p >> 3)) * C::TW + tox + (p & 7)) * 32; };
        f32x16 h2;
#pragma unroll
        for (int i = 0; i < 16; ++i) h2[i] = b1n[0];
#pragma unroll
        for (int s = 0; s < C::KS1; ++s)
          h2 = mma16<T>(lds_chunk<T>(XC, pc * C::KX + (2 * s + hh) * 8), wsrc.get(C::W1_OFF + et * C::KS1 + s, lane), h2);
        f32x16 dh2 = zero16();
#pragma unroll
        for (int s = 0; s < 2; ++s)
          dh2 = mma16<T>(lds_chunk<T>(IMG, pc * 32 + (2 * s + hh) * 8), wsrc.get(LW2N + 2 * et + s, lane), dh2);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          dh2[i] = h2[i] > 0.f ? dh2[i] : 0.f;
          sum += dh2[i];
        }
        db1 += sum;
        accA = mma16<T>(tr_frag<T>(XC, 0, lane, rowx), acc_to_frag<T, 0>(dh2), accA);
        accA = mma16<T>(tr_frag<T>(XC, 1, lane, rowx), acc_to_frag<T, 1>(dh2), accA);
        accB = mma16<T>(tr_frag<T>(IMG, 0, lane, rowi), acc_to_frag_relu<T, 0>(h2), accB);
        accB = mma16<T>(tr_frag<T>(IMG, 1, lane, rowi), acc_to_frag_relu<T, 1>(h2), accB);
      }
    } else {
      const T* DYs = ACT;
      const int uy = wave / 3, ux = wave - uy * 3;    // tap u = wave
#pragma unroll 3
      for (int ot = 0; ot < C::NPT_O; ++ot) {
        const int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
        auto rowi = [=](int p) { return ((toy + (p >> 3)) * C::TW + tox + (p & 7)) * 32; };
        auto rowd = [=](int p) { return ((toy + (p >> 3) + uy) * C::HW + tox + (p & 7) + ux) * C::F; };
        accA = mma16<T>(tr_frag<T>(IMG, 0, lane, rowi), tr_frag<T>(DYs, 0, lane, rowd), accA);
        accA = mma16<T>(tr_frag<T>(IMG, 1, lane, rowi), tr_frag<T>(DYs, 1, lane, rowd), accA);
      }
    }
    if (has_next) {
      cur ^= 1;
      store(cur);
    }
  }

  float* out = partial + ((size_t)layer * gridDim.x + blockIdx.x) * SLAB;
  if constexpr (ROLE == 0) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);
    float* mine = red + half * SLAB;
    float* Q = P + IS * NTHREADS * 8;                 // [8 parts][32]
    slab_store_tile(mine, et, accA, lane);
    slab_store_tile(mine, C::NET + et, accB, lane);
    const float d1 = db1 + __shfl_xor(db1, 32);
    if (hh == 0) mine[2 * C::NET * 1024 + et * 32 + r] = d1;
    __syncthreads();
    constexpr int NSUM = 2 * C::NET * 1024 + C::NET * 32;
    for (int i = tid; i < NSUM; i += NTHREADS) {
      float v = red[i];
#pragma unroll
      for (int q = 1; q < G::NSPLIT; ++q) v += red[q * SLAB + i];
      out[i] = v;
    }
    if (tid < 8 * C::LP) {                            // db2[ch] = sum over pixels of chunk (px * CPT + ch / 8), element ch % 8
      const int part = tid / C::LP, ch = tid - part * C::LP;
      constexpr int PER = (B::NPXC + 7) / 8;
      float v = 0.f;
      for (int px = part * PER; px < (part + 1) * PER && px < B::NPXC; ++px) v += P[(size_t)(px * C::CPT + (ch >> 3)) * 8 + (ch & 7)];
      Q[part * 32 + ch] = v;
    }
    __syncthreads();
    if (tid < 32) {
      float v = 0.f;
      if (tid < C::LP) {
#pragma unroll
        for (int part = 0; part < 8; ++part) v += Q[part * 32 + tid];
      }
      out[NSUM + tid] = v;
    }
  } else {
    slab_store_tile(out, wave, accA, lane);
  }
}

