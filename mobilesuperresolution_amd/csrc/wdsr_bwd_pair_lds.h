// Backward-data of TWO fused WDSR-B residual blocks with every weight fragment read from LDS at use (bf16; the 32-unit network's
// route, round 3).  Reference op: the autograd backward of Block.forward (models/basic_wdsr_b.py:142-144) for blocks b (the later
// one) and a:   dt = conv3x3^T(dy)     h = relu(W1 x + b1) recomputed     dx = dy + W1^T [ 1(h > 0) . W2^T dt ],   dx_b = dy_a.
//
// wdsr_bwd_rs.h keeps a phase's weights in registers and therefore splits a block into a dt phase and a dx phase with dt handed
// over through LDS; its layouts are written around 24 units (three chunks per pixel, the ones channel in the padding), and the
// 32-unit weight sets (18 + 36 fragments) would not fit.  Here, as in wdsr_block_bwd_data_kernel, a wave runs a pixel tile's WHOLE
// chain -- dt (in registers) -> per e-tile: h, dh, mask, dx -- with the fragments coming from LDS, so a block is ONE pass without
// a barrier inside; two passes, twelve waves (pass b: the 14 x 26 region = 12 pixel tiles, pass a: the 12 x 24 core = 9):
//     stage dy_b (tile + 2), x_b (tile + 1), block b's fragments | pass b: dx_b -> LDS (= dy_a, zero outside the image) and -> HBM
//     (core) | stage block a's fragments over block b's and x_a over dy_b | pass a: dx_a -> HBM.
// Per pixel exactly the arithmetic of two wdsr_block_bwd_data launches (same products, same order, dx_b rounded to bf16 in between
// as the global tensor is): BIT-IDENTICAL to them, including the saved dt images.  What the pair saves is a launch, the HBM round
// trip of dx_b and the second dispatch ramp; the second block's fragments (56 KB) are staged in the open between the passes.
#pragma once
#include <type_traits>
#include "wdsr_block.h"
#include "wdsr_fwd_rs.h"

template <int F_, int E_, int L_> struct BwdPairCfg {
  typedef BlockCfg<F_, E_, L_> C;
  typedef BwdCfg<C> B;
  static_assert(!C::FOLD_B1 && C::KX == C::F, "x rows are the F real channels (b1 from the C-init table): 32 units");
  static constexpr int RW0 = C::TW + 4, NP0 = RW0 * (C::TH + 4);        // dy_b: tile + 2
  static constexpr int RW1 = C::TW + 2, NP1 = RW1 * (C::TH + 2);        // x_b, dx_b (= dy_a): tile + 1
  static constexpr int NPC = C::TH * C::TW;
  static constexpr int NT1 = (NP1 + 31) / 32, NTC = (NPC + 31) / 32, NWAVES = NT1, NTHREADS = 64 * NWAVES;
  // LDS rows of F elements (64 bytes at 32 units).  Rows padded to 80 bytes (conflict-free for the 16-byte window reads of 32
  // consecutive pixels) were measured: 19.80 against 19.83 us -- the passes wait on their dependent chains, not on LDS bandwidth
  static constexpr int RS = C::F, CH = RS / 8;                          // row stride in elements, 16-byte chunks per LDS row
  static constexpr int rows_elems(int nrows) { return (nrows * CH + 63) / 64 * 64 * 8; }   // whole 1 KB DMA pieces
  static constexpr int DY_ELEMS = rows_elems(NP0 + 2), XB_ELEMS = rows_elems(NP1 + 1), DX_ELEMS = rows_elems(NP1 + 2);
  static_assert(rows_elems(NPC + 1) <= DY_ELEMS, "x_a fits the dy_b buffer");
  // staged fragments of one block: W1 as it lies at the head of the blob, then W3T | W2T | W1T | ID from behind the forward section
  static constexpr int NW1 = C::NET * C::KS1, NREST = B::W2N_OFF - B::W3T_OFF, NFR = NW1 + NREST;
  static constexpr int LW3T = NW1, LW2T = LW3T + B::KS3B, LW1T = LW2T + 2 * C::NET, LID = LW1T + C::KS2;
  static constexpr int LDS_BYTES = (DY_ELEMS + XB_ELEMS + DX_ELEMS + NFR * 512) * 2;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// a region of an NHWC image (F channels per pixel) -> LDS rows of RS elements by LDS-DMA, zero outside the image, past NLIVE and in
// the padding chunk
template <int F, int RS, int NWAVES>
SR_DEV void bwp_dma_region(__bf16* dst, const __bf16* __restrict__ img, int H, int W, int y0, int x0, int rw, int nlive, int nrows, int tid) {
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr int CH = RS / 8;
  const int np = (nrows * CH + 63) / 64;
  const int lane = tid & 63, wave = tid >> 6;
  const char* const zeros = reinterpret_cast<const char*>(g_sr_const_chunks) + 16;
  char* const d = reinterpret_cast<char*>(dst);
  for (int p = wave; p < np; p += NWAVES) {
    const int idx = p * 64 + lane;
    const int pix = idx / CH, c = idx - pix * CH;
    const char* src = zeros;
    if (pix < nlive && c < F / 8) {
      const int py = pix / rw, px = pix - py * rw;
      const int Y = y0 + py, X = x0 + px;
      if (Y >= 0 && Y < H && X >= 0 && X < W) src = reinterpret_cast<const char*>(img + ((size_t)Y * W + X) * F + c * 8);
    }
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(d + p * 1024), 16, 0, 0);
  }
}

// grid = (tiles, N), 768 threads.  dta / dtb: saved dt images [N][tiles][288][LP] (nullptr: not kept).
template <int F, int E, int L>
__global__ __launch_bounds__((BwdPairCfg<F, E, L>::NTHREADS)) void wdsr_bwd_pair_lds_kernel(
    const __bf16* __restrict__ xa, const __bf16* __restrict__ xb, const __bf16* __restrict__ dyb, __bf16* __restrict__ dxb,
    __bf16* __restrict__ dxa, const __bf16* __restrict__ wa, const __bf16* __restrict__ wb, const float* __restrict__ cia,
    const float* __restrict__ cib, __bf16* __restrict__ dta, __bf16* __restrict__ dtb, int H, int W, int tiles_x) {
  typedef __bf16 T;
  typedef BwdPairCfg<F, E, L> R;
  typedef typename R::C C;
  typedef typename R::B B;
  typedef typename FragOf<T>::type FragT;
  typedef typename FragOf<T>::half_type HalfT;
  constexpr int NTHREADS = R::NTHREADS;
  __shared__ __attribute__((aligned(16))) T smem[R::DY_ELEMS + R::XB_ELEMS + R::DX_ELEMS + R::NFR * 512];
  T* const DY = smem;                       // dy_b on the tile + 2; between the passes: x_a on the core
  T* const XB = DY + R::DY_ELEMS;           // x_b on the tile + 1
  T* const DX = XB + R::XB_ELEMS;           // dx_b = dy_a on the tile + 1 (zero outside the image)
  T* const WL = DX + R::DX_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * F;
  const size_t tile_g = (size_t)n * gridDim.x + tile;
  WSrc<T, true> wsrc;
  wsrc.p = WL;
  auto stage_block_weights = [&](const T* wblob) {
    stage_weights<T, NTHREADS>(WL, wblob, R::NW1, tid);
    stage_weights<T, NTHREADS>(WL + R::NW1 * 512, wblob + (size_t)B::W3T_OFF * 512, R::NREST, tid);
  };
  stage_block_weights(wb);
  bwp_dma_region<F, R::RS, R::NWAVES>(DY, dyb + img, H, W, ty0 - 2, tx0 - 2, R::RW0, R::NP0, R::NP0 + 2, tid);
  bwp_dma_region<F, R::RS, R::NWAVES>(XB, xb + img, H, W, ty0 - 1, tx0 - 1, R::RW1, R::NP1, R::NP1 + 1, tid);
  __syncthreads();

  // one block on an output region (RWO wide, NPO pixels, its pixel (hy, hx) = pixel (hy + 1, hx + 1) of the dy image, RWO + 2 wide;
  // HALO = offset of the region inside the tile frame): the chain of wdsr_block_bwd_data_kernel for this wave's pixel tile
  auto block_pass = [&](auto rwo_c, auto npo_c, auto halo_c, const T* DYimg, const T* Ximg, const float* cinit, T* dx_lds, T* dx_glob,
                        T* dtsave) {
    constexpr int RWO = decltype(rwo_c)::value, NPO = decltype(npo_c)::value, HALO = decltype(halo_c)::value;
    constexpr int RWI = RWO + 2, NT = (NPO + 31) / 32;
    if (wave >= NT) return;                             // (wave-uniform; no barrier inside a pass)
    wsrc.tile();
    const int hp = wave * 32 + r;
    const bool live = hp < NPO;
    const int hpc = live ? hp : 0;
    const int hy = hpc / RWO, hx = hpc - hy * RWO;
    const int hbase = hy * RWI + hx;
    const int Y = ty0 - HALO + hy, X = tx0 - HALO + hx;
    const bool inimg = Y >= 0 && Y < H && X >= 0 && X < W;
    const bool core = live && hy >= HALO && hy < HALO + C::TH && hx >= HALO && hx < HALO + C::TW;
    // dt^T[l, px] = sum_{u, f} W3[f, l, 8 - u] dy[px + u - 1, f]
    f32x16 dtacc = zero16();
#pragma unroll
    for (int s = 0; s < B::KS3B; ++s) {
      const int q = 2 * s + hh;
      int off = hbase * R::RS;
      if (q < 9 * C::FC) {
        const int u = q / C::FC, c = q - u * C::FC;
        off = (hbase + (u / 3) * RWI + (u % 3)) * R::RS + c * 8;
      }
      dtacc = mma16<T>(wsrc.get(R::LW3T + s, lane), lds_chunk<T>(DYimg, off), dtacc);
    }
    if (dtsave && core) {                               // dt of the core pixels (zero outside the image)
      const int pc = (hy - HALO) * C::TW + hx - HALO;
      T* o = dtsave + (tile_g * R::NPC + pc) * C::LP;
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
    FragT xf[C::KS1];
#pragma unroll
    for (int s = 0; s < C::KS1; ++s) xf[s] = lds_chunk<T>(Ximg, hpc * R::RS + (2 * s + hh) * 8);
    f32x16 dxacc = zero16();
#pragma unroll
    for (int et = 0; et < C::NET; ++et) {
      f32x16 hacc = load_cinit(cinit + 32 + et * 32, hh);
#pragma unroll
      for (int s = 0; s < C::KS1; ++s) hacc = mma16<T>(wsrc.get(C::W1_OFF + et * C::KS1 + s, lane), xf[s], hacc);
      f32x16 dh = zero16();
      dh = mma16<T>(wsrc.get(R::LW2T + 2 * et, lane), dtb0, dh);
      dh = mma16<T>(wsrc.get(R::LW2T + 2 * et + 1, lane), dtb1, dh);
#pragma unroll
      for (int i = 0; i < 16; ++i) dh[i] = hacc[i] > 0.f ? dh[i] : 0.f;
      if (2 * et < C::KS2) dxacc = mma16<T>(wsrc.get(R::LW1T + 2 * et, lane), acc_to_frag<T, 0>(dh), dxacc);
      if (2 * et + 1 < C::KS2) dxacc = mma16<T>(wsrc.get(R::LW1T + 2 * et + 1, lane), acc_to_frag<T, 1>(dh), dxacc);
    }
#pragma unroll
    for (int s = 0; s < B::KSI; ++s) {                  // + dy (the skip connection), as an identity product
      int c = 2 * s + hh;
      if (c >= C::FC) c = 0;
      dxacc = mma16<T>(wsrc.get(R::LID + s, lane), lds_chunk<T>(DYimg, (hbase + RWI + 1) * R::RS + c * 8), dxacc);
    }
#pragma unroll
    for (int g = 0; g < C::FC; ++g) {
      HalfT v = acc_group<T>(dxacc, g);
      if (dx_lds && live) {
        HalfT z = v;
        if (!inimg) {
#pragma unroll
          for (int j = 0; j < 4; ++j) z[j] = (T)0.f;
        }
        *reinterpret_cast<HalfT*>(dx_lds + hp * R::RS + g * 8 + hh * 4) = z;
      }
      if (core && inimg) stream_store(reinterpret_cast<HalfT*>(dx_glob + ((size_t)Y * W + X) * F + g * 8 + hh * 4), v);
    }
  };
  T* const dtb_tile = dtb, * const dta_tile = dta;
  block_pass(std::integral_constant<int, R::RW1>{}, std::integral_constant<int, R::NP1>{}, std::integral_constant<int, 1>{}, DY, XB, cib, DX,
             dxb + img, dtb_tile);
  __syncthreads();                                      // dx_b (= dy_a) complete in LDS; block b's fragments and dy_b are free
  stage_block_weights(wa);
  bwp_dma_region<F, R::RS, R::NWAVES>(DY, xa + img, H, W, ty0, tx0, C::TW, R::NPC, R::NPC + 1, tid);
  __syncthreads();
  block_pass(std::integral_constant<int, C::TW>{}, std::integral_constant<int, R::NPC>{}, std::integral_constant<int, 0>{}, DX, DY, cia, (T*)nullptr,
             dxa + img, dta_tile);
}
