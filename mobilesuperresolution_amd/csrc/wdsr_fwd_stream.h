// STREAMING forward of two fused WDSR-B residual blocks over whole 48-pixel-wide images (bf16, gfx950): one workgroup walks an
// image top to bottom in bands of four rows and keeps only rings of rows in LDS.  Reference op: Block.forward,
// models/basic_wdsr_b.py:108-144, applied twice.
//
// Why a third forward design (numbers: DESIGN.md section 5).  The tile kernels of wdsr_fwd_rs.h recompute a halo: a 12 x 24
// tile of two blocks runs conv1 / conv2 on 28 x 16 and 26 x 14 pixels and the 3x3 conv on 26 x 14 and 24 x 12 -- 1.56 / 1.26 /
// 1.26 / 1.0 times the 288 pixels it owns -- and stops at four workgroup barriers per tile.  When a launch has at least one
// whole image per CU (training or inference at batch >= 256 of 48 x 48 patches) nothing needs recomputing: an image row is 48
// pixels = 1.5 MFMA pixel tiles, the zero padding left and right is two extra columns of the t rings, above and below a zeroed
// ring row, and a block's 3x3 conv simply runs two bands behind its 1x1 convs.  25 % fewer MFMAs per image than the tile
// kernels, one barrier per band, nothing reloaded: every wave keeps ONE block's weights in registers for the whole image
// (waves 0-3: W1 | W2 | W3D of block 0 = 124 VGPRs, waves 4-7: block 1), so both waves of a SIMD carry the same load, and
// they run their two phases in opposite order (group 0: conv1/conv2 then 3x3, group 1: 3x3 then conv1/conv2) so that one
// wave's convert / ReLU work sits under the other's dense 3x3 chain.
//
// Round i (one __syncthreads() each; band b = rows 4 b .. 4 b + 3):
//   group 0:  A0 (x band i -> t0 band i)            B0 (t0 bands i-3 .. i-1, x band i-2 -> y0 band i-2 [-> HBM])
//   group 1:  B1 (t1 bands i-6 .. i-4, y0 band i-5 -> y1 band i-5 -> HBM)      A1 (y0 band i-3 -> t1 band i-3)
//   all:      LDS-DMA of x band i+1
// Rings (16 rows = 4 band slots each): x 36 KB, t0 32 KB, y0 36 KB, t1 32 KB.  The phase bodies (rw_t_tile, rw_b_chain, the
// dense-K operand layout, the stores) are those of wdsr_fwd_rs.h: per pixel the same products in the same order, so the
// results are bit-identical to the tile kernels'.
#pragma once
#include "wdsr_fwd_rs.h"

template <int F_, int E_, int L_> struct StreamCfg {
  typedef BlockCfg<F_, E_, L_> C;
  typedef RsCfg<F_, E_, L_, 2> R;
  static constexpr int W = 48, BR = 4, BPX = BR * W, NTB = BPX / 32;    // band: 4 rows = 192 pixels = 6 pixel tiles
  static constexpr int RING = 16;                                      // ring rows (4 band slots)
  static constexpr int TWP = W + 2;                                    // t ring rows: one zero column either side
  static constexpr int TD = C::L, KXL = C::F;
  static constexpr int X_ELEMS = RING * W * KXL;
  static constexpr int T_ELEMS = (RING * TWP + 2) * TD;                // + slack: the 3x3's spare chunk slot reads 8 bytes past a row
  static constexpr int ONES_ELEMS = 8;
  static constexpr int LDS_BYTES = (2 * X_ELEMS + 2 * T_ELEMS + ONES_ELEMS) * 2 + 2 * R::CL_FLOATS * 4;
  static_assert(BPX % 32 == 0 && (BPX * KXL * 2) % 1024 == 0, "a band is whole pixel tiles and whole 1 KB DMA pieces");
  static_assert((X_ELEMS * 2) % 16 == 0 && (T_ELEMS * 2) % 16 == 0, "16-byte aligned regions");
  static_assert(2 * T_ELEMS + X_ELEMS >= 2 * R::W_ELEMS, "both blocks' weights are parked behind the x ring until they are in registers");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static constexpr int XPIECES = BPX * KXL * 2 / 1024;                 // 9
};

// pixel operands of the dense-K 3x3 conv out of a t RING: the three window rows are three ring rows (the ring wraps), each with
// its own base; inside a row the layout is RwBAddrD's (16 contiguous bytes per lane half and k-step, the ones chunk in the
// last slot of the last row)
template <typename C, int TWP> struct StBAddr {
  static constexpr int TD = C::L, KS = C::KS3D;
  static_assert(C::DENSE3 && KS == 12, "dense-K layout: 16 chunk slots per window row");
  typedef __attribute__((address_space(3))) const volatile bf16x4* lds_chunk_p;    // (volatile: see RwBAddrD)
  lds_chunk_p b[3];
  lds_chunk_p last;
  SR_DEV void init(const __bf16* T, const __bf16* ones, int y, int c, int hh) {    // window of image pixel (y, c): padded columns c .. c + 2
    const __bf16* g2 = nullptr;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const __bf16* g = T + (((y - 1 + ky) & 15) * TWP + c) * TD + hh * 32;
      b[ky] = (lds_chunk_p)(g);
      g2 = g;
    }
    last = (lds_chunk_p)(hh ? ones : g2 + 3 * 8 + 4);
  }
  SR_DEV bf16x8 frag(int s) const {
    const int off = (s % 4) * 2;
    const bf16x4 lo = b[s / 4][off];
    const bf16x4 hi = s == KS - 1 ? *last : b[s / 4][off + 1];
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
};

// conv1 -> ReLU -> conv2 of one band: Xin = the band's 192 pixel rows (24 channels each), t -> ring rows (y & 15), padded column
// c + 1; SAVE_T: also into the weight-gradient kernels' tile-local image [tile 12 x 24][288][LP] of this image.  Wave gw of the
// group takes pixel tiles gw and gw + 4.
template <typename S, bool SAVE_T>
SR_DEV void st_phase_a(const __bf16* Xin, const __bf16* ones, __bf16* Tring, const RwA<typename S::C>& w, const float* cl,
                       __bf16* tsave_img, int band, int H, int gw, int lane) {
  typedef typename S::C C;
  const int r = lane & 31, hh = lane >> 5;
  bf16x8 xb[C::KS1];
  int tile = gw;
  rw_x_frags<C, S::KXL>(xb, Xin, ones, tile * 32 + r, hh);
#pragma unroll 1
  for (; tile < S::NTB; tile += 4) {
    const f32x16 t = rw_t_tile<C>(xb, w, cl, hh, [](int) {});
    const int p = tile * 32 + r;
    if (tile + 4 < S::NTB) rw_x_frags<C, S::KXL>(xb, Xin, ones, p + 128, hh);      // the next tile's operands land under the stores
    const int row = p / S::W, c = p - row * S::W, y = band * S::BR + row;
    RwPix px;
    px.hp = (y & 15) * S::TWP + c + 1;
    px.valid = y < H;
    px.tso = -1;
    if constexpr (SAVE_T) {
      const int ty = y / C::TH, tx = c / C::TW;
      px.tso = px.valid ? (((ty * (S::W / C::TW) + tx) * (C::TH * C::TW)) + (y - ty * C::TH) * C::TW + (c - tx * C::TW)) * C::LP : -1;
    }
    rw_store_t<C, SAVE_T>(t, px, Tring, tsave_img, hh);
  }
}

// 3x3 conv + bias + residual of one band: t from the ring, the residual from Xres (the band's 192 pixel rows of the block
// input), y -> Ynext (the band's rows of the next block's input ring; nullptr = none) and -> the global image yout (nullptr =
// none).  Wave gw takes pixel tiles 3 - gw and 7 - gw: the waves with ONE conv1/conv2 tile per band take two 3x3 tiles.
template <typename S>
SR_DEV void st_phase_b(const __bf16* Tring, const __bf16* Xres, const __bf16* ones, __bf16* Ynext, __bf16* yout,
                       const RwB<typename S::C>& w, int band, int H, int gw, int lane) {
  typedef typename S::C C;
  typedef StBAddr<C, S::TWP> A;
  const int r = lane & 31, hh = lane >> 5;
  const bool to_global = yout != nullptr;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yout, 0, to_global ? H * S::W * C::F * 2 : 0, 0x00020000);
#pragma unroll 1
  for (int tile = 3 - gw; tile < S::NTB; tile += 4) {
    const int p = tile * 32 + r;
    const int row = p / S::W, c = p - row * S::W, y = band * S::BR + row;
    A a;
    a.init(Tring, ones, y, c, hh);
    const f32x16 acc = rw_b_chain<C, A, 4, 4>(a, w, rw_resid_init<C>(Xres + p * S::KXL, hh), [] {}, [](int) {});
    RwPixB pb;
    pb.hy = row;
    pb.hx = c;
    pb.xno = p * S::KXL;
    pb.go = (to_global && y < H) ? (unsigned)((y * S::W + c) * C::F * 2) : 0xFFFFFF00u;
    rw_store_y<C>(acc, pb, Ynext, yrs, to_global, hh);
  }
}

// grid = (N images), 512 threads; W = 48, H % 4 == 0.  x -> ya (block 0's output; nullptr = not stored) -> yb.
// tsa / tsb (SAVE_T): saved t images [N][tiles][288][LP].
template <int F, int E, int L, bool SAVE_T>
__global__ __launch_bounds__(512) void wdsr_fwd_stream_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ ya,
                                                              __bf16* __restrict__ yb, const __bf16* __restrict__ wa,
                                                              const __bf16* __restrict__ wb, const float* __restrict__ cia,
                                                              const float* __restrict__ cib, __bf16* __restrict__ tsa,
                                                              __bf16* __restrict__ tsb, int H) {
  typedef StreamCfg<F, E, L> S;
  typedef typename S::C C;
  typedef typename S::R R;
  __shared__ __attribute__((aligned(16))) char smem_raw[S::LDS_BYTES];
  __bf16* const XR = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* const T0 = XR + S::X_ELEMS;
  __bf16* const Y0 = T0 + S::T_ELEMS;
  __bf16* const T1 = Y0 + S::X_ELEMS;
  __bf16* const ONES = T1 + S::T_ELEMS;
  float* const CL = reinterpret_cast<float*>(ONES + S::ONES_ELEMS);
  __bf16* const PARK = T0;                                             // prologue only: both blocks' weight fragments
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, gw = wave & 3;
  const int n = blockIdx.x;
  const int NB = H / S::BR;
  const size_t img = (size_t)n * H * S::W * F;
  const char* const xg = reinterpret_cast<const char*>(x + img);
  const int tiles_img = ((H + C::TH - 1) / C::TH) * (S::W / C::TW);
  __bf16* const ts_img = SAVE_T ? (grp ? tsb : tsa) + (size_t)n * tiles_img * (C::TH * C::TW) * C::LP : nullptr;
  const char* zeros = reinterpret_cast<const char*>(g_sr_const_chunks) + 16;

  auto stage_x = [&](int band) {                        // a band's rows are ONE contiguous 9 KB run of the image
    const unsigned dst = lds_addr(XR) + (band & 3) * (S::BPX * S::KXL * 2);
#pragma unroll 1
    for (int p = wave; p < S::XPIECES; p += 8) dma_piece16(xg + (size_t)band * (S::BPX * S::KXL * 2) + p * 1024 + lane * 16, dst + p * 1024);
  };

  // ---- prologue: C-init tables, both blocks' weights (parked in the rings), x band 0; weights -> registers; rings zeroed ----
  stage_x(0);
#pragma unroll 1
  for (int p = R::P_C + wave; p < R::P_END; p += 8) {
    if (p < R::P_W) {
      const int k = p - R::P_C, blk = k / (R::CL_FLOATS / 64), i = (k % (R::CL_FLOATS / 64)) * 64 + lane;
      const float* tab = blk == 1 ? cib : cia;
      const char* src = i < C::CINIT_FWD ? reinterpret_cast<const char*>(tab + i) : zeros + (lane & 3) * 4;
      dma_piece4(src, lds_addr(CL) + k * 256);
    } else {
      const int fr = p - R::P_W;
      const __bf16* wsrc = fr >= R::NFR ? wb + (size_t)R::src_frag(fr - R::NFR) * 512 : wa + (size_t)R::src_frag(fr) * 512;
      dma_piece16(reinterpret_cast<const char*>(wsrc + lane * 8), lds_addr(PARK) + fr * 1024);
    }
  }
  if (tid < 8) ONES[tid] = tid == 0 ? (__bf16)1.f : (__bf16)0.f;
  wait_vmcnt<0>();
  __syncthreads();
  RwA<C> rwa;
  RwB<C> rwb;
  rwa.load(PARK + grp * R::W_ELEMS, lane);
  rwb.load(PARK + grp * R::W_ELEMS, lane);
  __syncthreads();                                     // every wave holds its block's weights: the parking area is free
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    u32x4* t0 = reinterpret_cast<u32x4*>(T0);
    u32x4* t1 = reinterpret_cast<u32x4*>(T1);
    for (int i = tid; i < S::T_ELEMS * 2 / 16; i += 512) { t0[i] = z; t1[i] = z; }
  }
  __syncthreads();
  auto zero_ring_row = [&](__bf16* T, int row) {       // one t ring row incl. its padding columns (by the four waves of a group)
    const u32x4 z = {0u, 0u, 0u, 0u};
    u32x4* q = reinterpret_cast<u32x4*>(T + row * S::TWP * S::TD);
    for (int i = gw * 64 + lane; i < S::TWP * S::TD * 2 / 16; i += 256) q[i] = z;
  };
  static_assert((S::TWP * S::TD * 2) % 16 == 0, "a t ring row is whole 16-byte pieces");
  if constexpr (SAVE_T) {
    // rows of the last 12-row tiles below the image: the weight-gradient kernels read whole tiles, and the tile kernels write
    // zeros there
    const int rows_pad = ((H + C::TH - 1) / C::TH) * C::TH - H;
    static_assert((C::LP * 2) % 16 == 0, "a saved t pixel is whole 16-byte pieces");
    constexpr int PPX = C::LP * 2 / 16;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = gw * 64 + lane; i < rows_pad * S::W * PPX; i += 256) {
      const int px = i / PPX, q = i - px * PPX;
      const int y = H + px / S::W, c = px % S::W;
      const int ty = y / C::TH, tx = c / C::TW;
      const int off = (((ty * (S::W / C::TW) + tx) * (C::TH * C::TW)) + (y - ty * C::TH) * C::TW + (c - tx * C::TW)) * C::LP + q * 8;
      stream_store(reinterpret_cast<u32x4*>(ts_img + off), z);
    }
  }

  const float* const cl = CL + grp * R::CL_FLOATS;
  // global stores a wave issues per phase of a round (3 per pixel tile): the x pieces of the next band are issued BEFORE them, so a
  // counted wait retires the pieces and leaves this round's stores in flight (vmcnt retires in issue order)
  const int st_a = SAVE_T ? 3 * (gw < 2 ? 2 : 1) : 0;
  const int st_b = 3 * (gw < 2 ? 1 : 2);
#pragma unroll 1
  for (int i = 0; i < NB + 5; ++i) {
    if (i + 1 < NB) stage_x(i + 1);
    int nst = 0;
    if (grp == 0) {
      if (i < NB) {
        st_phase_a<S, SAVE_T>(XR + (i & 3) * (S::BPX * S::KXL), ONES, T0, rwa, cl, ts_img, i, H, gw, lane);
        nst += st_a;
      }
      if (i == NB) zero_ring_row(T0, (NB * S::BR) & 15);                // the row below the image (B0 of the last band reads it next round)
      const int j = i - 2;
      if (j >= 0 && j < NB) {
        st_phase_b<S>(T0, XR + (j & 3) * (S::BPX * S::KXL), ONES, Y0 + (j & 3) * (S::BPX * S::KXL), ya ? ya + img : nullptr, rwb, j, H,
                      gw, lane);
        nst += ya ? st_b : 0;
      }
    } else {
      const int j = i - 5;
      if (j >= 0 && j < NB) {
        st_phase_b<S>(T1, Y0 + (j & 3) * (S::BPX * S::KXL), ONES, nullptr, yb + img, rwb, j, H, gw, lane);
        nst += st_b;
      }
      const int k = i - 3;
      if (k >= 0 && k < NB) {
        st_phase_a<S, SAVE_T>(Y0 + (k & 3) * (S::BPX * S::KXL), ONES, T1, rwa, cl, ts_img, k, H, gw, lane);
        nst += st_a;
      }
      if (k == NB) zero_ring_row(T1, (NB * S::BR) & 15);
    }
    if (nst >= 9) wait_vmcnt<9>();                     // this wave's x pieces of the next band have landed
    else if (nst >= 6) wait_vmcnt<6>();
    else if (nst >= 3) wait_vmcnt<3>();
    else wait_vmcnt<0>();
    __syncthreads();
  }
}
