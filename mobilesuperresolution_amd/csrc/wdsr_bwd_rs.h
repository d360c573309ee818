// Backward-data of TWO fused WDSR-B residual blocks with REGISTER-RESIDENT weights (bf16, 24 units, gfx950): the backward
// counterpart of wdsr_fwd_rs.h.  Reference op: the autograd backward of Block.forward (models/basic_wdsr_b.py:142-144) for blocks
// b (the later one) and a (the earlier one):
//     dt = conv3x3^T(dy)          h = relu(W1 x + b1) recomputed          dx = dy + W1^T [ 1(h > 0) . W2^T dt ]
// and dx_b is dy_a.  Also writes both dt images the weight-gradient kernels contract over (tile-local [tile][288][LP]).
//
// Why a rewrite of wdsr_block2_bwd_data_kernel (wdsr_block.h): that kernel gives every 32-pixel tile its own wave (12 + 9 of
// them), reads BOTH operands of every MFMA from LDS and spends 10.6 VALU instructions per MFMA (fp32 compare + select for the
// ReLU mask, an identity product for the skip connection): 13.0 us per launch, 8 launches per training step = the largest
// item of the step.  Here, as in the forward kernel: 8 waves, the phase's weights in registers (3x3^T: 14 fragments = 56 VGPRs;
// conv1 + W2^T + W1^T: 29 fragments = 116 VGPRs), everything staged by LDS-DMA, four phases
//     P1b: dt_b on the 14 x 26 region   P2b: dx_b there (-> LDS as dy_a, -> HBM on the core)   P1a: dt_a on the core   P2a: dx_a
// with dt handed from P1 to P2 through LDS in exactly the channel order the chained W2^T fragments expect (swap the 8s and
// the 4s bit of the channel index when storing an accumulator group), the skip connection as the dx accumulator's INITIAL value
// and the ReLU mask as packed 16-bit integer ops on the converted operands (min(relu(h) bits, 1) is 0 / 1 per half; a packed
// multiply keeps or clears dh's bit pattern: exactly 1(h > 0), including h = +0).  Block a's weights are streamed into block
// b's LDS copy once that is in registers; x_a lands in the dy_b buffer while P2b runs.
// Same products as the old kernel; the skip term enters the sum first instead of last: equal up to fp32 summation order.
#pragma once
#include <type_traits>
#include "wdsr_fwd_rs.h"

template <int F_, int E_, int L_> struct BwdRsCfg {
  typedef BlockCfg<F_, E_, L_> C;
  typedef BwdCfg<C> B;
  static constexpr int NW = 8, NTHREADS = 64 * NW;
  // region k: the tile plus a halo of 2 - k pixels (0: dy_b, 1: dt_b / dx_b / x_b, 2: the core: dt_a / dx_a / x_a)
  static constexpr int rw(int k) { return C::TW + 2 * (2 - k); }
  static constexpr int rh(int k) { return C::TH + 2 * (2 - k); }
  static constexpr int np(int k) { return rw(k) * rh(k); }
  static constexpr int npad(int k) { return (np(k) + 31) / 32 * 32; }
  static constexpr int KXL = C::F;                                     // 48-byte rows (24 channels), as in the forward kernel
  static constexpr int DTL = 24;                                       // dt rows: L real channels in chained order + zeros
  static_assert(C::F == 24 && C::L <= DTL && C::FOLD_B1, "24 units");
  static constexpr int PXP = 63 / C::FC;                               // pixels per 1 KB DMA piece (+ one chunk of the next)
  static constexpr int pieces(int npx) { return (npx + PXP - 1) / PXP; }
  static constexpr int dma_bytes(int npx) { return pieces(npx) * PXP * C::FC * 16 + 16; }
  static constexpr int DY_ELEMS = npad(0) * KXL;                       // dy_b on region 0; later x_a on the core
  static constexpr int DT_ELEMS = npad(1) * DTL;
  static constexpr int XB_ELEMS = npad(1) * KXL;
  static constexpr int DX_ELEMS = (npad(1) + 2) * KXL;                 // dx_b on region 1 = dy_a (+ slack rows for the padded taps)
  // the last piece of a staged region runs up to 1 KB past it: it may only hit a buffer that is written later
  static_assert(dma_bytes(npad(0)) <= (DY_ELEMS + DT_ELEMS) * 2, "dy staging spills into the (not yet written) dt image only");
  static_assert(dma_bytes(npad(1)) <= (XB_ELEMS + DX_ELEMS) * 2, "x_b staging spills into the (not yet written) dx image only");
  static_assert(dma_bytes(np(2)) <= DY_ELEMS * 2, "x_a fits the dy buffer");
  // staged fragments of one block: W1 as it lies at the head of the blob, then W3T | W2T | W1T from behind the forward section
  static constexpr int NF1 = C::NET * C::KS1, NF3 = B::KS3B, NF2T = 2 * C::NET, NF1T = C::KS2;
  static constexpr int NFR = NF1 + NF3 + NF2T + NF1T;                  // 10 + 14 + 10 + 9 = 43
  static constexpr int L_W1 = 0, L_W3T = NF1, L_W2T = L_W3T + NF3, L_W1T = L_W2T + NF2T;
  static constexpr int src_frag(int fr) { return fr < NF1 ? fr : B::W3T_OFF + (fr - NF1); }
  static_assert(B::W2T_OFF == B::W3T_OFF + NF3 && B::W1T_OFF == B::W2T_OFF + NF2T, "W3T | W2T | W1T are contiguous in the blob");
  static constexpr int W_ELEMS = NFR * 512;
  static constexpr int ONES_ELEMS = 8;
  static constexpr int LDS_BYTES = (DY_ELEMS + DT_ELEMS + XB_ELEMS + DX_ELEMS + W_ELEMS + ONES_ELEMS) * 2;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <typename R> struct BwP1 {                                    // 3x3^T weights
  static constexpr int N = R::NF3;
  bf16x8 w[N];
  SR_DEV void load_one(const __bf16* wl, int lane, int i) { if (i < N) w[i] = lds_chunk<__bf16>(wl, ((R::L_W3T + i) * 64 + lane) * 8); }
  SR_DEV void load(const __bf16* wl, int lane) {
#pragma unroll
    for (int i = 0; i < N; ++i) load_one(wl, lane, i);
  }
};
template <typename R> struct BwP2 {                                    // conv1 (recompute), W2^T, W1^T
  static constexpr int N = R::NF1 + R::NF2T + R::NF1T;
  bf16x8 w1[R::NF1], w2t[R::NF2T], w1t[R::NF1T];
  SR_DEV void load_one(const __bf16* wl, int lane, int i) {
    if (i < R::NF1) w1[i] = lds_chunk<__bf16>(wl, ((R::L_W1 + i) * 64 + lane) * 8);
    else if (i < R::NF1 + R::NF2T) w2t[i - R::NF1] = lds_chunk<__bf16>(wl, ((R::L_W2T + i - R::NF1) * 64 + lane) * 8);
    else if (i < N) w1t[i - R::NF1 - R::NF2T] = lds_chunk<__bf16>(wl, ((R::L_W1T + i - R::NF1 - R::NF2T) * 64 + lane) * 8);
  }
  SR_DEV void load(const __bf16* wl, int lane) {
#pragma unroll
    for (int i = 0; i < N; ++i) load_one(wl, lane, i);
  }
};

// ---- P1: dt on an output region (RWO wide, NPO pixels; its pixel (hy, hx) = pixel (hy + 1, hx + 1) of the dy image, which is
// RWO + 2 wide).  Chunk q = 2 s + hh of k-step s is 8 channels of the window in row-major order (9 chunks per window row: the three
// taps of a row are 72 contiguous channels of the dy image), as W3T is packed.  dt -> DT (chained channel order) and, for core
// pixels, -> the saved image (natural order, zero outside the image).  HALOO = offset of the region inside the tile frame. ----
template <typename R, int RWO, int NPO, int HALOO, typename PF>
SR_DEV void bw_phase_dt(const __bf16* DYimg, __bf16* DT, const BwP1<R>& w, __bf16* dtsave_tile, int H, int W, int ty0, int tx0, int wave,
                        int lane, PF prefetch) {
  typedef typename R::C C;
  constexpr int NT = (NPO + 31) / 32, RWI = RWO + 2, KS = R::NF3;
  static_assert(KS == 14, "27 chunks of 8 channels + one of padding");
  const int r = lane & 31, hh = lane >> 5;
  auto do_tile = [&](int tile, auto pfon) {
    const int hp = tile * 32 + r;
    const int hpc = hp < NPO ? hp : 0;
    const int hy = hpc / RWO, hx = hpc - hy * RWO;
    const __bf16* const base = DYimg + (hy * RWI + hx) * R::KXL + hh * 8;
    // chunk 2 s + hh sits 8 (2 s % 9) + (2 s / 9) RWI KXL elements behind the window's first chunk, + 8 for hh = 1 -- except
    // where 2 s + 1 starts the next window row (s = 4: q = 9) and at the padding chunk (s = 13, hh = 1: zero weights, any
    // FINITE data: the lane's first chunk)
    const __bf16* const alt4 = hh ? DYimg + ((hy + 1) * RWI + hx) * R::KXL : base + 8 * 8;
    const __bf16* const alt13 = hh ? base : base + 2 * RWI * R::KXL + 8 * 8;
    auto frag = [&](int s) {
      const int q0 = 2 * s;
      const int off = (q0 / 9) * RWI * R::KXL + (q0 % 9) * 8;
      if (s == 4) return *reinterpret_cast<const bf16x8*>(alt4);
      if (s == 13) return *reinterpret_cast<const bf16x8*>(alt13);
      return *reinterpret_cast<const bf16x8*>(base + off);
    };
    bf16x8 f[KS];
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = frag(s);
    f32x16 acc = zero16();
    SR_RS_PRIO(2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      acc = mma16<__bf16>(w.w[s], f[s], acc);
      if (s + 4 < KS) f[s + 4] = frag(s + 4);
      if constexpr (decltype(pfon)::value) prefetch(2 * s, 2 * s + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    SR_RS_PRIO(0);
    if constexpr (decltype(pfon)::value) prefetch(2 * KS, 64);
    // rows (channels) 8 g + 4 hh + k of dt in regs 4 g + k.  DT row position = the channel with its 8s and 4s bits swapped
    // (the chained k order of W2T); positions 20 .. 23 take the zero rows 20 .. 23 (lane half 1 of group 2)
    bf16x4 v[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) v[g] = acc_group<__bf16>(acc, g);
#pragma unroll
    for (int g = 0; g < 3; ++g) *reinterpret_cast<bf16x4*>(DT + hp * R::DTL + (g < 2 ? 8 * hh + 4 * g : 16 + 4 * hh)) = v[g];
    if (dtsave_tile) {
      const int cy = hy - HALOO, cx = hx - HALOO;
      if (hp < NPO && cy >= 0 && cy < C::TH && cx >= 0 && cx < C::TW) {
        if (!(ty0 + cy < H && tx0 + cx < W)) {
#pragma unroll
          for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[g][k] = (__bf16)0.f;
        }
        __bf16* o = dtsave_tile + (cy * C::TW + cx) * C::LP + hh * 4;
#pragma unroll
        for (int g = 0; g < 3; ++g) stream_store(reinterpret_cast<bf16x4*>(o + g * 8), v[g]);
      }
    }
  };
  if (wave >= NT) {                                     // (a wave without a tile still moves its share of the next weights)
    prefetch(0, 64);
    return;
  }
  do_tile(wave, std::true_type{});
#pragma unroll 1
  for (int tile = wave + R::NW; tile < NT; tile += R::NW) do_tile(tile, std::false_type{});
}

// ---- P2: dx on the same region.  x rows at Ximg (row = region pixel), dt rows at DT, dy of the pixel itself at the centre of its
// window in the dy image.  dx -> DXimg (the next block's dy: zero outside the image; nullptr = none) and -> the global image for core
// pixels inside the image. ----
template <typename R, int RWO, int NPO, int HALOO>
SR_DEV void bw_phase_dx(const __bf16* Ximg, const __bf16* ones, const __bf16* DT, const __bf16* DYimg, __bf16* DXimg, __bf16* dxout,
                        const BwP2<R>& w, int H, int W, int ty0, int tx0, int wave, int lane) {
  typedef typename R::C C;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
  constexpr int NT = (NPO + 31) / 32, RWI = RWO + 2;
  const int r = lane & 31, hh = lane >> 5;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(dxout, 0, H * W * C::F * 2, 0x00020000);
#pragma unroll 1
  for (int tile = wave; tile < NT; tile += R::NW) {
    const int hp = tile * 32 + r;
    const int hpc = hp < NPO ? hp : 0;
    const int hy = hpc / RWO, hx = hpc - hy * RWO;
    bf16x8 xb[C::KS1], dtb[2];
    rw_x_frags<C, R::KXL>(xb, Ximg, ones, hpc, hh);
#pragma unroll
    for (int s = 0; s < 2; ++s) dtb[s] = *reinterpret_cast<const bf16x8*>(DT + hpc * R::DTL + 16 * s + 8 * hh);
    f32x16 dx = rw_resid_init<C>(DYimg + ((hy + 1) * RWI + hx + 1) * R::KXL, hh);      // the skip connection: dx starts as dy
    auto conv1 = [&](int et) {
      f32x16 h = mma16<__bf16>(w.w1[et * C::KS1], xb[0], zero16());
      return mma16<__bf16>(w.w1[et * C::KS1 + 1], xb[1], h);
    };
    auto dhof = [&](int et) {
      f32x16 d = mma16<__bf16>(w.w2t[2 * et], dtb[0], zero16());
      return mma16<__bf16>(w.w2t[2 * et + 1], dtb[1], d);
    };
    SR_RS_PRIO(2);
    f32x16 h = conv1(0), dh = dhof(0);
#pragma unroll
    for (int et = 0; et < C::NET; ++et) {
      f32x16 hn = h, dhn = dh;
      if (et + 1 < C::NET) {                            // the next e-tile's products are in flight while this one is masked
        hn = conv1(et + 1);
        dhn = dhof(et + 1);
      }
      bf16x8 hr[2] = {acc_to_frag_relu<__bf16, 0>(h), acc_to_frag_relu<__bf16, 1>(h)};
      bf16x8 dq[2] = {acc_to_frag<__bf16, 0>(dh), acc_to_frag<__bf16, 1>(dh)};
#pragma unroll
      for (int s = 0; s < 2; ++s) {                     // dq *= 1(h > 0): packed 16-bit integer ops (see wdsr_wgrad_rs.h)
        const u32x4_ hq = __builtin_bit_cast(u32x4_, hr[s]);
        u32x4_ dd = __builtin_bit_cast(u32x4_, dq[s]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned m;
          asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(hq[j]), "s"(0x00010001u));
          asm volatile("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(dd[j]) : "v"(dd[j]), "v"(m));
        }
        dq[s] = __builtin_bit_cast(bf16x8, dd);
      }
      asm volatile("s_nop 4" ::: "memory");             // VALU write (inline asm: invisible to the hazard recogniser) -> MFMA read
      if (2 * et < C::KS2) dx = mma16<__bf16>(w.w1t[2 * et], dq[0], dx);
      if (2 * et + 1 < C::KS2) dx = mma16<__bf16>(w.w1t[2 * et + 1], dq[1], dx);
      h = hn;
      dh = dhn;
    }
    SR_RS_PRIO(0);
    const int Y = ty0 - HALOO + hy, X = tx0 - HALOO + hx;
    const bool live = hp < NPO, inimg = live && Y >= 0 && Y < H && X >= 0 && X < W;
    bf16x4 v[C::FC];
#pragma unroll
    for (int g = 0; g < C::FC; ++g) {
      v[g] = acc_group<__bf16>(dx, g);
      if (!inimg) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[g][k] = (__bf16)0.f;
      }
    }
    if (DXimg) {
#pragma unroll
      for (int g = 0; g < C::FC; ++g) *reinterpret_cast<bf16x4*>(DXimg + hp * R::KXL + g * 8 + hh * 4) = v[g];
    }
    const int cy = hy - HALOO, cx = hx - HALOO;
    const bool core = inimg && cy >= 0 && cy < C::TH && cx >= 0 && cx < C::TW;
    const unsigned off = core ? (unsigned)((Y * W + X) * C::F * 2) + hh * 8 : 0xFFFFFF00u;
#pragma unroll
    for (int g = 0; g < C::FC; ++g)
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, v[g]), yrs, off + g * 16, 0, 2 /* nt */);
  }
}

// grid = (tiles_y * tiles_x, N), 512 threads.  xa / xb: inputs of block a / b; dyb: gradient at block b's output;
// dxb = gradient at block b's input (= dy_a), dxa = at block a's input; dta / dtb: saved dt images (nullptr = not kept).
template <int F, int E, int L>
__global__ __launch_bounds__(512) void wdsr_bwd_rs_kernel(const __bf16* __restrict__ xa, const __bf16* __restrict__ xb,
                                                          const __bf16* __restrict__ dyb, __bf16* __restrict__ dxb,
                                                          __bf16* __restrict__ dxa, const __bf16* __restrict__ wa,
                                                          const __bf16* __restrict__ wb, __bf16* __restrict__ dta,
                                                          __bf16* __restrict__ dtb, int H, int W, int tiles_x) {
  typedef BwdRsCfg<F, E, L> R;
  typedef typename R::C C;
  __shared__ __attribute__((aligned(16))) char smem_raw[R::LDS_BYTES];
  __bf16* const DY = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* const DT = DY + R::DY_ELEMS;
  __bf16* const XB = DT + R::DT_ELEMS;
  __bf16* const DX = XB + R::XB_ELEMS;
  __bf16* const WL = DX + R::DX_ELEMS;
  __bf16* const ONES = WL + R::W_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * F;
  const size_t tile_g = (size_t)n * gridDim.x + tile;
  const char* zeros = reinterpret_cast<const char*>(g_sr_const_chunks) + 16;
  const int lq = lane / C::FC, lc = lane - lq * C::FC;  // this lane's fixed place in every activation piece
  SR_STAMP_DECL;
  SR_STAMP();

  // a region of an image (zero outside it) by LDS-DMA: 21 pixels x 3 chunks (+ 1 chunk of the next pixel) per piece
  auto stage_region = [&](__bf16* dst, const __bf16* src, int y0, int x0, int rw, int npx, int p0) {
    const int np_ = R::pieces(npx);
#pragma unroll 1
    for (int p = (wave - p0) & 7; p < np_; p += 8) {
      const int px_ = p * R::PXP + lq;
      const int py = px_ / rw, pxx = px_ - py * rw;
      const int Y = y0 + py, X = x0 + pxx;
      const char* s = zeros;
      if (px_ < npx && Y >= 0 && Y < H && X >= 0 && X < W) s = reinterpret_cast<const char*>(src + img + ((size_t)Y * W + X) * C::F + lc * 8);
      dma_piece16(s, lds_addr(dst) + p * (R::PXP * C::FC * 16));
    }
  };
  auto stage_w = [&](const __bf16* wsrc, int lo, int hi, int p0) {     // staged fragments [lo, hi)
#pragma unroll 1
    for (int fr = lo + ((wave - p0 - lo) & 7); fr < hi; fr += 8)
      dma_piece16(reinterpret_cast<const char*>(wsrc + (size_t)R::src_frag(fr) * 512 + lane * 8), lds_addr(WL) + fr * 1024);
  };
  // what P1b needs first -- the 3x3^T fragments and dy_b -- then what only P2b needs: x_b and the other 29 fragments, which land
  // underneath P1b.  The second set is exactly SET2 / 8 pieces per wave, so a counted wait retires the first set only.
  constexpr int SET2_REAL = R::pieces(R::np(1)) + R::NF1 + R::NF2T + R::NF1T;
  constexpr int SET2 = (SET2_REAL + 7) / 8 * 8;        // (padded with repeats of one fragment: the same bytes to the same place)
  stage_w(wb, R::L_W3T, R::L_W3T + R::NF3, 0);
  stage_region(DY, dyb, ty0 - 2, tx0 - 2, R::rw(0), R::np(0), 6);
  {                                                    // set 2 as ONE piece list: x_b's pieces, then W1, then W2T | W1T
    constexpr int NPX1 = R::pieces(R::np(1));
#pragma unroll 1
    for (int p = wave; p < SET2; p += 8) {
      if (p < NPX1) {
        const int px_ = p * R::PXP + lq;
        const int py = px_ / R::rw(1), pxx = px_ - py * R::rw(1);
        const int Y = ty0 - 1 + py, X = tx0 - 1 + pxx;
        const char* sp = zeros;
        if (px_ < R::np(1) && Y >= 0 && Y < H && X >= 0 && X < W) sp = reinterpret_cast<const char*>(xb + img + ((size_t)Y * W + X) * C::F + lc * 8);
        dma_piece16(sp, lds_addr(XB) + p * (R::PXP * C::FC * 16));
      } else {
        const int k0 = p - NPX1, k = k0 < R::NF1 + R::NF2T + R::NF1T ? k0 : 0, fr = k < R::NF1 ? k : R::L_W2T + (k - R::NF1);
        dma_piece16(reinterpret_cast<const char*>(wb + (size_t)R::src_frag(fr) * 512 + lane * 8), lds_addr(WL) + fr * 1024);
      }
    }
  }
  if (tid < 8) ONES[tid] = tid == 0 ? (__bf16)1.f : (__bf16)0.f;
  // rows of the dx image past region 1 (read by the padded taps of P1a, written by no tile)
  for (int i = tid; i < (R::npad(1) + 2 - R::np(1)) * R::KXL / 8; i += R::NTHREADS) {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    *reinterpret_cast<bf16x8*>(DX + R::np(1) * R::KXL + i * 8) = z;
  }
  SR_STAMP();
  wait_vmcnt<SET2 / 8>();                              // set 1 landed (this wave's pieces; the barrier joins the waves)
  __syncthreads();
  SR_STAMP();

  BwP1<R> w1;
  BwP2<R> w2;
  w1.load(WL, lane);
  auto pf2 = [&](int lo, int hi) {
#pragma unroll
    for (int i = 0; i < BwP2<R>::N; ++i)
      if (i >= lo && i < hi) w2.load_one(WL, lane, i);
  };
  __bf16* const dtb_tile = dtb ? dtb + tile_g * (C::TH * C::TW) * C::LP : nullptr;
  __bf16* const dta_tile = dta ? dta + tile_g * (C::TH * C::TW) * C::LP : nullptr;
  // ---- block b ----
  bw_phase_dt<R, R::rw(1), R::np(1), 1>(DY, DT, w1, dtb_tile, H, W, ty0, tx0, wave, lane, [](int, int) {});
  SR_STAMP();
  wait_vmcnt<3>();                                     // set 2 (issued before this wave's >= 3 stores of the saved dt image ... or none)
  if (!dtb_tile) wait_vmcnt<0>();
  __syncthreads();                                     // dt_b complete; x_b and block b's other weights landed; dy_b's window reads are over
  w2.load(WL, lane);
  __syncthreads();                                     // every wave holds block b's weights: their LDS copy is free
  stage_w(wa, 0, R::NFR, 0);                           // block a's weights land underneath P2b
  SR_STAMP();
  // (P2b still reads dy_b's centre pixels from DY: x_a goes there only after P2b)
  bw_phase_dx<R, R::rw(1), R::np(1), 1>(XB, ONES, DT, DY, DX, dxb + img, w2, H, W, ty0, tx0, wave, lane);
  SR_STAMP();
  static_assert((R::np(1) + 31) / 32 >= R::NW, "every wave stores at least one tile of dx_b after its weight pieces");
  wait_vmcnt<3>();                                     // this wave's weight pieces (issued before its >= 3 dx stores) have landed
  __syncthreads();                                     // dx_b (= dy_a) complete in LDS; block a's weights landed; DY is free
  SR_STAMP();
  stage_region(DY, xa, ty0, tx0, R::rw(2), R::np(2), 0);
  // ---- block a ----
  w1.load(WL, lane);
  bw_phase_dt<R, R::rw(2), R::np(2), 0>(DX, DT, w1, dta_tile, H, W, ty0, tx0, wave, lane, pf2);
  SR_STAMP();
  static_assert((R::np(2) + 31) / 32 >= R::NW, "every wave has a tile of dt_a");
  if (dta_tile) wait_vmcnt<3>();                       // x_a's pieces, issued before this wave's >= 3 stores of the saved dt image
  else wait_vmcnt<0>();
  __syncthreads();                                     // dt_a complete, x_a landed
  SR_STAMP();
  bw_phase_dx<R, R::rw(2), R::np(2), 0>(DY, ONES, DT, DX, nullptr, dxa + img, w2, H, W, ty0, tx0, wave, lane);
  SR_STAMP();
}
