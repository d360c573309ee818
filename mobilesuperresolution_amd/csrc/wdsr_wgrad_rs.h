// Weight gradients of conv1 / conv2 of every residual block from the saved dt images, second design (bf16, 24 units).
// Reference op: the autograd backward of Block.body[0] and body[2] (models/basic_wdsr_b.py:108-144): dW1, db1, dW2, db2.
// Products and slab layout are those of wdsr_block_wgrad_saved_kernel<ROLE 0> (wdsr_block.h); what changes is who does what:
//   * there, 15 waves = 5 e-tiles x 3 pixel splits: every wave re-reads the SAME x / dt pixel fragments its e-tile needs
//     (60 KB of LDS reads per 32-pixel tile over the workgroup, against 40 MFMAs) and a spatial tile's 9 pixel tiles never
//     divide evenly over the splits; the kernel sat at 29 % MFMA busy with half of its LDS cycles in bank conflicts;
//   * here a wave owns ALL e-tiles of its pixel tiles (10 accumulator tiles = 160 registers, 8 waves x 256 VGPRs): the four
//     pixel-major and four transposed fragments of a pixel tile are read once and serve the 40 MFMAs (28 KB with the weight
//     fragments); the layer's pixels are taken as ONE stream in the saved images' own order ([tile][288 px], contiguous), in
//     chunks of 256 pixels = one 32-pixel tile per wave, so every wave does the same work between two barriers;
//   * both images are staged by LDS-DMA (no registers, no store phase): dt as it lies, x gathered per pixel, with constant
//     16-byte chunks for the "ones" channel and the zero padding;
//   * the bias gradients cost nothing: db1 is the row of dW1 that belongs to x's ones channel, and db2 is the column of dW2
//     that belongs to one padding column of the last e-tile (E = 144 of 160), whose h is forced to 1.
#pragma once
#include "wdsr_fwd_rs.h"

template <int F, int E, int L> struct WgradA8Cfg {
  typedef BlockCfg<F, E, L> C;
  typedef BwdCfg<C> B;
  static constexpr int NWAVES = 8, NTHREADS = 64 * NWAVES, CHUNK = 32 * NWAVES;      // pixels per chunk
  static constexpr int XROW = C::KX, DROW = 32;                                        // LDS row widths (elements)
  static constexpr int X_ELEMS = CHUNK * XROW, D_ELEMS = CHUNK * DROW, BUF_ELEMS = X_ELEMS + D_ELEMS;
  static constexpr int NW1 = C::NET * C::KS1, NWL = NW1 + 2 * C::NET;
  static constexpr int LDS_BYTES = (2 * BUF_ELEMS + NWL * 512) * 2;
  static constexpr int PAD_COL = E % 32;                                               // first padding column of the last e-tile
  static_assert(C::FOLD_B1 && C::KX == 32 && C::KS1 == 2, "x rows carry the ones channel (24 units)");
  static_assert(E % 32 != 0, "needs a padding column in the last e-tile for db2");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert(2 * BUF_ELEMS * 2 >= NWAVES * 2 * 4096, "the epilogue reuses the staging buffers for one e-tile of every wave");
};

template <int F, int E, int L>
__global__ __launch_bounds__((WgradA8Cfg<F, E, L>::NTHREADS)) void wdsr_wgrad_a8_kernel(
    const __bf16* __restrict__ act, const __bf16* __restrict__ side, const __bf16* __restrict__ wblob, float* __restrict__ partial,
    int N, int H, int W, int tiles_x, int tiles_per_img, long act_ls, long side_ls, long w_ls) {
  typedef __bf16 T;
  typedef WgradA8Cfg<F, E, L> G;
  typedef typename G::C C;
  typedef typename G::B B;
  __shared__ __attribute__((aligned(16))) char smem_raw[G::LDS_BYTES];
  T* const BUF = reinterpret_cast<T*>(smem_raw);
  T* const WL = BUF + 2 * G::BUF_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int layer = blockIdx.y;
  act += (size_t)layer * act_ls;
  side += (size_t)layer * side_ls;
  wblob += (size_t)layer * w_ls;
  stage_weights<T, G::NTHREADS>(WL, wblob + (size_t)C::W1_OFF * 512, G::NW1, tid);
  stage_weights<T, G::NTHREADS>(WL + G::NW1 * 512, wblob + (size_t)B::W2N_OFF * 512, 2 * C::NET, tid);
  WSrc<T, true> wsrc;
  wsrc.p = WL;

  const int total_tiles = N * tiles_per_img;
  const int total_px = total_tiles * B::NPXC;                          // (host side: N * tiles * 288 < 2^31)
  const int nchunks = (total_px + G::CHUNK - 1) / G::CHUNK;
  const char* ones = reinterpret_cast<const char*>(g_sr_const_chunks);
  const char* zeros = ones + 16;
  // One chunk = 16 pieces of x + 16 pieces of dt, two of each per wave; a piece = 16 pixels x 4 chunks of 16 bytes
  // (lane & 3 = chunk).  A chunk of 256 consecutive saved pixels touches at most two spatial tiles: everything that needs
  // a division by a run-time value is done once per chunk on wave-uniform values, the lanes only split q into (row, column).
  auto stage = [&](int chunk, int buf) {
    T* Xb = BUF + buf * G::BUF_ELEMS;
    T* Db = Xb + G::X_ELEMS;
    const int g0 = chunk * G::CHUNK;
    const int t0 = g0 / B::NPXC, q0 = g0 - t0 * B::NPXC;
    const int n0 = t0 / tiles_per_img, tile0 = t0 - n0 * tiles_per_img;
    const bool roll = tile0 + 1 == tiles_per_img;
    const int n1 = roll ? n0 + 1 : n0, tile1 = roll ? 0 : tile0 + 1;
    const int by0 = (tile0 / tiles_x) * C::TH, bx0 = (tile0 % tiles_x) * C::TW;
    const int by1 = (tile1 / tiles_x) * C::TH, bx1 = (tile1 % tiles_x) * C::TW;
    const int c = lane & 3, lp = lane >> 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave + i * G::NWAVES, k = piece * 16 + lp;
      {                                              // x: NHWC pixel of the saved order, channels 0..23 | ones | zeros
        int q = q0 + k;
        const bool wrap = q >= B::NPXC;
        q -= wrap ? B::NPXC : 0;
        const int oy = q / C::TW, ox = q - oy * C::TW;
        const int Y = (wrap ? by1 : by0) + oy, X = (wrap ? bx1 : bx0) + ox, n = wrap ? n1 : n0;
        const char* src = zeros;
        if (c == 3) src = ones;
        else if (t0 + (wrap ? 1 : 0) < total_tiles && Y < H && X < W)
          src = reinterpret_cast<const char*>(act + (((size_t)n * H + Y) * W + X) * F + c * 8);
        dma_piece16(src, lds_addr(Xb) + piece * 1024);
      }
      {                                              // dt: [pixel][LP] as saved, rows padded to 32 channels
        const int g = g0 + k;
        const char* src = (c < C::CPT && g < total_px) ? reinterpret_cast<const char*>(side + (size_t)g * C::LP + c * 8) : zeros;
        dma_piece16(src, lds_addr(Db) + piece * 1024);
      }
    }
  };

  f32x16 accA[C::NET], accB[C::NET];
#pragma unroll
  for (int et = 0; et < C::NET; ++et) { accA[et] = zero16(); accB[et] = zero16(); }

  int cur = 0;
  if ((int)blockIdx.x < nchunks) stage(blockIdx.x, 0);
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    wait_vmcnt<0>();
    __syncthreads();                                 // chunk ch (and, first time, the weights) landed; the other buffer is free
    if (ch + (int)gridDim.x < nchunks) stage(ch + gridDim.x, cur ^ 1);
    const T* XC = BUF + cur * G::BUF_ELEMS;
    const T* IMG = XC + G::X_ELEMS;
    const int pc = wave * 32 + r;                    // this lane's pixel row for the pixel-major fragments
    auto rowx = [=](int p) { return (wave * 32 + p) * G::XROW; };
    auto rowi = [=](int p) { return (wave * 32 + p) * G::DROW; };
    bf16x8 xa[2], da[2], xt[2], dtt[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      xa[s] = lds_chunk<T>(XC, pc * G::XROW + (2 * s + hh) * 8);
      da[s] = lds_chunk<T>(IMG, pc * G::DROW + (2 * s + hh) * 8);
      xt[s] = tr_frag<T>(XC, s, lane, rowx);
      dtt[s] = tr_frag<T>(IMG, s, lane, rowi);
    }
#pragma unroll
    for (int et = 0; et < C::NET; ++et) {
      f32x16 h2 = mma16<T>(xa[0], wsrc.get(et * 2, lane), zero16());
      h2 = mma16<T>(xa[1], wsrc.get(et * 2 + 1, lane), h2);
      f32x16 dh2 = mma16<T>(da[0], wsrc.get(G::NW1 + 2 * et, lane), zero16());
      dh2 = mma16<T>(da[1], wsrc.get(G::NW1 + 2 * et + 1, lane), dh2);
      // relu(h) and dh as bf16 fragments; the ReLU mask 1[h > 0] is applied to the PACKED dh: min(relu(h) bits, 1) is 0 / 1
      // per 16-bit half and a packed integer multiply keeps or clears dh's bit pattern -- 2 packed ops per 2 values where
      // compare + select on the fp32 accumulators cost 2 per value (this kernel is bound by VALU issue: a 32-cycle MFMA
      // hides about four VALU ops, and it had 12.6 per MFMA)
      bf16x8 hr[2] = {acc_to_frag_relu<T, 0>(h2), acc_to_frag_relu<T, 1>(h2)};
      bf16x8 dq[2] = {acc_to_frag<T, 0>(dh2), acc_to_frag<T, 1>(dh2)};
      if (et == C::NET - 1 && r == G::PAD_COL) {     // padding column e = E: relu(h) := 1, so that dW2's column there sums dt -> db2
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) hr[s][j] = (__bf16)1.f;
      }
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
      for (int s = 0; s < 2; ++s) {                  // (inline asm: hipcc expands the vector forms into compares, selects and byte permutes)
        const u32x4 hq = __builtin_bit_cast(u32x4, hr[s]);
        u32x4 dd = __builtin_bit_cast(u32x4, dq[s]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned m;
          asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(hq[j]), "s"(0x00010001u));
          asm volatile("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(dd[j]) : "v"(dd[j]), "v"(m));
        }
        dq[s] = __builtin_bit_cast(bf16x8, dd);
      }
      // the hazard recogniser does not see what the asm statements wrote: give the MFMAs that read dq the wait states a
      // VALU write -> MFMA read needs (without them the results were not reproducible)
      asm volatile("s_nop 4" ::: "memory");
      accA[et] = mma16<T>(xt[0], dq[0], accA[et]);
      accA[et] = mma16<T>(xt[1], dq[1], accA[et]);
      accB[et] = mma16<T>(dtt[0], hr[0], accB[et]);
      accB[et] = mma16<T>(dtt[1], hr[1], accB[et]);
    }
    cur ^= 1;
  }

  // ---- reduce the eight waves' accumulators, one e-tile at a time, through the staging buffers ----
  float* out = partial + ((size_t)layer * gridDim.x + blockIdx.x) * B::SLAB_A;
  float* red = reinterpret_cast<float*>(smem_raw);   // [wave][2 tiles][16 regs][64 lanes]
  wait_vmcnt<0>();
#pragma unroll
  for (int et = 0; et < C::NET; ++et) {
    __syncthreads();
    slab_store_tile(red + wave * 2048, 0, accA[et], lane);
    slab_store_tile(red + wave * 2048, 1, accB[et], lane);
    __syncthreads();
    for (int i = tid; i < 2048; i += G::NTHREADS) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < G::NWAVES; ++w) v += red[w * 2048 + i];
      const int which = i >> 10, reg = (i >> 6) & 15, ln = i & 63;      // which: 0 = dW1^T tile, 1 = dW2 tile
      out[((which ? C::NET + et : et) * 16 + reg) * 64 + ln] = v;
      // db1[e] = row 24 (x's ones channel) of dW1^T: register 12 of the lanes with hh = 0, column = lane
      if (which == 0 && reg == 12 && ln < 32) out[2 * C::NET * 1024 + et * 32 + ln] = v;
      // db2[l] = column PAD_COL of the last dW2 tile: lane (PAD_COL, hh), register i -> row (i & 3) + 8 (i >> 2) + 4 hh
      if (which == 1 && et == C::NET - 1 && (ln & 31) == G::PAD_COL)
        out[2 * C::NET * 1024 + C::NET * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5)] = v;
    }
  }
}

// =============================================================================================
// Weight gradients of the 3x3 conv (dW3, and b3 through t's ones channel) from the saved t images, second design.
// wdsr_block_wgrad_saved_kernel<ROLE 1> gives every tap its own wave: nine waves read the SAME transposed t fragments and
// each MFMA fetches both of its operands from LDS (2 KB per MFMA; the kernel sat at 21 % MFMA busy).  Here a wave owns all
// nine taps (144 accumulator registers) of one 4x8 pixel tile of the spatial tile: t's two fragments are read once for
// 18 MFMAs, only dy's shifted fragments are per tap.  A 12x24 tile has nine pixel tiles for eight waves: the ninth is
// split by TAP (every wave holds every tap's accumulator), one tap per wave and two for wave 0, so the waves stay level.
// Both operands are staged by LDS-DMA, t as it lies, dy on its 1-pixel halo with 48-byte rows (conflict-free).
// Slab layout: that of ROLE 1 ([tap][16 regs][64 lanes]).
// =============================================================================================
template <int F, int E, int L> struct WgradB8Cfg {
  typedef BlockCfg<F, E, L> C;
  typedef BwdCfg<C> B;
  static constexpr int NWAVES = 8, NTHREADS = 64 * NWAVES;
  static constexpr int IMG_ELEMS = (B::NPXC + 1) * 32;                 // t: [core px][32 ch]
  static constexpr int PXP = 64 / C::FC;                               // dy pixels per DMA piece: 21 (63 chunks + 1 of the next pixel) / 16
  static constexpr int NPD = (C::NPXH + PXP - 1) / PXP;                // dy pieces per tile
  static constexpr int DY_ELEMS = (NPD * PXP + 1) * C::F + 32;         // dy halo tile, rows of F, + the last piece's spill
  static constexpr int NPI = B::NPXC / 16;                             // t pieces per tile (16 px x 4 chunks)
  static constexpr int BUF_ELEMS = IMG_ELEMS + DY_ELEMS;
  static constexpr int LDS_BYTES = 2 * BUF_ELEMS * 2;
  static_assert((C::FC == 3 || C::FC == 4) && C::NPT_O == 9 && B::NPXC % 16 == 0, "24 or 32 units, 12x24 tiles");
  static_assert(LDS_BYTES >= NWAVES * 2 * 4096, "the epilogue reduces two taps of every wave at a time in the staging buffers");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int F, int E, int L>
__global__ __launch_bounds__((WgradB8Cfg<F, E, L>::NTHREADS)) void wdsr_wgrad_b8_kernel(
    const __bf16* __restrict__ dyact, const __bf16* __restrict__ side, float* __restrict__ partial, int N, int H, int W,
    int tiles_x, int tiles_per_img, long act_ls, long side_ls) {
  typedef __bf16 T;
  typedef WgradB8Cfg<F, E, L> G;
  typedef typename G::C C;
  typedef typename G::B B;
  __shared__ __attribute__((aligned(16))) char smem_raw[G::LDS_BYTES];
  T* const BUF = reinterpret_cast<T*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int layer = blockIdx.y;
  dyact += (size_t)layer * act_ls;
  side += (size_t)layer * side_ls;
  const int total = N * tiles_per_img;
  const char* zeros = reinterpret_cast<const char*>(g_sr_const_chunks) + 16;

  auto stage = [&](int t, int buf) {
    T* IMGb = BUF + buf * G::BUF_ELEMS;
    T* DYb = IMGb + G::IMG_ELEMS;
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
    const T* sp = side + (size_t)t * (B::NPXC * C::LP);
    const T* dimg = dyact + (size_t)n * H * W * F;
#pragma unroll 1
    for (int p = wave; p < G::NPI + G::NPD; p += G::NWAVES) {
      if (p < G::NPI) {                              // t: 16 pixels x (3 chunks + a zero chunk) per piece, rows of 32 channels
        const int c = lane & 3, px = p * 16 + (lane >> 2);
        const char* src = c < C::CPT ? reinterpret_cast<const char*>(sp + (size_t)px * C::LP + c * 8) : zeros;
        dma_piece16(src, lds_addr(IMGb) + p * 1024);
      } else {                                       // dy on the 1-pixel halo: 21 pixels x 3 chunks (+ 1 chunk the next piece rewrites), or 16 x 4
        const int q = p - G::NPI, lq = lane / C::FC, lc = lane - lq * C::FC;
        const int hp = q * G::PXP + lq;
        const int hy = hp / C::HW, hx = hp - hy * C::HW;
        const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
        const char* src = zeros;
        if (hp < C::NPXH && Y >= 0 && Y < H && X >= 0 && X < W)
          src = reinterpret_cast<const char*>(dimg + ((size_t)Y * W + X) * F + lc * 8);
        dma_piece16(src, lds_addr(DYb) + q * (G::PXP * C::F * 2));
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int u = 0; u < 9; ++u) acc[u] = zero16();

  int cur = 0;
  if ((int)blockIdx.x < total) stage(blockIdx.x, 0);
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    wait_vmcnt<0>();
    __syncthreads();
    if (t + (int)gridDim.x < total) stage(t + gridDim.x, cur ^ 1);
    const T* IMG = BUF + cur * G::BUF_ELEMS;
    const T* DYs = IMG + G::IMG_ELEMS;
    {                                                // this wave's own pixel tile: all nine taps
      const int ot = wave;
      const int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
      auto rowi = [=](int p) { return ((toy + (p >> 3)) * C::TW + tox + (p & 7)) * 32; };
      const bf16x8 a0 = tr_frag<T>(IMG, 0, lane, rowi), a1 = tr_frag<T>(IMG, 1, lane, rowi);
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const int uy = u / 3, ux = u - uy * 3;
        auto rowd = [=](int p) { return ((toy + (p >> 3) + uy) * C::HW + tox + (p & 7) + ux) * C::F; };
        acc[u] = mma16<T>(a0, tr_frag<T>(DYs, 0, lane, rowd), acc[u]);
        acc[u] = mma16<T>(a1, tr_frag<T>(DYs, 1, lane, rowd), acc[u]);
      }
    }
    {                                                // the ninth pixel tile, split by tap: tap = wave (and tap 8 for wave 0)
      constexpr int ot = 8;
      constexpr int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
      auto rowi = [=](int p) { return ((toy + (p >> 3)) * C::TW + tox + (p & 7)) * 32; };
      const bf16x8 a0 = tr_frag<T>(IMG, 0, lane, rowi), a1 = tr_frag<T>(IMG, 1, lane, rowi);
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        if (u == wave || (u == 8 && wave == 0)) {    // wave-uniform
          const int uy = u / 3, ux = u - uy * 3;
          auto rowd = [=](int p) { return ((toy + (p >> 3) + uy) * C::HW + tox + (p & 7) + ux) * C::F; };
          acc[u] = mma16<T>(a0, tr_frag<T>(DYs, 0, lane, rowd), acc[u]);
          acc[u] = mma16<T>(a1, tr_frag<T>(DYs, 1, lane, rowd), acc[u]);
        }
      }
    }
    cur ^= 1;
  }

  // ---- reduce the eight waves' accumulators, two taps at a time, through the staging buffers ----
  float* out = partial + ((size_t)layer * gridDim.x + blockIdx.x) * B::SLAB_B;
  float* red = reinterpret_cast<float*>(smem_raw);   // [wave][2 taps][16 regs][64 lanes]
  wait_vmcnt<0>();
#pragma unroll
  for (int u0 = 0; u0 < 9; u0 += 2) {
    __syncthreads();
    slab_store_tile(red + wave * 2048, 0, acc[u0], lane);
    if (u0 + 1 < 9) slab_store_tile(red + wave * 2048, 1, acc[u0 + 1 < 9 ? u0 + 1 : 8], lane);
    __syncthreads();
    const int nval = u0 + 1 < 9 ? 2048 : 1024;
    for (int i = tid; i < nval; i += G::NTHREADS) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < G::NWAVES; ++w) v += red[w * 2048 + i];
      out[u0 * 1024 + i] = v;
    }
  }
}
