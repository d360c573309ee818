// Forward of one or two fused WDSR-B residual blocks with REGISTER-RESIDENT weights (bf16, gfx950).
// Reference op: Block.forward, models/basic_wdsr_b.py:108-144, applied NBLK times.
//
// Why a second forward design (measured on MI355X: per-wave stamps of the diagnostic build, tools/ubench/*):
//   * the one-tile-per-wave kernels of wdsr_block.h read BOTH operands of every MFMA from LDS (1 KiB each); the LDS
//     delivers 256 B/clk per CU and the four SIMDs can take one 32x32x16 MFMA per 8 clk between them, so those
//     kernels need all of the LDS bandwidth to reach the matrix rate and sit at ~45 % of either.  Here every wave
//     keeps the weights of its current phase in registers (conv1 + conv2: 76 VGPRs, 3x3: 60 at 24 units) and
//     only the pixel operand (x, t) and the t / y hand-off go through LDS;
//   * a lone wave can issue ~4 VALU ops per MFMA for free and pays 8 cycles for every further one, and the
//     convert + ReLU between conv1 and conv2 is exactly 4 per MFMA: the products are issued in the order
//     G1a(et+1) cvt G1b(et+1) relu G2b(et-1) cvt G2a(et) relu (sched_group_barrier pins it), two waves per SIMD
//     run the same phase so that one wave's remaining VALU work (addresses, masks, stores) hides under the other's
//     MFMAs, and per-tile bookkeeping is hoisted out of the MFMA stream;
//   * everything that comes from HBM / L2 is staged by LDS-DMA (no registers, no per-iteration wait): the weights
//     as they lie, the halo'd x tile by per-lane source addresses, with two 16-byte constants standing in for the
//     zero padding and for the "ones" channel that carries conv1's bias.
// Products, k-order and roundings are those of wdsr_block_fwd_kernel: results are bit-identical to it.
#pragma once
#include <type_traits>
#include "wdsr_block.h"

#ifdef SR_RS_NO_SETPRIO
#define SR_RS_PRIO(p) do {} while (0)
#else
#define SR_RS_PRIO(p) __builtin_amdgcn_s_setprio(p)
#endif
#ifndef SR_RS_PRIO_A
#define SR_RS_PRIO_A 2
#endif
#ifndef SR_RS_PRIO_B
#define SR_RS_PRIO_B 2
#endif

// 16-byte chunks the staging DMA reads for lanes without pixel data: [0..7] = ones chunk (bf16 1.0, then
// zeros), [8..15] = zeros
__device__ __attribute__((aligned(16))) const unsigned short g_sr_const_chunks[16] = {0x3F80, 0, 0, 0, 0, 0, 0, 0,
                                                                                      0,      0, 0, 0, 0, 0, 0, 0};

// ---- LDS-DMA issued from inline asm: hipcc neither sees these operations nor waits for them, so the kernel can
// leave the later phases' weights in flight across the first barrier and retire them with COUNTED s_waitcnt vmcnt(N)
// (vmcnt retires in issue order; younger stores only make a counted wait conservative).  M0 (the LDS base of the
// 64-lane piece) is written in the same statement that uses it. ----
SR_DEV unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
SR_DEV void dma_piece16(const void* gsrc, unsigned lds_base) {     // 64 lanes x 16 B -> lds_base + lane * 16
  unsigned keep;
  const unsigned base = __builtin_amdgcn_readfirstlane(lds_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(base) : "memory");
}
SR_DEV void dma_piece4(const void* gsrc, unsigned lds_base) {      // 64 lanes x 4 B -> lds_base + lane * 4
  unsigned keep;
  const unsigned base = __builtin_amdgcn_readfirstlane(lds_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(base) : "memory");
}
template <int N> SR_DEV void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
// pieces [lo, hi) of a global piece list dealt round-robin to NW waves: how many does wave w issue?
constexpr int pieces_of_wave(int lo, int hi, int w, int nw) {
  int c = 0;
  for (int p = lo; p < hi; ++p) c += (p % nw == w) ? 1 : 0;
  return c;
}

template <int F_, int E_, int L_, int NBLK_> struct RsCfg {
  typedef BlockCfg<F_, E_, L_> C;
  static constexpr int NBLK = NBLK_, NWAVES = 8, NTHREADS = 64 * NWAVES;
  // region k (k = 0 .. NBLK): the tile plus a halo of NBLK - k pixels; block k maps region k -> region k + 1
  static constexpr int rw(int k) { return C::TW + 2 * (NBLK - k); }
  static constexpr int rh(int k) { return C::TH + 2 * (NBLK - k); }
  static constexpr int np(int k) { return rw(k) * rh(k); }
  static constexpr int npad(int k) { return (np(k) + 31) / 32 * 32; }
  // LDS rows of the block inputs hold the F real channels only (48 bytes at 24 units: a conflict-free stride for
  // 16-byte reads, where the 64-byte rows of wdsr_block.h are 4-way conflicted for reads and 8-way for the 8-byte
  // hand-off stores: 67 % of the LDS-active cycles of phase B were conflicts); the "ones" chunk that carries conv1's
  // bias is ONE 16-byte constant in LDS that every lane half needing it reads (a broadcast)
  static constexpr int KXL = C::F;
  static constexpr int X0_ELEMS = npad(0) * KXL;                       // staged x
  static_assert(C::DENSE3, "the register-resident forward runs the 3x3 conv in its dense-K form");
  static constexpr int TD = C::TD;                                     // t channels per LDS row: the real ones (40 bytes at L = 20; 26 + 2 at 32 units)
  static constexpr int TT_ELEMS = npad(0) * TD;                        // t of the current block
  static constexpr int X1_ELEMS = NBLK > 1 ? npad(1) * KXL : 0;        // block 0's output = block 1's input
  static constexpr int ONES_ELEMS = 8;                                 // [1, 0, 0, 0 | 0, 0, 0, 0]: conv1's ones chunk (16 B) = the 3x3's ones chunk + zero chunk (8 B each)
  // fragments staged per block: W1 | W2 as they lie at the head of the blob, then W3D from its tail
  static constexpr int NFR_A = C::W2_OFF + C::KS2, NFR = NFR_A + C::KS3D, W3D_SRC = BwdCfg<C>::NFRAG;
  static constexpr int W_ELEMS = NFR * 512;
  static constexpr int src_frag(int fr) { return fr < NFR_A ? fr : W3D_SRC + (fr - NFR_A); }
  static constexpr int CL_FLOATS = (C::CINIT_FWD + 63) / 64 * 64;      // whole 64-float DMA pieces per block
  // one weight region per block where they fit (24 units); otherwise (32 units, two blocks: 2 x 42 KB) ONE region that block 1's
  // fragments take over phase by phase once block 0's phase is through with them (wdsr_fwd_rs16_kernel)
  static constexpr bool REUSE_W = (X0_ELEMS + TT_ELEMS + X1_ELEMS + NBLK * W_ELEMS + ONES_ELEMS) * 2 + NBLK * CL_FLOATS * 4 > 160 * 1024;
  static constexpr int W_REGIONS = REUSE_W ? 1 : NBLK;
  static constexpr int LDS_BYTES = (X0_ELEMS + TT_ELEMS + X1_ELEMS + W_REGIONS * W_ELEMS + ONES_ELEMS) * 2 + NBLK * CL_FLOATS * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  // staging piece list (one piece = one wave-instruction of LDS-DMA), dealt round-robin to the waves in this order:
  // x region | C-init tables | block 0's weights || block 1's weights.  Only what precedes `||` is waited for
  // before the first phase; the rest lands underneath it.
  // x pieces: 21 pixels each (63 chunks) + the first chunk of the 22nd, which the next piece rewrites with the same
  // bytes: every lane then owns a FIXED (pixel offset, chunk) pair and a piece costs one division by the row width
  // (64 arbitrary chunks per piece cost two divisions and the staging loop was VALU-bound: 275 of a wave's 714 VALU ops)
  // (32 units: 16 pixels of 4 chunks = 64 lanes exactly)
  static constexpr int PXP = (C::FC == 3 ? 63 : 64) / C::FC, P_X = 0, NPX = (npad(0) + PXP - 1) / PXP;
  static_assert(C::FC == 3 || C::FC == 4, "piece geometry: 3 or 4 chunks per pixel");
  static_assert(NPX * PXP * C::FC * 16 + 16 <= (X0_ELEMS + TT_ELEMS) * 2, "the last x piece may spill into the (not yet written) t image only");
  static constexpr int P_C = P_X + NPX, NPC = NBLK * (CL_FLOATS / 64);
  static constexpr int P_W = P_C + NPC, P_W1 = P_W + NFR, P_END = P_W + NBLK * NFR;
};

// register-resident weights of one phase
template <typename C> struct RwA {
  static constexpr int N = C::NET * C::KS1 + C::KS2;
  bf16x8 w1[C::NET * C::KS1], w2[C::KS2];
  SR_DEV void load_one(const __bf16* wl, int lane, int i) {            // i: compile-time after unrolling
    if (i < C::NET * C::KS1) w1[i] = lds_chunk<__bf16>(wl, ((C::W1_OFF + i) * 64 + lane) * 8);
    else if (i < N) w2[i - C::NET * C::KS1] = lds_chunk<__bf16>(wl, ((C::W2_OFF + i - C::NET * C::KS1) * 64 + lane) * 8);
  }
  SR_DEV void load(const __bf16* wl, int lane) {
#pragma unroll
    for (int i = 0; i < C::NET * C::KS1; ++i) w1[i] = lds_chunk<__bf16>(wl, ((C::W1_OFF + i) * 64 + lane) * 8);
#pragma unroll
    for (int i = 0; i < C::KS2; ++i) w2[i] = lds_chunk<__bf16>(wl, ((C::W2_OFF + i) * 64 + lane) * 8);
  }
};
template <typename C> struct RwB {
  static constexpr int N = C::KS3D, OFF = C::W2_OFF + C::KS2;          // W3D sits behind W1 | W2 in the staged copy
  bf16x8 w3[N];
  SR_DEV void load_one(const __bf16* wl, int lane, int i) {
    if (i < N) w3[i] = lds_chunk<__bf16>(wl, ((OFF + i) * 64 + lane) * 8);
  }
  SR_DEV void load(const __bf16* wl, int lane) {
#pragma unroll
    for (int i = 0; i < N; ++i) w3[i] = lds_chunk<__bf16>(wl, ((OFF + i) * 64 + lane) * 8);
  }
};

// t^T for one 32-pixel tile, written in the order one wave should issue it: per e-tile
//   G1a(et+1) | cvt f0(et) | G1b(et+1) | relu f0(et) | G2b(et-1) | cvt f1(et) | G2a(et) | relu f1(et)
// i.e. four MFMAs with four VALU ops in each gap, every MFMA independent of the VALU ops right before it, and no
// VALU op closer than two MFMAs behind the product it reads.  The conv2 accumulation order (a0, b0, a1, b1, ...)
// is that of t_from_xb: bit-identical results.
template <typename C, int PFN = 0, typename PF>
SR_DEV f32x16 rw_t_tile(const bf16x8 (&xb)[C::KS1], const RwA<C>& w, const float* cl, int hh, PF pf) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  // b2 and the ones channel of t, straight from the LDS table into the accumulator (4 broadcast reads; held in
  // registers it would cost 16 VGPRs and 16 copies per tile)
  f32x16 tacc = load_cinit(cl, hh);
  auto conv1_step = [&](f32x16 acc, int et, int s) { return mma16<__bf16>(w.w1[et * C::KS1 + s], xb[s], acc); };
  auto conv1_init = [&](int et) { return C::FOLD_B1 ? zero16() : load_cinit(cl + 32 + et * 32, hh); };
  auto cvt4 = [&](const f32x16& a, int base) {       // regs base..base+7 -> 4 packed dwords (no ReLU yet)
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      __bf16 lo, hi;
      cvt_pair<__bf16>(lo, hi, a[base + j], a[base + j + 1]);
      f[j] = lo;
      f[j + 1] = hi;
    }
    return f;
  };
  auto relu8 = [&](bf16x8 f) {
    s16x8 v = __builtin_bit_cast(s16x8, f);
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(v, z));
  };
  static_assert(C::KS1 == 2, "two conv1 k-steps per e-tile");
  // a wave inside an MFMA chain goes first at the issue arbiter: the other wave of the SIMD is then the one that fills the
  // gaps with its epilogue VALU work and not the other way round (measured: two-role pipeline kernel at batch 512 89.4 -> 84.5 us,
  // the per-tile kernel at batch 32 9.40 -> 9.31 us; results unchanged)
  SR_RS_PRIO(SR_RS_PRIO_A);
  f32x16 h = conv1_init(0);
  h = conv1_step(h, 0, 0);
  h = conv1_step(h, 0, 1);
  bf16x8 f1prev = {};
#pragma unroll
  for (int et = 0; et < C::NET; ++et) {
    const bool more = et + 1 < C::NET;
    f32x16 hn = h;
    pf(et);                                                                                     // PFN LDS reads for a later phase
    if (more) hn = conv1_step(conv1_init(et + 1), et + 1, 0);                                   // MFMA
    bf16x8 f0 = cvt4(h, 0);                                                                     // 4 VALU
    if (more) hn = conv1_step(hn, et + 1, 1);                                                   // MFMA
    f0 = relu8(f0);                                                                             // 4 VALU
    if (et > 0 && 2 * (et - 1) + 1 < C::KS2) tacc = mma16<__bf16>(w.w2[2 * et - 1], f1prev, tacc);   // MFMA
    bf16x8 f1 = cvt4(h, 8);                                                                     // 4 VALU
    if (2 * et < C::KS2) tacc = mma16<__bf16>(w.w2[2 * et], f0, tacc);                          // MFMA
    f1prev = relu8(f1);                                                                         // 4 VALU
    h = hn;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);    // 4 VALU
      if (k < PFN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 LDS read (prefetch)
    }
  }
  if (2 * (C::NET - 1) + 1 < C::KS2) tacc = mma16<__bf16>(w.w2[2 * C::NET - 1], f1prev, tacc);
  SR_RS_PRIO(0);
  return tacc;
}

// per-lane description of one pixel of a phase-A tile, computed BEFORE the MFMA stream starts
struct RwPix {
  int hp;        // row of the x / t images
  int tso;       // element offset in the saved-t tile, -1 = not a core pixel
  bool valid;    // inside the image (t is zero elsewhere)
};
template <typename C, int RW, int NP, int HALO>
SR_DEV RwPix rw_pix_a(int hp, int H, int W, int ty0, int tx0) {
  RwPix p;
  p.hp = hp;
  const int hy = hp / RW, hx = hp - hy * RW;
  const int Y = ty0 - HALO + hy, X = tx0 - HALO + hx;
  p.valid = hp < NP && Y >= 0 && Y < H && X >= 0 && X < W;
  const bool core = hp < NP && hy >= HALO && hy < HALO + C::TH && hx >= HALO && hx < HALO + C::TW;
  p.tso = core ? ((hy - HALO) * C::TW + hx - HALO) * C::LP : -1;
  return p;
}
template <typename C, bool SAVE_T>
SR_DEV void rw_store_t(const f32x16& tacc, const RwPix& p, __bf16* TT, __bf16* tsave_tile, int hh) {
  bf16x4 v[C::CPT];
#pragma unroll
  for (int g = 0; g < C::CPT; ++g) {
    v[g] = acc_group<__bf16>(tacc, g);
    if (!p.valid) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[g][j] = (__bf16)0.f;
    }
  }
  // LDS rows hold TD channels (dense-K operand of the 3x3 conv): the L real ones; at 32 units also the ones channel and one zero (26 -> 28:
  // both meet zero weights); whatever lies behind stays out
#pragma unroll
  for (int g = 0; g < C::CPT; ++g) {
    if (g * 8 + 8 <= C::TD) *reinterpret_cast<bf16x4*>(TT + p.hp * C::TD + g * 8 + hh * 4) = v[g];
    else if (g * 8 + 4 <= C::TD) { if (hh == 0) *reinterpret_cast<bf16x4*>(TT + p.hp * C::TD + g * 8) = v[g]; }
  }
  if constexpr (SAVE_T) {
    if (p.tso >= 0) {
#pragma unroll
      for (int g = 0; g < C::CPT; ++g) stream_store(reinterpret_cast<bf16x4*>(tsave_tile + p.tso + g * 8 + hh * 4), v[g]);
    }
  }
}

// x fragments of one pixel row for conv1: chunk q = 2 s + hh of the F real channels, or the ones chunk behind them
template <typename C, int KXL>
SR_DEV void rw_x_frags(bf16x8 (&xb)[C::KS1], const __bf16* Xin, const __bf16* ones, int hp, int hh) {
#pragma unroll
  for (int s = 0; s < C::KS1; ++s) {
    const int q = 2 * s + hh;
    const __bf16* src = Xin + hp * KXL + q * 8;
    if (C::FOLD_B1 && 2 * s + 1 >= C::FC) src = q < C::FC ? src : ones;     // (per-lane select only where a half can leave the row)
    xb[s] = *reinterpret_cast<const bf16x8*>(src);
  }
}

// phase A of one block over a region (RW wide, NP pixels, HALO around the tile): t -> TT.  Every wave takes tile
// `wave` and, if there is one, tile `wave + NW`.  `prefetch()` is called once the wave's own operand reads are
// issued: the next phase's weights travel LDS -> registers underneath this phase's MFMAs.
template <typename C, int KXL, int RW, int NP, int HALO, int NW, bool SAVE_T, typename PF, typename BS>
SR_DEV void rw_phase_a(const __bf16* Xin, const __bf16* ones, __bf16* TT, const RwA<C>& w, const float* cl, __bf16* tsave_tile,
                       int H, int W, int ty0, int tx0, int wave, int lane, PF prefetch, BS before_stores) {
  constexpr int NT = (NP + 31) / 32;
  const int r = lane & 31, hh = lane >> 5;
  if (wave >= NT) {                                    // wave-uniform
    prefetch(0, 64);
    before_stores();
    return;
  }
  RwPix p = rw_pix_a<C, RW, NP, HALO>(wave * 32 + r, H, W, ty0, tx0);
  bf16x8 xb[C::KS1];
  rw_x_frags<C, KXL>(xb, Xin, ones, p.hp, hh);
  SR_STAMP_AT(11);
  // first tile: the next phase's weights travel LDS -> registers in its MFMA gaps, PFN reads per e-tile (all at
  // once they would queue ~500 LDS cycles in front of this phase's own operands)
  constexpr int PFN = 4;
  {
    const f32x16 t = rw_t_tile<C, PFN>(xb, w, cl, hh, [&](int et) { prefetch(et * PFN, et * PFN + PFN); });
    SR_STAMP_AT(12);
    const RwPix pc = p;
    if (wave + NW < NT) {                              // the next tile's operands land while this one is converted and stored
      p = rw_pix_a<C, RW, NP, HALO>((wave + NW) * 32 + r, H, W, ty0, tx0);
      rw_x_frags<C, KXL>(xb, Xin, ones, p.hp, hh);
    }
    prefetch(C::NET * PFN, 64);                        // (whatever is left)
    before_stores();                                   // (counted-wait hook: no VMEM store of this wave has been issued yet)
    rw_store_t<C, SAVE_T>(t, pc, TT, tsave_tile, hh);
    SR_STAMP_AT(13);
  }
#pragma unroll 1
  for (int tile = wave + NW; tile < NT; tile += NW) {
    const f32x16 t = rw_t_tile<C>(xb, w, cl, hh, [](int) {});
    const RwPix pc = p;
    if (tile + NW < NT) {
      p = rw_pix_a<C, RW, NP, HALO>((tile + NW) * 32 + r, H, W, ty0, tx0);
      rw_x_frags<C, KXL>(xb, Xin, ones, p.hp, hh);
    }
    rw_store_t<C, SAVE_T>(t, pc, TT, tsave_tile, hh);
  }
}

// Pixel operands of the DENSE-K 3x3 conv.  The three taps of window row ky are 3 L contiguous elements of the t image (rows of
// L real channels: 40 bytes at L = 20, 32 consecutive pixels hit 32 different bank pairs with 8-byte reads): 15 four-channel
// chunks, and k-step 4 ky + q gives lane half hh the chunks 8 hh + 2 q and 8 hh + 2 q + 1, i.e. 16 contiguous bytes at
// (window row) + 64 hh + 16 q.  With the lane-half term folded into the base every read is `base + compile-time offset`:
// one address computation per tile.  Chunk 15 of a row (half 1, q = 3) is the first chunk of the next pixel: zero weights in
// W3D; in the last row that slot reads a "ones" chunk instead, whose weight is b3.  Against the 8-channel chunks of
// wdsr_block.h (t rows padded to LP = 24 with the ones channel, residual as identity chunks): 12 k-steps instead of 15.
template <typename C, int RWI> struct RwBAddrD {
  static constexpr int TD = C::TD, KS = C::KS3D, KPR = C::KPR;
  static_assert(C::DENSE3, "dense-K layout");
  typedef __attribute__((address_space(3))) const volatile bf16x4* lds_chunk_p;
  // VOLATILE 8-byte LDS reads: hipcc otherwise merges two of them -- the two of one k-step, or the first chunks of two k-steps --
  // into one ds_read2_b64, which takes 8 LDS cycles where two ds_read_b64 take 4 (at one read pair per MFMA and four SIMDs
  // that is the whole LDS issue rate).
  lds_chunk_p b0;         // first chunk of this lane half in window row 0
  lds_chunk_p last;       // second chunk of k-step QONE of the last row: half 0 its own chunk, half 1 the ones chunk
  SR_DEV void init(const __bf16* TT, const __bf16* ones, int hy, int hx, int hh) {
    const __bf16* g0 = TT + (hy * RWI + hx) * TD + hh * (C::HALF * 4);
    b0 = (lds_chunk_p)(g0);
    last = (lds_chunk_p)(hh ? ones : g0 + 2 * RWI * TD + C::QONE * 8 + 4);
  }
  SR_DEV bf16x8 frag(int s) const {
    const int off = ((s / KPR) * RWI * TD + (s % KPR) * 8) / 4;  // in 4-element chunks
    const bf16x4 lo = b0[off];
    const bf16x4 hi = s == 2 * KPR + C::QONE ? *last : b0[off + 1];
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
};
// the residual as the accumulator's initial value: rows f = 8 g + 4 hh + (0..3) of this pixel's x row (exact: bf16 -> fp32)
template <typename C> SR_DEV f32x16 rw_resid_init(const __bf16* xrow, int hh) {
  f32x16 acc;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g < C::FC) {
      const bf16x4 v = *reinterpret_cast<const bf16x4*>(xrow + g * 8 + hh * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[4 * g + k] = (float)v[k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[4 * g + k] = 0.f;
    }
  }
  return acc;
}

struct RwPixB {
  int hy, hx;
  int xno;          // element offset of this pixel's row in the next block's input image
  unsigned go;      // byte offset in the global output image, out of range (dropped by the bounds check) = do not store
};
template <typename C, int KXL, int RWO, int NPO, int HALOO>
SR_DEV RwPixB rw_pix_b(int tile, int r, int H, int W, int ty0, int tx0) {
  RwPixB p;
  // 32 consecutive pixels of the row-major region per tile (also for the core: consecutive 48-byte t rows are a
  // conflict-free stride for the 16-byte operand reads, 4 x 8 pixel patches were 2-3-way conflicted)
  const int hp = tile * 32 + r;
  const int hpc = hp < NPO ? hp : 0;                  // rows past the region: compute something finite, store it in the slack rows
  p.hy = hpc / RWO;
  p.hx = hpc - p.hy * RWO;
  p.xno = hp * KXL;
  const int Y = ty0 - HALOO + p.hy, X = tx0 - HALOO + p.hx;
  const bool st = hp < NPO && p.hy >= HALOO && p.hy < HALOO + C::TH && p.hx >= HALOO && p.hx < HALOO + C::TW && Y < H && X < W;
  p.go = st ? (unsigned)((Y * W + X) * C::F * 2) : 0xFFFFFF00u;
  return p;
}
// y of one tile: LDS hand-off to the next block (every lane: rows past the region land in the image's slack rows)
// and / or the global store of the core pixels.  The global store is a buffer store whose offset is out of range
// for the lanes that must not store (the bounds check drops them): no branch, so the stores can sit in the middle
// of the next tile's MFMA chain.
template <typename C>
SR_DEV void rw_store_y(const f32x16& oacc, const RwPixB& p, __bf16* Xnext, __amdgpu_buffer_rsrc_t yrs, bool to_global, int hh) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
  bf16x4 v[C::FC];
#pragma unroll
  for (int g = 0; g < C::FC; ++g) v[g] = acc_group<__bf16>(oacc, g);
  if (Xnext) {
#pragma unroll
    for (int g = 0; g < C::FC; ++g) *reinterpret_cast<bf16x4*>(Xnext + p.xno + g * 8 + hh * 4) = v[g];
  }
  if (to_global) {
    const unsigned off = p.go + hh * 8;
#pragma unroll
    for (int g = 0; g < C::FC; ++g) {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, v[g]), yrs, off + g * 16, 0, 2 /* nt */);
    }
  }
}

// One 3x3 chain: (MFMA, reads) in source order, pinned by sched_barrier: the pixel operand of k-step s + AHEAD is
// requested right after the MFMA of k-step s (a 3x3 chain is one dependent accumulation, which issues back to back at
// full rate).  `mid()` runs LAG MFMAs into the chain: the previous tile's stores go there.  `acc` = the residual.
template <typename C, typename A, int AHEAD, int LAG, typename MID, typename PF>
SR_DEV f32x16 rw_b_chain(const A& a, const RwB<C>& w, f32x16 acc, MID mid, PF pf) {
  constexpr int KS = C::KS3D;
  bf16x8 f[KS];
#pragma unroll
  for (int s = 0; s < AHEAD; ++s) f[s] = a.frag(s);
  SR_RS_PRIO(SR_RS_PRIO_B);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    acc = mma16<__bf16>(w.w3[s], f[s], acc);
    if (s + AHEAD < KS) f[s + AHEAD] = a.frag(s + AHEAD);
    pf(s);
    __builtin_amdgcn_sched_barrier(0);
    if (s == LAG) {
      mid();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  SR_RS_PRIO(0);
  return acc;
}

template <typename C, int KXL, int RWO, int NPO, int HALOO, int NW, typename PF>
SR_DEV void rw_phase_b(const __bf16* TT, const __bf16* Xin, const __bf16* ones, __bf16* Xnext, __bf16* yout, const RwB<C>& w, int H,
                       int W, int ty0, int tx0, int wave, int lane, PF prefetch) {
  constexpr int NT = (NPO + 31) / 32, RWI = RWO + 2, AHEAD = 4, LAG = 4;
  typedef RwBAddrD<C, RWI> A;
  const int r = lane & 31, hh = lane >> 5;
  if (wave >= NT) {                                    // wave-uniform
    prefetch(0, 64);
    return;
  }
  SR_STAMP_AT(6);
  const bool to_global = yout != nullptr;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yout, 0, to_global ? H * W * C::F * 2 : 0, 0x00020000);
  RwPixB p = rw_pix_b<C, KXL, RWO, NPO, HALOO>(wave, r, H, W, ty0, tx0);
  A a;
  a.init(TT, ones, p.hy, p.hx, hh);
  SR_STAMP_AT(7);
  // first tile: nothing to store yet; the next phase's weights travel LDS -> registers two reads per MFMA
  f32x16 acc = rw_b_chain<C, A, AHEAD, LAG>(a, w, rw_resid_init<C>(Xin + ((p.hy + 1) * RWI + p.hx + 1) * KXL, hh), [] {},
                                            [&](int s) { prefetch(2 * s, 2 * s + 2); });
  prefetch(2 * C::KS3D, 128);
  SR_STAMP_AT(8);
#pragma unroll 1
  for (int tile = wave + NW; tile < NT; tile += NW) {
    const RwPixB pp = p;
    const f32x16 pacc = acc;
    p = rw_pix_b<C, KXL, RWO, NPO, HALOO>(tile, r, H, W, ty0, tx0);
    a.init(TT, ones, p.hy, p.hx, hh);
    acc = rw_b_chain<C, A, AHEAD, LAG>(a, w, rw_resid_init<C>(Xin + ((p.hy + 1) * RWI + p.hx + 1) * KXL, hh),
                                       [&] { rw_store_y<C>(pacc, pp, Xnext, yrs, to_global, hh); }, [](int) {});
  }
  SR_STAMP_AT(9);
  rw_store_y<C>(acc, p, Xnext, yrs, to_global, hh);
  SR_STAMP_AT(10);
}

// grid = (tiles_y * tiles_x, N), 512 threads.  NBLK = 1: x -> yb (ya, wb, cib, tsb unused).  NBLK = 2: x -> ya (block
// A's output, nullptr = not stored) -> yb.  tsa / tsb (SAVE_T): saved t images, [N][tiles][288][LP].
template <int F, int E, int L, int NBLK, bool SAVE_T>
__global__ __launch_bounds__(512) void wdsr_fwd_rs_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ ya,
                                                          __bf16* __restrict__ yb, const __bf16* __restrict__ wa,
                                                          const __bf16* __restrict__ wb, const float* __restrict__ cia,
                                                          const float* __restrict__ cib, __bf16* __restrict__ tsa,
                                                          __bf16* __restrict__ tsb, int H, int W, int tiles_x) {
  typedef BlockCfg<F, E, L> C;
  typedef RsCfg<F, E, L, NBLK> R;
  constexpr int NTHREADS = R::NTHREADS, NW = R::NWAVES;
  __shared__ __attribute__((aligned(16))) char smem_raw[R::LDS_BYTES];
  __bf16* const X0 = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* const TT = X0 + R::X0_ELEMS;
  __bf16* const X1 = TT + R::TT_ELEMS;
  __bf16* const WL = X1 + R::X1_ELEMS;
  __bf16* const ONES = WL + NBLK * R::W_ELEMS;
  float* const CL = reinterpret_cast<float*>(ONES + R::ONES_ELEMS);
  constexpr int KXL = R::KXL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // provably wave-uniform: scalar branches
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * F;
  const size_t tile_g = (size_t)n * gridDim.x + tile;
  SR_STAMP_DECL;
  SR_STAMP();

  // ---- stage (LDS-DMA from inline asm): x on the halo'd region, the C-init tables, block 0's weights.  A wave's
  // DMA issue blocks once the CU's queue is full, i.e. issuing IS transferring (~55 GB/s per CU): block 1's weights
  // are therefore issued only after the first barrier and land underneath phase A of block 0. ----
  auto stage_pieces = [&](int lo, int hi) {
    const char* zeros = reinterpret_cast<const char*>(g_sr_const_chunks) + 16;
    const int y0 = ty0 - NBLK, x0 = tx0 - NBLK;
    const int lq = lane / C::FC, lc = lane - lq * C::FC;      // this lane's fixed place in every x piece
#pragma unroll 1
    for (int p = lo + ((wave - lo) & (NW - 1)); p < hi; p += NW) {
      if (p < R::P_C) {                                // x: 21 pixels x 3 chunks (+ 1 chunk of the next pixel) per piece
        const int px_ = p * R::PXP + lq;
        const int py = px_ / R::rw(0), pxx = px_ - py * R::rw(0);
        const int Y = y0 + py, X = x0 + pxx;
        const char* src = zeros;
        if (px_ < R::np(0) && Y >= 0 && Y < H && X >= 0 && X < W)
          src = reinterpret_cast<const char*>(x + img + ((size_t)Y * W + X) * C::F + lc * 8);
        dma_piece16(src, lds_addr(X0) + p * (R::PXP * C::FC * 16));
      } else if (p < R::P_W) {                         // C-init tables: 64 floats per piece, zeros past the table
        const int k = p - R::P_C, blk = k / (R::CL_FLOATS / 64), i = (k % (R::CL_FLOATS / 64)) * 64 + lane;
        const float* tab = (NBLK > 1 && blk == 1) ? cib : cia;
        const char* src = i < C::CINIT_FWD ? reinterpret_cast<const char*>(tab + i) : zeros + (lane & 3) * 4;
        dma_piece4(src, lds_addr(CL) + k * 256);
      } else {                                         // weight fragments as they lie
        const int fr = p - R::P_W;
        const __bf16* wsrc = (NBLK > 1 && fr >= R::NFR) ? wb + (size_t)R::src_frag(fr - R::NFR) * 512 : wa + (size_t)R::src_frag(fr) * 512;
        dma_piece16(reinterpret_cast<const char*>(wsrc + lane * 8), lds_addr(WL) + fr * 1024);
      }
    }
  };
  static_assert((NW & (NW - 1)) == 0, "power-of-two wave count");
  stage_pieces(0, R::P_W1);
  if (tid < 8) ONES[tid] = tid == 0 ? (__bf16)1.f : (__bf16)0.f;
  SR_STAMP();
  wait_vmcnt<0>();                                      // each wave waits for its own pieces, the barrier joins them
  __syncthreads();
  SR_STAMP();
  if constexpr (NBLK > 1) stage_pieces(R::P_W1, R::P_END);

  __bf16* const tsa_tile = SAVE_T ? tsa + tile_g * (C::TH * C::TW) * C::LP : nullptr;
  __bf16* const tsb_tile = (SAVE_T && NBLK > 1) ? tsb + tile_g * (C::TH * C::TW) * C::LP : nullptr;
  RwA<C> rwa;
  RwB<C> rwb;
  rwa.load(WL, lane);
  auto pf_a = [&](const __bf16* wl, int lo, int hi) {   // fragments [lo, hi) of the next conv1 / conv2 weights
#pragma unroll
    for (int i = 0; i < RwA<C>::N; ++i)
      if (i >= lo && i < hi) rwa.load_one(wl, lane, i);
  };
  auto pf_b = [&](const __bf16* wl, int lo, int hi) {
#pragma unroll
    for (int i = 0; i < RwB<C>::N; ++i)
      if (i >= lo && i < hi) rwb.load_one(wl, lane, i);
  };
  // ---- block 0 ----
  rw_phase_a<C, KXL, R::rw(0), R::np(0), NBLK, NW, SAVE_T>(X0, ONES, TT, rwa, CL, tsa_tile, H, W, ty0, tx0, wave, lane,
                                                             [&](int lo, int hi) { pf_b(WL, lo, hi); }, [] { wait_vmcnt<0>(); });
  SR_STAMP();
  __syncthreads();
  SR_STAMP();
  if constexpr (NBLK == 1) {
    rw_phase_b<C, KXL, C::TW, C::TH * C::TW, 0, NW>(TT, X0, ONES, nullptr, yb + img, rwb, H, W, ty0, tx0, wave, lane, [](int, int) {});
  } else {
    rw_phase_b<C, KXL, R::rw(1), R::np(1), NBLK - 1, NW>(TT, X0, ONES, X1, ya ? ya + img : nullptr, rwb, H, W, ty0, tx0, wave, lane,
                                                         [&](int lo, int hi) { pf_a(WL + R::W_ELEMS, lo, hi); });
    SR_STAMP();
    __syncthreads();
    SR_STAMP();
    // ---- block 1 ----
    rw_phase_a<C, KXL, R::rw(1), R::np(1), NBLK - 1, NW, SAVE_T>(X1, ONES, TT, rwa, CL + R::CL_FLOATS, tsb_tile, H, W, ty0, tx0, wave,
                                                                 lane, [&](int lo, int hi) { pf_b(WL + R::W_ELEMS, lo, hi); }, [] {});
    SR_STAMP();
    __syncthreads();
    SR_STAMP();
    rw_phase_b<C, KXL, C::TW, C::TH * C::TW, 0, NW>(TT, X1, ONES, nullptr, yb + img, rwb, H, W, ty0, tx0, wave, lane, [](int, int) {});
  }
  SR_STAMP();
}

// =============================================================================================
// SIXTEEN waves, weights read from LDS at use (round 3: the route of the 32-unit network; at 24 units an experiment, variant builds).  With one tile per CU a phase has 9-14
// pixel tiles: eight waves take them in two rounds, each a dependent LDS -> MFMA -> convert -> MFMA -> store chain, and a wave
// that runs one tile uses each weight fragment ONCE -- loading it into a register first moves exactly the bytes a read at
// use moves.  Here every pixel tile of a phase has its own wave (four per SIMD, 128 VGPRs), the weight fragment of an MFMA is
// read from LDS one e-tile (conv1 / conv2) or four k-steps (3x3) ahead of it, and nothing is prefetched across phases.
// Same products in the same order: bit-identical to wdsr_fwd_rs_kernel.
// =============================================================================================
template <typename C>
SR_DEV f32x16 rw_t_tile_lds(const bf16x8 (&xb)[C::KS1], const __bf16* wl, int lane, const float* cl, int hh) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  static_assert(C::KS1 == 2, "two conv1 k-steps per e-tile");
  RwA<C> w;                                            // a bag of values: each fragment lives from its read to its MFMA
  constexpr int W2 = C::NET * C::KS1;                  // index of w2[0] in load_one's numbering
  auto need = [&](int et) {                            // the fragments iteration `et` multiplies with (et = -1: the prologue)
    if (et < 0) { w.load_one(wl, lane, 0); w.load_one(wl, lane, 1); return; }
    if (et + 1 < C::NET) { w.load_one(wl, lane, 2 * (et + 1)); w.load_one(wl, lane, 2 * (et + 1) + 1); }
    if (et > 0 && 2 * et - 1 < C::KS2) w.load_one(wl, lane, W2 + 2 * et - 1);
    if (2 * et < C::KS2) w.load_one(wl, lane, W2 + 2 * et);
  };
  f32x16 tacc = load_cinit(cl, hh);
  auto cvt4 = [&](const f32x16& a, int base) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      __bf16 lo, hi;
      cvt_pair<__bf16>(lo, hi, a[base + j], a[base + j + 1]);
      f[j] = lo;
      f[j + 1] = hi;
    }
    return f;
  };
  auto relu8 = [&](bf16x8 f) {
    s16x8 v = __builtin_bit_cast(s16x8, f);
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(v, z));
  };
  need(-1);
  need(0);
  auto conv1_init = [&](int et) { return C::FOLD_B1 ? zero16() : load_cinit(cl + 32 + et * 32, hh); };   // (b1: ones channel or table)
  f32x16 h = conv1_init(0);
  h = mma16<__bf16>(w.w1[0], xb[0], h);
  h = mma16<__bf16>(w.w1[1], xb[1], h);
  bf16x8 f1prev = {};
#pragma unroll
  for (int et = 0; et < C::NET; ++et) {
    const bool more = et + 1 < C::NET;
    f32x16 hn = h;
    if (more) need(et + 1);                                                                     // <= 4 LDS reads, one iteration ahead
    else if (2 * C::NET - 1 < C::KS2) w.load_one(wl, lane, W2 + 2 * C::NET - 1);
    if (more) hn = mma16<__bf16>(w.w1[2 * (et + 1)], xb[0], conv1_init(et + 1));                // MFMA
    bf16x8 f0 = cvt4(h, 0);                                                                     // 4 VALU
    if (more) hn = mma16<__bf16>(w.w1[2 * (et + 1) + 1], xb[1], hn);                            // MFMA
    f0 = relu8(f0);                                                                             // 4 VALU
    if (et > 0 && 2 * (et - 1) + 1 < C::KS2) tacc = mma16<__bf16>(w.w2[2 * et - 1], f1prev, tacc);   // MFMA
    bf16x8 f1 = cvt4(h, 8);                                                                     // 4 VALU
    if (2 * et < C::KS2) tacc = mma16<__bf16>(w.w2[2 * et], f0, tacc);                          // MFMA
    f1prev = relu8(f1);                                                                         // 4 VALU
    h = hn;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);    // 4 VALU
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // 1 LDS read
    }
  }
  if (2 * (C::NET - 1) + 1 < C::KS2) tacc = mma16<__bf16>(w.w2[2 * C::NET - 1], f1prev, tacc);
  return tacc;
}

template <typename C, typename A, int AHEAD>
SR_DEV f32x16 rw_b_chain_lds(const A& a, const __bf16* wl, int lane, f32x16 acc) {
  constexpr int KS = C::KS3D;
  RwB<C> w;
  bf16x8 f[KS];
#pragma unroll
  for (int s = 0; s < AHEAD; ++s) {
    w.load_one(wl, lane, s);
    f[s] = a.frag(s);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    acc = mma16<__bf16>(w.w3[s], f[s], acc);
    if (s + AHEAD < KS) {
      w.load_one(wl, lane, s + AHEAD);
      f[s + AHEAD] = a.frag(s + AHEAD);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  return acc;
}

template <int F, int E, int L, int NBLK, bool SAVE_T>
__global__ __launch_bounds__(1024) void wdsr_fwd_rs16_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ ya,
                                                             __bf16* __restrict__ yb, const __bf16* __restrict__ wa,
                                                             const __bf16* __restrict__ wb, const float* __restrict__ cia,
                                                             const float* __restrict__ cib, __bf16* __restrict__ tsa,
                                                             __bf16* __restrict__ tsb, int H, int W, int tiles_x) {
  typedef BlockCfg<F, E, L> C;
  typedef RsCfg<F, E, L, NBLK> R;
  constexpr int NW = 16;
  __shared__ __attribute__((aligned(16))) char smem_raw[R::LDS_BYTES];
  __bf16* const X0 = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* const TT = X0 + R::X0_ELEMS;
  __bf16* const X1 = TT + R::TT_ELEMS;
  __bf16* const WL = X1 + R::X1_ELEMS;
  __bf16* const ONES = WL + R::W_REGIONS * R::W_ELEMS;
  __bf16* const WL1 = R::REUSE_W ? WL : WL + R::W_ELEMS;             // where block 1's fragments go
  float* const CL = reinterpret_cast<float*>(ONES + R::ONES_ELEMS);
  constexpr int KXL = R::KXL;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * F;
  const size_t tile_g = (size_t)n * gridDim.x + tile;

  auto stage_pieces = [&](int lo, int hi) {              // (as in wdsr_fwd_rs_kernel)
    const char* zeros = reinterpret_cast<const char*>(g_sr_const_chunks) + 16;
    const int y0 = ty0 - NBLK, x0 = tx0 - NBLK;
    const int lq = lane / C::FC, lc = lane - lq * C::FC;
#pragma unroll 1
    for (int p = lo + ((wave - lo) & (NW - 1)); p < hi; p += NW) {
      if (p < R::P_C) {
        const int px_ = p * R::PXP + lq;
        const int py = px_ / R::rw(0), pxx = px_ - py * R::rw(0);
        const int Y = y0 + py, X = x0 + pxx;
        const char* src = zeros;
        if (px_ < R::np(0) && Y >= 0 && Y < H && X >= 0 && X < W)
          src = reinterpret_cast<const char*>(x + img + ((size_t)Y * W + X) * C::F + lc * 8);
        dma_piece16(src, lds_addr(X0) + p * (R::PXP * C::FC * 16));
      } else if (p < R::P_W) {
        const int k = p - R::P_C, blk = k / (R::CL_FLOATS / 64), i = (k % (R::CL_FLOATS / 64)) * 64 + lane;
        const float* tab = (NBLK > 1 && blk == 1) ? cib : cia;
        const char* src = i < C::CINIT_FWD ? reinterpret_cast<const char*>(tab + i) : zeros + (lane & 3) * 4;
        dma_piece4(src, lds_addr(CL) + k * 256);
      } else {
        const int fr = p - R::P_W;
        const __bf16* wsrc = (NBLK > 1 && fr >= R::NFR) ? wb + (size_t)R::src_frag(fr - R::NFR) * 512 : wa + (size_t)R::src_frag(fr) * 512;
        dma_piece16(reinterpret_cast<const char*>(wsrc + lane * 8), lds_addr(WL) + ((R::REUSE_W && fr >= R::NFR) ? fr - R::NFR : fr) * 1024);
      }
    }
  };
  stage_pieces(0, R::P_W1);
  if (tid < 8) ONES[tid] = tid == 0 ? (__bf16)1.f : (__bf16)0.f;
  wait_vmcnt<0>();
  __syncthreads();
  if constexpr (NBLK > 1 && !R::REUSE_W) stage_pieces(R::P_W1, R::P_END);

  __bf16* const tsa_tile = SAVE_T ? tsa + tile_g * (C::TH * C::TW) * C::LP : nullptr;
  __bf16* const tsb_tile = (SAVE_T && NBLK > 1) ? tsb + tile_g * (C::TH * C::TW) * C::LP : nullptr;

  auto phase_a = [&](auto rw_c, auto np_c, auto halo_c, const __bf16* Xin, const __bf16* wl, const float* cl, __bf16* ts_tile, bool join_dma) {
    constexpr int RW = decltype(rw_c)::value, NP = decltype(np_c)::value, HALO = decltype(halo_c)::value, NT = (NP + 31) / 32;
    bool waited = false;
#pragma unroll 1
    for (int t_ = wave; t_ < NT; t_ += NW) {
      const RwPix p = rw_pix_a<C, RW, NP, HALO>(t_ * 32 + r, H, W, ty0, tx0);
      bf16x8 xb[C::KS1];
      rw_x_frags<C, KXL>(xb, Xin, ONES, p.hp, hh);
      const f32x16 t = rw_t_tile_lds<C>(xb, wl, lane, cl, hh);
      if (join_dma && !waited) { wait_vmcnt<0>(); waited = true; }     // block 1's weight pieces (before this wave's first global store)
      rw_store_t<C, SAVE_T>(t, p, TT, ts_tile, hh);
    }
    if (join_dma && !waited) wait_vmcnt<0>();
  };
  auto phase_b = [&](auto rwo_c, auto npo_c, auto halo_c, const __bf16* Xin, __bf16* Xnext, __bf16* yout, const __bf16* wl) {
    constexpr int RWO = decltype(rwo_c)::value, NPO = decltype(npo_c)::value, HALOO = decltype(halo_c)::value, NT = (NPO + 31) / 32;
    constexpr int RWI = RWO + 2;
    typedef RwBAddrD<C, RWI> A;
    const bool to_global = yout != nullptr;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yout, 0, to_global ? H * W * C::F * 2 : 0, 0x00020000);
#pragma unroll 1
    for (int t_ = wave; t_ < NT; t_ += NW) {
      const RwPixB p = rw_pix_b<C, KXL, RWO, NPO, HALOO>(t_, r, H, W, ty0, tx0);
      A a;
      a.init(TT, ONES, p.hy, p.hx, hh);
      const f32x16 acc = rw_b_chain_lds<C, A, 4>(a, wl, lane, rw_resid_init<C>(Xin + ((p.hy + 1) * RWI + p.hx + 1) * KXL, hh));
      rw_store_y<C>(acc, p, Xnext, yrs, to_global, hh);
    }
  };
  typedef std::integral_constant<int, R::rw(0)> RW0;
  typedef std::integral_constant<int, R::np(0)> NP0;
  typedef std::integral_constant<int, NBLK> H0;
  phase_a(RW0{}, NP0{}, H0{}, X0, WL, CL, tsa_tile, NBLK > 1 && !R::REUSE_W);
  __syncthreads();
  if constexpr (NBLK == 1) {
    phase_b(std::integral_constant<int, C::TW>{}, std::integral_constant<int, C::TH * C::TW>{}, std::integral_constant<int, 0>{}, X0, nullptr,
            yb + img, WL);
  } else {
    typedef std::integral_constant<int, R::rw(1)> RW1;
    typedef std::integral_constant<int, R::np(1)> NP1;
    typedef std::integral_constant<int, NBLK - 1> H1;
    // (one weight region: block 1's conv1 / conv2 fragments take over block 0's while its 3x3 runs, block 1's W3D takes over block
    // 0's while block 1's conv1 / conv2 run; each is waited for in front of the barrier that precedes its first use)
    if constexpr (R::REUSE_W) stage_pieces(R::P_W1, R::P_W1 + R::NFR_A);
    phase_b(RW1{}, NP1{}, H1{}, X0, X1, ya ? ya + img : nullptr, WL);
    if constexpr (R::REUSE_W) wait_vmcnt<0>();
    __syncthreads();
    if constexpr (R::REUSE_W) stage_pieces(R::P_W1 + R::NFR_A, R::P_END);
    phase_a(RW1{}, NP1{}, H1{}, X1, WL1, CL + R::CL_FLOATS, tsb_tile, false);
    if constexpr (R::REUSE_W) wait_vmcnt<0>();
    __syncthreads();
    phase_b(std::integral_constant<int, C::TW>{}, std::integral_constant<int, C::TH * C::TW>{}, std::integral_constant<int, 0>{}, X1, nullptr,
            yb + img, WL1);
  }
}

// =============================================================================================
// PERSISTENT form for grids of many tiles per CU (batch >> 32): one workgroup per CU walks the tiles
// t = blockIdx.x, += gridDim.x.  Both blocks' weights and the C-init tables are staged ONCE per workgroup and stay in
// LDS; the next tile's x region lands by LDS-DMA in a second buffer while the current tile computes; block 0's conv1 /
// conv2 weights for the next tile are prefetched LDS -> registers under the current tile's last phase.  What a tile pays
// is its four phases and five barriers: no dispatch ramp, no staging in front of the first MFMA.  (The two-block variant
// that also saves the t images does not fit 256 registers with the loop-carried state: training at such batch sizes keeps
// the per-tile launches.)  Same phases, same
// results as wdsr_fwd_rs_kernel (bit-identical).
// =============================================================================================
template <int F, int E, int L, int NBLK, bool SAVE_T>
__global__ __launch_bounds__(512) void wdsr_fwd_rs_persist_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ ya,
                                                                  __bf16* __restrict__ yb, const __bf16* __restrict__ wa,
                                                                  const __bf16* __restrict__ wb, const float* __restrict__ cia,
                                                                  const float* __restrict__ cib, __bf16* __restrict__ tsa,
                                                                  __bf16* __restrict__ tsb, int N, int H, int W, int tiles_x,
                                                                  int tiles_per_img) {
  typedef BlockCfg<F, E, L> C;
  typedef RsCfg<F, E, L, NBLK> R;
  constexpr int NW = R::NWAVES;
  // The last DMA piece of an x region runs up to 1 KB past it.  In the per-tile kernel that lands in the not-yet-written t
  // image; here the NEXT tile's x arrives while the current tile's t image is in use, so both x buffers carry their own slack.
  constexpr int X0S_ELEMS = R::X0_ELEMS + 512;
  static_assert(R::LDS_BYTES + (X0S_ELEMS + 512) * 2 <= 160 * 1024, "LDS budget with the second x buffer");
  __shared__ __attribute__((aligned(16))) char smem_raw[R::LDS_BYTES + (X0S_ELEMS + 512) * 2];
  __bf16* const X0a = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* const TT = X0a + X0S_ELEMS;
  __bf16* const X1 = TT + R::TT_ELEMS;
  __bf16* const WL = X1 + R::X1_ELEMS;
  __bf16* const ONES = WL + NBLK * R::W_ELEMS;
  float* const CL = reinterpret_cast<float*>(ONES + R::ONES_ELEMS);
  __bf16* const X0b = reinterpret_cast<__bf16*>(smem_raw + R::LDS_BYTES + 1024);
  constexpr int KXL = R::KXL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = N * tiles_per_img;
  const char* zeros = reinterpret_cast<const char*>(g_sr_const_chunks) + 16;

  auto stage_x = [&](int t, __bf16* X0) {              // the halo'd x region of tile t: 21 pixels x 3 chunks (+ 1 chunk) per piece
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int y0 = (tile / tiles_x) * C::TH - NBLK, x0 = (tile % tiles_x) * C::TW - NBLK;
    const size_t img = (size_t)n * H * W * F;
    const int lq = lane / C::FC, lc = lane - lq * C::FC;
#pragma unroll 1
    for (int p = wave; p < R::NPX; p += NW) {
      const int px_ = p * R::PXP + lq;
      const int py = px_ / R::rw(0), pxx = px_ - py * R::rw(0);
      const int Y = y0 + py, X = x0 + pxx;
      const char* src = zeros;
      if (px_ < R::np(0) && Y >= 0 && Y < H && X >= 0 && X < W)
        src = reinterpret_cast<const char*>(x + img + ((size_t)Y * W + X) * C::F + lc * 8);
      dma_piece16(src, lds_addr(X0) + p * (R::PXP * C::FC * 16));
    }
  };
  auto stage_const = [&]() {                           // C-init tables and every block's weights, once
#pragma unroll 1
    for (int p = R::P_C + wave; p < R::P_END; p += NW) {
      if (p < R::P_W) {
        const int k = p - R::P_C, blk = k / (R::CL_FLOATS / 64), i = (k % (R::CL_FLOATS / 64)) * 64 + lane;
        const float* tab = (NBLK > 1 && blk == 1) ? cib : cia;
        const char* src = i < C::CINIT_FWD ? reinterpret_cast<const char*>(tab + i) : zeros + (lane & 3) * 4;
        dma_piece4(src, lds_addr(CL) + k * 256);
      } else {
        const int fr = p - R::P_W;
        const __bf16* wsrc = (NBLK > 1 && fr >= R::NFR) ? wb + (size_t)R::src_frag(fr - R::NFR) * 512 : wa + (size_t)R::src_frag(fr) * 512;
        dma_piece16(reinterpret_cast<const char*>(wsrc + lane * 8), lds_addr(WL) + fr * 1024);
      }
    }
  };

  if ((int)blockIdx.x < total) stage_x(blockIdx.x, X0a);
  stage_const();
  if (tid < 8) ONES[tid] = tid == 0 ? (__bf16)1.f : (__bf16)0.f;
  RwA<C> rwa;
  RwB<C> rwb;
  auto pf_a = [&](const __bf16* wl, int lo, int hi) {
#pragma unroll
    for (int i = 0; i < RwA<C>::N; ++i)
      if (i >= lo && i < hi) rwa.load_one(wl, lane, i);
  };
  auto pf_b = [&](const __bf16* wl, int lo, int hi) {
#pragma unroll
    for (int i = 0; i < RwB<C>::N; ++i)
      if (i >= lo && i < hi) rwb.load_one(wl, lane, i);
  };
  bool first = true;
  int cur = 0;
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    __bf16* const X0 = cur ? X0b : X0a;
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
    const size_t img = (size_t)n * H * W * F;
    wait_vmcnt<0>();                                   // this tile's x (and, the first time, the constants) landed
    __syncthreads();                                   // ... for every wave; the previous tile's last phase is over
    if (t + (int)gridDim.x < total) stage_x(t + gridDim.x, cur ? X0a : X0b);
    if (first) {
      rwa.load(WL, lane);
      first = false;
    }
    __bf16* const tsa_tile = SAVE_T ? tsa + (size_t)t * (C::TH * C::TW) * C::LP : nullptr;
    __bf16* const tsb_tile = (SAVE_T && NBLK > 1) ? tsb + (size_t)t * (C::TH * C::TW) * C::LP : nullptr;
    const bool more = t + (int)gridDim.x < total;
    // ---- block 0 ----
    rw_phase_a<C, KXL, R::rw(0), R::np(0), NBLK, NW, SAVE_T>(X0, ONES, TT, rwa, CL, tsa_tile, H, W, ty0, tx0, wave, lane,
                                                               [&](int lo, int hi) { pf_b(WL, lo, hi); }, [] {});
    __syncthreads();
    if constexpr (NBLK == 1) {
      rw_phase_b<C, KXL, C::TW, C::TH * C::TW, 0, NW>(TT, X0, ONES, nullptr, yb + img, rwb, H, W, ty0, tx0, wave, lane,
                                                      [&](int lo, int hi) { if (more) pf_a(WL, lo, hi); });
    } else {
      rw_phase_b<C, KXL, R::rw(1), R::np(1), NBLK - 1, NW>(TT, X0, ONES, X1, ya ? ya + img : nullptr, rwb, H, W, ty0, tx0, wave, lane,
                                                           [&](int lo, int hi) { pf_a(WL + R::W_ELEMS, lo, hi); });
      __syncthreads();
      // ---- block 1 ----
      rw_phase_a<C, KXL, R::rw(1), R::np(1), NBLK - 1, NW, SAVE_T>(X1, ONES, TT, rwa, CL + R::CL_FLOATS, tsb_tile, H, W, ty0, tx0, wave,
                                                                   lane, [&](int lo, int hi) { pf_b(WL + R::W_ELEMS, lo, hi); }, [] {});
      __syncthreads();
      rw_phase_b<C, KXL, C::TW, C::TH * C::TW, 0, NW>(TT, X1, ONES, nullptr, yb + img, rwb, H, W, ty0, tx0, wave, lane,
                                                      [&](int lo, int hi) { if (more) pf_a(WL, lo, hi); });
    }
    cur ^= 1;
  }
}

// =============================================================================================
// TWO-ROLE PIPELINE for grids of many tiles per CU (two blocks per launch, inference): the persistent loop above still
// reloads a phase's weights LDS -> registers four times per tile (156 + 123 KB of LDS reads per block) and leaves half of
// its MFMA slots to 32-pixel-tile rounding.  Here waves 0..3 ("A") keep conv1 / conv2 of BOTH blocks in registers (152
// VGPRs), waves 4..7 ("B") both 3x3 convs (120): nothing is ever reloaded, and every SIMD hosts one wave of each role, so
// that A's convert / ReLU work sits under B's MFMAs and the other way round.  The four phases of consecutive tiles are
// skewed so that both roles always have work: round j is
//     step 1:  A0(tile j)      ||  B1(tile j - 2)        barrier
//     step 2:  A1(tile j - 1)  ||  B0(tile j)            barrier
// (A0(j) -> B0(j) -> A1(j) -> B1(j) stay in order across the steps.)  LDS: x double-buffered (the next tile lands by
// LDS-DMA), t of block 0, block 0's output double-buffered (A1 reads tile j - 1 while B0 writes tile j), t of block 1.
// Same phase code, same results as wdsr_fwd_rs_kernel (bit-identical).
// =============================================================================================
template <int F, int E, int L>
__global__ __launch_bounds__(512) void wdsr_fwd_rs_pipe_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ ya,
                                                               __bf16* __restrict__ yb, const __bf16* __restrict__ wa,
                                                               const __bf16* __restrict__ wb, const float* __restrict__ cia,
                                                               const float* __restrict__ cib, int N, int H, int W, int tiles_x,
                                                               int tiles_per_img) {
  typedef BlockCfg<F, E, L> C;
  typedef RsCfg<F, E, L, 2> R;
  constexpr int NR = 4;                                                // waves per role
  constexpr int X0S = R::X0_ELEMS + 512;                               // x region + its DMA slack
  constexpr int TT0_ELEMS = R::npad(0) * R::TD, TT1_ELEMS = R::npad(1) * R::TD, X1_ELEMS = R::npad(1) * R::KXL;
  constexpr int BUF_ELEMS = 2 * X0S + TT0_ELEMS + 2 * X1_ELEMS + TT1_ELEMS;
  constexpr int LDS_BYTES = (BUF_ELEMS + R::ONES_ELEMS) * 2 + 2 * R::CL_FLOATS * 4;
  static_assert((TT0_ELEMS + 2 * X1_ELEMS + TT1_ELEMS) >= 2 * R::W_ELEMS, "the weights are parked behind the x buffers until they are in registers");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char smem_raw[LDS_BYTES];
  __bf16* const X0 = reinterpret_cast<__bf16*>(smem_raw);              // [2][X0S]
  __bf16* const TT0 = X0 + 2 * X0S;
  __bf16* const X1 = TT0 + TT0_ELEMS;                                  // [2][X1_ELEMS]
  __bf16* const TT1 = X1 + 2 * X1_ELEMS;
  __bf16* const ONES = TT1 + TT1_ELEMS;
  float* const CL = reinterpret_cast<float*>(ONES + R::ONES_ELEMS);
  __bf16* const WL = TT0;                                              // prologue only
  constexpr int KXL = R::KXL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = N * tiles_per_img;
  const int K = ((int)blockIdx.x < total) ? (total - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;   // this workgroup's tiles
  const char* zeros = reinterpret_cast<const char*>(g_sr_const_chunks) + 16;

  auto tile_of = [&](int k, int& ty0, int& tx0, size_t& img) {
    const int t = blockIdx.x + k * gridDim.x;
    const int n = t / tiles_per_img, tile = t - n * tiles_per_img;
    ty0 = (tile / tiles_x) * C::TH;
    tx0 = (tile % tiles_x) * C::TW;
    img = (size_t)n * H * W * F;
  };
  auto stage_x = [&](int k, __bf16* dst) {             // the halo'd x region of this workgroup's k-th tile (all 8 waves)
    int ty0, tx0;
    size_t img;
    tile_of(k, ty0, tx0, img);
    const int y0 = ty0 - 2, x0 = tx0 - 2;
    const int lq = lane / C::FC, lc = lane - lq * C::FC;
#pragma unroll 1
    for (int p = wave; p < R::NPX; p += R::NWAVES) {
      const int px_ = p * R::PXP + lq;
      const int py = px_ / R::rw(0), pxx = px_ - py * R::rw(0);
      const int Y = y0 + py, X = x0 + pxx;
      const char* src = zeros;
      if (px_ < R::np(0) && Y >= 0 && Y < H && X >= 0 && X < W)
        src = reinterpret_cast<const char*>(x + img + ((size_t)Y * W + X) * C::F + lc * 8);
      dma_piece16(src, lds_addr(dst) + p * (R::PXP * C::FC * 16));
    }
  };

  // ---- prologue: first x region, C-init tables, both blocks' weights (parked in the t / y buffers); weights -> registers ----
  if (K > 0) stage_x(0, X0);
#pragma unroll 1
  for (int p = R::P_C + wave; p < R::P_END; p += R::NWAVES) {
    if (p < R::P_W) {
      const int k = p - R::P_C, blk = k / (R::CL_FLOATS / 64), i = (k % (R::CL_FLOATS / 64)) * 64 + lane;
      const float* tab = blk == 1 ? cib : cia;
      const char* src = i < C::CINIT_FWD ? reinterpret_cast<const char*>(tab + i) : zeros + (lane & 3) * 4;
      dma_piece4(src, lds_addr(CL) + k * 256);
    } else {
      const int fr = p - R::P_W;
      const __bf16* wsrc = fr >= R::NFR ? wb + (size_t)R::src_frag(fr - R::NFR) * 512 : wa + (size_t)R::src_frag(fr) * 512;
      dma_piece16(reinterpret_cast<const char*>(wsrc + lane * 8), lds_addr(WL) + fr * 1024);
    }
  }
  if (tid < 8) ONES[tid] = tid == 0 ? (__bf16)1.f : (__bf16)0.f;
  wait_vmcnt<0>();
  __syncthreads();

  if (wave < NR) {
    // ================= role A: conv1 -> ReLU -> conv2 of both blocks =================
    RwA<C> w0, w1;
    w0.load(WL, lane);
    w1.load(WL + R::W_ELEMS, lane);
    __syncthreads();                                   // every wave holds its weights: the parking area is free
    for (int j = 0; j < K + 2; ++j) {
      if (j + 1 < K) stage_x(j + 1, X0 + ((j + 1) & 1) * X0S);
      if (j < K) {                                     // step 1: A0(j)
        int ty0, tx0;
        size_t img;
        tile_of(j, ty0, tx0, img);
        rw_phase_a<C, KXL, R::rw(0), R::np(0), 2, NR, false>(X0 + (j & 1) * X0S, ONES, TT0, w0, CL, nullptr, H, W, ty0, tx0, wave, lane,
                                                             [](int, int) {}, [] {});
      }
      __syncthreads();
      if (j >= 1 && j - 1 < K) {                       // step 2: A1(j - 1)
        int ty0, tx0;
        size_t img;
        tile_of(j - 1, ty0, tx0, img);
        rw_phase_a<C, KXL, R::rw(1), R::np(1), 1, NR, false>(X1 + ((j - 1) & 1) * X1_ELEMS, ONES, TT1, w1, CL + R::CL_FLOATS, nullptr, H, W,
                                                             ty0, tx0, wave, lane, [](int, int) {}, [] {});
      }
      wait_vmcnt<0>();                                 // the next tile's x pieces this wave issued
      __syncthreads();
    }
  } else {
    // ================= role B: the 3x3 convs (+ residual) of both blocks =================
    const int bw = wave - NR;
    RwB<C> w0, w1;
    w0.load(WL, lane);
    w1.load(WL + R::W_ELEMS, lane);
    __syncthreads();
    for (int j = 0; j < K + 2; ++j) {
      if (j + 1 < K) stage_x(j + 1, X0 + ((j + 1) & 1) * X0S);
      if (j >= 2 && j - 2 < K) {                       // step 1: B1(j - 2)
        int ty0, tx0;
        size_t img;
        tile_of(j - 2, ty0, tx0, img);
        rw_phase_b<C, KXL, C::TW, C::TH * C::TW, 0, NR>(TT1, X1 + ((j - 2) & 1) * X1_ELEMS, ONES, nullptr, yb + img, w1, H, W, ty0, tx0, bw, lane,
                                                        [](int, int) {});
      }
      __syncthreads();
      if (j < K) {                                     // step 2: B0(j)
        int ty0, tx0;
        size_t img;
        tile_of(j, ty0, tx0, img);
        rw_phase_b<C, KXL, R::rw(1), R::np(1), 1, NR>(TT0, X0 + (j & 1) * X0S, ONES, X1 + (j & 1) * X1_ELEMS, ya ? ya + img : nullptr, w0, H, W,
                                                      ty0, tx0, bw, lane, [](int, int) {});
      }
      wait_vmcnt<0>();
      __syncthreads();
    }
  }
}
