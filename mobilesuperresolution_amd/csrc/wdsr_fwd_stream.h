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
// A workgroup's images follow one another as ONE stream of bands (the 3x3 reads a zero row instead of the ring across an image
// boundary), so the pipeline fills and drains once per workgroup, not per image.
// Rings (16 rows = 4 band slots each): x 36 KB, t0 32 KB, y0 36 KB, t1 32 KB.  The phase bodies (rw_t_tile, rw_b_chain, the
// dense-K operand layout, the stores) are those of wdsr_fwd_rs.h: per pixel the same products in the same order, so the
// results are bit-identical to the tile kernels'.
#pragma once
#include <type_traits>
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
  static constexpr int ONES_ELEMS = 8 + 64;                            // ones chunk | zero chunk | a 64-element zero row
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
  // window of image pixel (y, c): padded columns c .. c + 2 of the ring rows of image rows y - 1 .. y + 1 (row0 = the ring's
  // row count at the image's first row); rows above / below the image read `zrow` (64 zero elements)
  SR_DEV void init(const __bf16* T, const __bf16* ones, const __bf16* zrow, int row0, int y, int c, int H, int hh) {
    const __bf16* g2 = nullptr;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y - 1 + ky;
      const __bf16* g = (yy >= 0 && yy < H) ? T + (((row0 + yy) & 15) * TWP + c) * TD + hh * 32 : zrow + hh * 32;
      b[ky] = (lds_chunk_p)(g);
      g2 = g;
    }
    last = (lds_chunk_p)(hh ? ones : g2 + 3 * 8 + 4);
  }
  SR_DEV bf16x4 lo(int s) const { return b[s / 4][(s % 4) * 2]; }
  SR_DEV bf16x4 hi(int s) const { return s == KS - 1 ? *last : b[s / 4][(s % 4) * 2 + 1]; }
  static SR_DEV bf16x8 join(bf16x4 l, bf16x4 h) {
    bf16x8 f;
    f[0] = l[0]; f[1] = l[1]; f[2] = l[2]; f[3] = l[3];
    f[4] = h[0]; f[5] = h[1]; f[6] = h[2]; f[7] = h[3];
    return f;
  }
  SR_DEV bf16x8 frag(int s) const { return join(lo(s), hi(s)); }
};

// conv1 -> ReLU -> conv2 of one band: Xin = the band's 192 pixel rows (24 channels each), t -> ring rows (y & 15), padded column
// c + 1; SAVE_T: also into the weight-gradient kernels' tile-local image [tile 12 x 24][288][LP] of this image.  Wave gw of the
// group takes pixel tiles gw and gw + 4.
template <typename S, bool SAVE_T, int NWR = 4>
SR_DEV void st_phase_a(const __bf16* Xin, const __bf16* ones, __bf16* Tring, const RwA<typename S::C>& w, const float* cl,
                       __bf16* tsave_img, int row0, int band, int H, int gw, int lane) {
  typedef typename S::C C;
  const int r = lane & 31, hh = lane >> 5;
  bf16x8 xb[C::KS1];
  int tile = gw;
  rw_x_frags<C, S::KXL>(xb, Xin, ones, tile * 32 + r, hh);
#pragma unroll 1
  for (; tile < S::NTB; tile += NWR) {
    const f32x16 t = rw_t_tile<C>(xb, w, cl, hh, [](int) {});
    const int p = tile * 32 + r;
    if (tile + NWR < S::NTB) rw_x_frags<C, S::KXL>(xb, Xin, ones, p + 32 * NWR, hh);      // the next tile's operands land under the stores
    const int row = p / S::W, c = p - row * S::W, y = band * S::BR + row;
    RwPix px;
    px.hp = ((row0 + y) & 15) * S::TWP + c + 1;
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
template <typename S, int NWR = 4>
SR_DEV void st_phase_b(const __bf16* Tring, const __bf16* Xres, const __bf16* ones, const __bf16* zrow, __bf16* Ynext, __bf16* yout,
                       const RwB<typename S::C>& w, int row0, int band, int H, int gw, int lane) {
  typedef typename S::C C;
  typedef StBAddr<C, S::TWP> A;
  const int r = lane & 31, hh = lane >> 5;
  const bool to_global = yout != nullptr;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yout, 0, to_global ? H * S::W * C::F * 2 : 0, 0x00020000);
#pragma unroll 1
  for (int tile = NWR - 1 - gw; tile < S::NTB; tile += NWR) {
    const int p = tile * 32 + r;
    const int row = p / S::W, c = p - row * S::W, y = band * S::BR + row;
    A a;
    a.init(Tring, ones, zrow, row0, y, c, H, hh);
    const f32x16 acc = rw_b_chain<C, A, 4, 4>(a, w, rw_resid_init<C>(Xres + p * S::KXL, hh), [] {}, [](int) {});
    RwPixB pb;
    pb.hy = row;
    pb.hx = c;
    pb.xno = p * S::KXL;
    pb.go = (to_global && y < H) ? (unsigned)((y * S::W + c) * C::F * 2) : 0xFFFFFF00u;
    rw_store_y<C>(acc, pb, Ynext, yrs, to_global, hh);
  }
}

// ---- a wave's jobs of one round as ONE software pipeline ----
// The unpipelined phases above run every pixel tile as address arithmetic -> LDS reads -> wait -> MFMA body -> conversions ->
// stores: of the ~1,000 cycles a 3x3 tile takes a lone wave, the matrix pipe works 384, and two waves per SIMD do not cover
// that for each other (ablations on MI355X: conv1/conv2 tiles alone ran at 56 % of the matrix rate, 3x3 tiles alone at 40 %, and
// the two together took the SUM of the two times).  Here a wave's three tiles of a steady-state round are jobs J0, J1, J2:
//     addresses(J0) reads(J0) | addresses(J1) body(J0){reads(J1)} epilogue(J0) | addresses(J2) body(J1){reads(J2)} epilogue(J1) | body(J2) ...
// the next job's operand reads are issued from hooks inside the current body (a few per MFMA gap), so a body starts with its
// operands in registers and only the first job of a round waits for LDS.
enum { ST_JA = 0, ST_JB = 1 };
template <typename S, bool SAVE_T> struct StPipe {
  typedef typename S::C C;
  typedef StBAddr<C, S::TWP> BA;
  static constexpr int NPRE = 2;                                       // k-steps of a 3x3 job whose operands are read ahead
  struct ACtx { const __bf16* Xin; __bf16* Tring; const float* cl; __bf16* tsave; int row0, band; };
  struct BCtx { const __bf16* Tring; const __bf16* Xres; __bf16* Ynext; __bf16* yout; int row0, band; };
  struct AJob { bf16x8 xb[C::KS1]; int p; };
  typedef __attribute__((address_space(3))) const bf16x4* lds_x4_p;
  struct BJob { BA a; lds_x4_p xrow; bf16x4 xr[C::FC]; bf16x4 fl[NPRE], fh[NPRE]; int p; };
  template <int K> using Job = std::conditional_t<K == ST_JA, AJob, BJob>;
  static constexpr int NREADS_A = C::KS1, NREADS_B = C::FC + 2 * NPRE;
  static_assert(C::FC <= 4 && C::KS1 == 2, "job operand lists");

  const __bf16* ones;
  const __bf16* zrow;
  ACtx ca;
  BCtx cb;
  int H, lane;

  // ---- conv1/conv2 job ----
  SR_DEV void a_addr(AJob& j, int tile) const { j.p = tile * 32 + (lane & 31); }
  SR_DEV void a_read(AJob& j, int i) const {                          // operand read #i (i = conv1 k-step)
    const int hh = lane >> 5, q = 2 * i + hh;
    const __bf16* src = ca.Xin + j.p * S::KXL + q * 8;
    if (C::FOLD_B1 && 2 * i + 1 >= C::FC) src = q < C::FC ? src : ones;
    j.xb[i] = *reinterpret_cast<const bf16x8*>(src);
  }
  template <typename HOOK> SR_DEV f32x16 a_body(const AJob& j, const RwA<C>& w, HOOK hook) const {
    return rw_t_tile<C, 4>(j.xb, w, ca.cl, lane >> 5, hook);
  }
  SR_DEV void a_epi(const f32x16& t, const AJob& j) const {
    const int row = j.p / S::W, c = j.p - row * S::W, y = ca.band * S::BR + row;
    RwPix px;
    px.hp = ((ca.row0 + y) & 15) * S::TWP + c + 1;
    px.valid = y < H;
    px.tso = -1;
    if constexpr (SAVE_T) {
      const int ty = y / C::TH, tx = c / C::TW;
      px.tso = px.valid ? (((ty * (S::W / C::TW) + tx) * (C::TH * C::TW)) + (y - ty * C::TH) * C::TW + (c - tx * C::TW)) * C::LP : -1;
    }
    rw_store_t<C, SAVE_T>(t, px, ca.Tring, ca.tsave, lane >> 5);
  }
  // ---- 3x3 job ----
  SR_DEV void b_addr(BJob& j, int tile) const {
    j.p = tile * 32 + (lane & 31);
    const int row = j.p / S::W, c = j.p - row * S::W;
    j.a.init(cb.Tring, ones, zrow, cb.row0, cb.band * S::BR + row, c, H, lane >> 5);
    j.xrow = (lds_x4_p)(cb.Xres + j.p * S::KXL + (lane >> 5) * 4);
  }
  SR_DEV void b_read(BJob& j, int i) const {                          // operand read #i: FC residual chunks, then lo / hi of k-steps 0 .. NPRE - 1
    if (i < C::FC) j.xr[i] = j.xrow[i * 2];
    else if ((i - C::FC) % 2 == 0) j.fl[(i - C::FC) / 2] = j.a.lo((i - C::FC) / 2);
    else j.fh[(i - C::FC) / 2] = j.a.hi((i - C::FC) / 2);
  }
  template <typename HOOK> SR_DEV f32x16 b_body(const BJob& j, const RwB<C>& w, HOOK hook) const {
    constexpr int KS = C::KS3D, AHEAD = 4;
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[4 * g + k] = g < C::FC ? (float)j.xr[g < C::FC ? g : 0][k] : 0.f;
    }
    bf16x8 f[KS];
#pragma unroll
    for (int s2 = 0; s2 < AHEAD; ++s2) f[s2] = s2 < NPRE ? BA::join(j.fl[s2 < NPRE ? s2 : 0], j.fh[s2 < NPRE ? s2 : 0]) : j.a.frag(s2);
    SR_RS_PRIO(SR_RS_PRIO_B);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
      acc = mma16<__bf16>(w.w3[s2], f[s2], acc);
      if (s2 + AHEAD < KS) f[s2 + AHEAD] = j.a.frag(s2 + AHEAD);
      hook(s2);
      __builtin_amdgcn_sched_barrier(0);
    }
    SR_RS_PRIO(0);
    return acc;
  }
  SR_DEV void b_epi(const f32x16& acc, const BJob& j) const {
    const bool to_global = cb.yout != nullptr;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(cb.yout, 0, to_global ? H * S::W * C::F * 2 : 0, 0x00020000);
    const int row = j.p / S::W, c = j.p - row * S::W, y = cb.band * S::BR + row;
    RwPixB pb;
    pb.hy = 0;
    pb.hx = c;
    pb.xno = j.p * S::KXL;
    pb.go = (to_global && y < H) ? (unsigned)((y * S::W + c) * C::F * 2) : 0xFFFFFF00u;
    rw_store_y<C>(acc, pb, cb.Ynext, yrs, to_global, lane >> 5);
  }
  // ---- the pipeline ----
  template <int K> SR_DEV void addr(Job<K>& j, int tile) const {
    if constexpr (K == ST_JA) a_addr(j, tile); else b_addr(j, tile);
  }
  template <int K> SR_DEV void read(Job<K>& j, int i) const {          // (i: compile-time after unrolling)
    if constexpr (K == ST_JA) { if (i < NREADS_A) a_read(j, i); } else { if (i < NREADS_B) b_read(j, i); }
  }
  // operand reads of the next job issued at hook point `step` of a body of kind KC: a conv1/conv2 body has a hook per e-tile
  // (four LDS reads each fit its MFMA gaps), a 3x3 body one per k-step (its own operand reads end at k-step 7: three each from 8 on)
  // how many of the next job's reads the hooks of a body of kind KC issue: a conv1/conv2 body has no registers to spare for the
  // read-ahead k-steps of a 3x3 job (those are issued behind its epilogue)
  template <int KC, int KN> static constexpr int hooked() { return KN == ST_JA ? NREADS_A : (KC == ST_JA ? 0 : NREADS_B); }
  template <int KC, int KN> SR_DEV void hook_reads(Job<KN>& nj, int step) const {
    const int lo = KC == ST_JA ? 4 * step : (step >= 8 ? 3 * (step - 8) : 0);
    const int hi = KC == ST_JA ? lo + 4 : (step >= 8 ? lo + 3 : 0);
#pragma unroll
    for (int i = 0; i < 12; ++i)
      if (i >= lo && i < hi && i < hooked<KC, KN>()) read<KN>(nj, i);
  }
  template <int KC, int KN> SR_DEV void rest_reads(Job<KN>& nj) const {
#pragma unroll
    for (int i = hooked<KC, KN>(); i < 12; ++i) read<KN>(nj, i);
  }
  template <int K, typename HOOK> SR_DEV f32x16 body(const Job<K>& j, const RwA<C>& wa, const RwB<C>& wb, HOOK hook) const {
    if constexpr (K == ST_JA) return a_body(j, wa, hook); else return b_body(j, wb, hook);
  }
  template <int K> SR_DEV void epi(const f32x16& r, const Job<K>& j) const {
    if constexpr (K == ST_JA) a_epi(r, j); else b_epi(r, j);
  }
  template <int K0, int K1> SR_DEV void run2(int t0, int t1, const RwA<C>& wa, const RwB<C>& wb) const {
    Job<K0> j0;
    addr<K0>(j0, t0);
#pragma unroll
    for (int i = 0; i < 12; ++i) read<K0>(j0, i);
    Job<K1> j1;
    addr<K1>(j1, t1);
    const f32x16 r0 = body<K0>(j0, wa, wb, [&](int st) { hook_reads<K0, K1>(j1, st); });
    epi<K0>(r0, j0);
    rest_reads<K0, K1>(j1);
    const f32x16 r1 = body<K1>(j1, wa, wb, [](int) {});
    epi<K1>(r1, j1);
  }
  template <int K0, int K1, int K2> SR_DEV void run3(int t0, int t1, int t2, const RwA<C>& wa, const RwB<C>& wb) const {
    Job<K0> j0;
    addr<K0>(j0, t0);
#pragma unroll
    for (int i = 0; i < 12; ++i) read<K0>(j0, i);
    Job<K1> j1;
    addr<K1>(j1, t1);
    const f32x16 r0 = body<K0>(j0, wa, wb, [&](int st) { hook_reads<K0, K1>(j1, st); });
    epi<K0>(r0, j0);
    rest_reads<K0, K1>(j1);
    Job<K2> j2;
    addr<K2>(j2, t2);
    const f32x16 r1 = body<K1>(j1, wa, wb, [&](int st) { hook_reads<K1, K2>(j2, st); });
    epi<K1>(r1, j1);
    rest_reads<K1, K2>(j2);
    const f32x16 r2 = body<K2>(j2, wa, wb, [](int) {});
    epi<K2>(r2, j2);
  }
};

// grid = min(N, 256) workgroups of 512 threads, each walking images blockIdx.x, blockIdx.x + gridDim.x, ... as ONE stream of
// bands (the pipeline runs on across image boundaries; weights are staged once); W = 48, H % 4 == 0.
// x -> ya (block 0's output; nullptr = not stored) -> yb.  tsa / tsb (SAVE_T): saved t images [N][tiles][288][LP].
template <int F, int E, int L, bool SAVE_T>
__global__ __launch_bounds__(512) void wdsr_fwd_stream_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ ya,
                                                              __bf16* __restrict__ yb, const __bf16* __restrict__ wa,
                                                              const __bf16* __restrict__ wb, const float* __restrict__ cia,
                                                              const float* __restrict__ cib, __bf16* __restrict__ tsa,
                                                              __bf16* __restrict__ tsb, int N, int H) {
  typedef StreamCfg<F, E, L> S;
  typedef typename S::C C;
  typedef typename S::R R;
  __shared__ __attribute__((aligned(16))) char smem_raw[S::LDS_BYTES];
  __bf16* const XR = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* const T0 = XR + S::X_ELEMS;
  __bf16* const Y0 = T0 + S::T_ELEMS;
  __bf16* const T1 = Y0 + S::X_ELEMS;
  __bf16* const ONES = T1 + S::T_ELEMS;
  __bf16* const ZROW = ONES + 8;
  float* const CL = reinterpret_cast<float*>(ONES + S::ONES_ELEMS);
  __bf16* const PARK = T0;                                             // prologue only: both blocks' weight fragments
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  // both waves of a SIMD (w and w + 4) get the same number of MFMAs per round: group 1 deals its tiles the other way round
  const int gw = grp ? 3 - (wave & 3) : (wave & 3);
  const int NB = H / S::BR;
  const int K = ((int)blockIdx.x < N) ? (N - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;   // this workgroup's images
  const int GB = K * NB;                                               // its bands
  const size_t img_elems = (size_t)H * S::W * F;
  const int tiles_img = ((H + C::TH - 1) / C::TH) * (S::W / C::TW);
  const char* zeros = reinterpret_cast<const char*>(g_sr_const_chunks) + 16;
  auto image_of = [&](int g, int& b) {                  // global band g -> image index, band in the image
    const int k = g / NB;
    b = g - k * NB;
    return (int)blockIdx.x + k * (int)gridDim.x;
  };
  auto stage_x = [&](int g) {                           // a band's rows are ONE contiguous 9 KB run of its image
    int b;
    const int n = image_of(g, b);
    const char* src = reinterpret_cast<const char*>(x + (size_t)n * img_elems) + (size_t)b * (S::BPX * S::KXL * 2);
    const unsigned dst = lds_addr(XR) + (g & 3) * (S::BPX * S::KXL * 2);
#pragma unroll 1
    for (int p = wave; p < S::XPIECES; p += 8) dma_piece16(src + p * 1024 + lane * 16, dst + p * 1024);
  };

  // ---- prologue: C-init tables, both blocks' weights (parked in the rings), x band 0; weights -> registers; rings zeroed ----
  SR_STAMP_AT(8);
  if (GB > 0) stage_x(0);
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
  if (tid < S::ONES_ELEMS) ONES[tid] = tid == 0 ? (__bf16)1.f : (__bf16)0.f;
  wait_vmcnt<0>();
  __syncthreads();
  RwA<C> rwa;
  RwB<C> rwb;
  rwa.load(PARK + grp * R::W_ELEMS, lane);
  rwb.load(PARK + grp * R::W_ELEMS, lane);
  __syncthreads();                                     // every wave holds its block's weights: the parking area is free
  {
    // the t rings' padding columns (and the slack behind the last row) stay zero for good: phase A writes columns 1 .. 48 only
    const u32x4 z = {0u, 0u, 0u, 0u};
    u32x4* t0 = reinterpret_cast<u32x4*>(T0);
    u32x4* t1 = reinterpret_cast<u32x4*>(T1);
    for (int i = tid; i < S::T_ELEMS * 2 / 16; i += 512) { t0[i] = z; t1[i] = z; }
  }
  if constexpr (SAVE_T) {
    // rows of the last 12-row tiles below the image: the weight-gradient kernels read whole tiles, and the tile kernels write
    // zeros there
    const int rows_pad = ((H + C::TH - 1) / C::TH) * C::TH - H;
    static_assert((C::LP * 2) % 16 == 0, "a saved t pixel is whole 16-byte pieces");
    constexpr int PPX = C::LP * 2 / 16;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int k = 0; k < K; ++k) {
      __bf16* const ts_img = (grp ? tsb : tsa) + (size_t)((int)blockIdx.x + k * (int)gridDim.x) * tiles_img * (C::TH * C::TW) * C::LP;
      for (int i = (wave & 3) * 64 + lane; i < rows_pad * S::W * PPX; i += 256) {
        const int px = i / PPX, q = i - px * PPX;
        const int y = H + px / S::W, c = px % S::W;
        const int ty = y / C::TH, tx = c / C::TW;
        const int off = (((ty * (S::W / C::TW) + tx) * (C::TH * C::TW)) + (y - ty * C::TH) * C::TW + (c - tx * C::TW)) * C::LP + q * 8;
        stream_store(reinterpret_cast<u32x4*>(ts_img + off), z);
      }
    }
  }
  __syncthreads();
  SR_STAMP_AT(9);

  const float* const cl = CL + grp * R::CL_FLOATS;
  // global stores a wave issues per phase of a round (3 per pixel tile): the x pieces of the next band are issued BEFORE them, so a
  // counted wait retires the pieces and leaves this round's stores in flight (vmcnt retires in issue order)
  const int st_a = SAVE_T ? 3 * (gw < 2 ? 2 : 1) : 0;
  const int st_b = 3 * (gw < 2 ? 1 : 2);
  constexpr int BAND_ELEMS = S::BPX * S::KXL;
  // phase lags in bands: block 0's 3x3 two bands behind its conv1/conv2 (it reads the t band below), block 1's conv1/conv2 one
  // more (block 0's output band must be complete), its 3x3 two more
#pragma unroll 1
  for (int i = 0; i < GB + 5; ++i) {
    if (i == 8) SR_STAMP_AT(0);
    if (i + 1 < GB) stage_x(i + 1);
    if (i == 8) SR_STAMP_AT(1);
    int nst = 0;
    {
      // bands of this round: conv1/conv2 on band ga, the 3x3 conv on band gb (of this wave's block)
      const int ga = grp ? i - 3 : i, gb = grp ? i - 5 : i - 2;
      const bool has_a = ga >= 0 && ga < GB, has_b = gb >= 0 && gb < GB;
      __bf16* const ts = grp ? tsb : tsa;
      __bf16* const Tr = grp ? T1 : T0;
      int ba = 0, bb = 0;
      const int na = has_a ? image_of(ga, ba) : 0, nb = has_b ? image_of(gb, bb) : 0;
      const __bf16* const a_in = (grp ? Y0 : XR) + (ga & 3) * BAND_ELEMS;                // block input rows of band ga
      const __bf16* const b_res = (grp ? Y0 : XR) + (gb & 3) * BAND_ELEMS;               // block input rows of band gb (residual)
      __bf16* const a_ts = SAVE_T ? ts + (size_t)na * tiles_img * (C::TH * C::TW) * C::LP : nullptr;
      __bf16* const b_next = grp ? nullptr : Y0 + (gb & 3) * BAND_ELEMS;
      __bf16* const b_out = grp ? yb + (size_t)nb * img_elems : (ya ? ya + (size_t)nb * img_elems : nullptr);
      // (with the saved t images the pipelined form is one register short of 256: those launches run the plain phases)
      if (!SAVE_T && has_a && has_b) {
        StPipe<S, SAVE_T> pp;
        pp.ones = ONES; pp.zrow = ZROW; pp.H = H; pp.lane = lane;
        pp.ca = {a_in, Tr, cl, a_ts, (ga - ba) * S::BR, ba};
        pp.cb = {Tr, b_res, b_next, b_out, (gb - bb) * S::BR, bb};
        if (grp == 0) {
          if (gw < 2) pp.template run3<ST_JA, ST_JA, ST_JB>(gw, gw + 4, 3 - gw, rwa, rwb);
          else pp.template run3<ST_JA, ST_JB, ST_JB>(gw, 3 - gw, 7 - gw, rwa, rwb);
        } else {
          if (gw < 2) pp.template run3<ST_JB, ST_JA, ST_JA>(3 - gw, gw, gw + 4, rwa, rwb);
          else pp.template run3<ST_JB, ST_JB, ST_JA>(3 - gw, 7 - gw, gw, rwa, rwb);
        }
      } else {
        if (grp == 0 && has_a) st_phase_a<S, SAVE_T>(a_in, ONES, Tr, rwa, cl, a_ts, (ga - ba) * S::BR, ba, H, gw, lane);
        if (has_b) st_phase_b<S>(Tr, b_res, ONES, ZROW, b_next, b_out, rwb, (gb - bb) * S::BR, bb, H, gw, lane);
        if (grp == 1 && has_a) st_phase_a<S, SAVE_T>(a_in, ONES, Tr, rwa, cl, a_ts, (ga - ba) * S::BR, ba, H, gw, lane);
      }
      nst = (has_a ? st_a : 0) + ((has_b && b_out) ? st_b : 0);
    }
    if (i == 8) SR_STAMP_AT(3);
    if (nst >= 9) wait_vmcnt<9>();                     // this wave's x pieces of the next band have landed
    else if (nst >= 6) wait_vmcnt<6>();
    else if (nst >= 3) wait_vmcnt<3>();
    else wait_vmcnt<0>();
    if (i == 8) SR_STAMP_AT(4);
    __syncthreads();
    if (i == 8) SR_STAMP_AT(5);
    if (i == 0) SR_STAMP_AT(6);
    if (i == GB + 4) SR_STAMP_AT(7);
  }
}


// =============================================================================================
// TWELVE waves, one ROLE per wave (round 3, second form).  The eight-wave kernel above gives every wave one block's whole
// weight set (124 VGPRs) and two waves per SIMD; its counters say the matrix pipe and the vector pipe are busy one AFTER the
// other (MFMA busy 40-48 %, VALU the rest): two waves do not cover each other's dependent chains.  Here a wave holds the
// weights of ONE phase only -- waves 0-2: conv1/conv2 of block 0 (76 VGPRs), 3-5: its 3x3 (48), 6-8: conv1/conv2 of block 1,
// 9-11: its 3x3 -- which fits three waves per SIMD (168 VGPRs each).  A role's three waves take two of the band's six pixel
// tiles each, pipelined as two jobs; the four roles of a round run side by side with the same lags (A0: band i, B0: i - 2,
// A1: i - 3, B1: i - 5) and the same single barrier.  Same per-pixel arithmetic: bit-identical.
// =============================================================================================
template <int F, int E, int L, bool SAVE_T>
__global__ __launch_bounds__(768) void wdsr_fwd_stream12_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ ya,
                                                                __bf16* __restrict__ yb, const __bf16* __restrict__ wa,
                                                                const __bf16* __restrict__ wb, const float* __restrict__ cia,
                                                                const float* __restrict__ cib, __bf16* __restrict__ tsa,
                                                                __bf16* __restrict__ tsb, int N, int H) {
  typedef StreamCfg<F, E, L> S;
  typedef typename S::C C;
  typedef typename S::R R;
  constexpr int NW = 12, NWR = 3, NTHREADS = 64 * NW;
  static_assert(S::NTB == 2 * NWR, "two pixel tiles of a band per wave of a role");
  __shared__ __attribute__((aligned(16))) char smem_raw[S::LDS_BYTES];
  __bf16* const XR = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* const T0 = XR + S::X_ELEMS;
  __bf16* const Y0 = T0 + S::T_ELEMS;
  __bf16* const T1 = Y0 + S::X_ELEMS;
  __bf16* const ONES = T1 + S::T_ELEMS;
  __bf16* const ZROW = ONES + 8;
  float* const CL = reinterpret_cast<float*>(ONES + S::ONES_ELEMS);
  __bf16* const PARK = T0;                                             // prologue only: both blocks' weight fragments
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave / NWR, gw = wave - role * NWR;                 // 0: A0, 1: B0, 2: A1, 3: B1
  const int blk = role >> 1;
  const bool is_b = role & 1;
  const int NB = H / S::BR;
  const int K = ((int)blockIdx.x < N) ? (N - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;   // this workgroup's images
  const int GB = K * NB;                                               // its bands
  const size_t img_elems = (size_t)H * S::W * F;
  const int tiles_img = ((H + C::TH - 1) / C::TH) * (S::W / C::TW);
  const char* zeros = reinterpret_cast<const char*>(g_sr_const_chunks) + 16;
  auto image_of = [&](int g, int& b) {                  // global band g -> image index, band in the image
    const int k = g / NB;
    b = g - k * NB;
    return (int)blockIdx.x + k * (int)gridDim.x;
  };
  auto stage_x = [&](int g) {                           // a band's rows are ONE contiguous 9 KB run of its image: waves 0 .. 8 one piece each
    if (wave < S::XPIECES) {
      int b;
      const int n = image_of(g, b);
      const char* src = reinterpret_cast<const char*>(x + (size_t)n * img_elems) + (size_t)b * (S::BPX * S::KXL * 2);
      const unsigned dst = lds_addr(XR) + (g & 3) * (S::BPX * S::KXL * 2);
      dma_piece16(src + wave * 1024 + lane * 16, dst + wave * 1024);
    }
  };
  static_assert(S::XPIECES <= NW, "one x piece per wave");

  // ---- prologue: as in the eight-wave kernel ----
  if (GB > 0) stage_x(0);
#pragma unroll 1
  for (int p = R::P_C + wave; p < R::P_END; p += NW) {
    if (p < R::P_W) {
      const int k = p - R::P_C, b_ = k / (R::CL_FLOATS / 64), i = (k % (R::CL_FLOATS / 64)) * 64 + lane;
      const float* tab = b_ == 1 ? cib : cia;
      const char* src = i < C::CINIT_FWD ? reinterpret_cast<const char*>(tab + i) : zeros + (lane & 3) * 4;
      dma_piece4(src, lds_addr(CL) + k * 256);
    } else {
      const int fr = p - R::P_W;
      const __bf16* wsrc = fr >= R::NFR ? wb + (size_t)R::src_frag(fr - R::NFR) * 512 : wa + (size_t)R::src_frag(fr) * 512;
      dma_piece16(reinterpret_cast<const char*>(wsrc + lane * 8), lds_addr(PARK) + fr * 1024);
    }
  }
  if (tid < S::ONES_ELEMS) ONES[tid] = tid == 0 ? (__bf16)1.f : (__bf16)0.f;
  wait_vmcnt<0>();
  __syncthreads();

  const float* const cl = CL + blk * R::CL_FLOATS;
  constexpr int BAND_ELEMS = S::BPX * S::KXL;
  __bf16* const ts = blk ? tsb : tsa;
  __bf16* const Tr = blk ? T1 : T0;
  __bf16* const In = blk ? Y0 : XR;                     // this block's input ring

  // after the weights are in registers: the parking area is freed, the rings zeroed (and the saved images' padding rows)
  auto after_weights = [&]() {
    __syncthreads();
    {
      const u32x4 z = {0u, 0u, 0u, 0u};
      u32x4* t0 = reinterpret_cast<u32x4*>(T0);
      u32x4* t1 = reinterpret_cast<u32x4*>(T1);
      for (int i = tid; i < S::T_ELEMS * 2 / 16; i += NTHREADS) { t0[i] = z; t1[i] = z; }
    }
    if constexpr (SAVE_T) {
      if (!is_b) {                                      // the conv1/conv2 waves of a block: its saved image's rows below the picture
        const int rows_pad = ((H + C::TH - 1) / C::TH) * C::TH - H;
        constexpr int PPX = C::LP * 2 / 16;
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int k = 0; k < K; ++k) {
          __bf16* const ts_img = ts + (size_t)((int)blockIdx.x + k * (int)gridDim.x) * tiles_img * (C::TH * C::TW) * C::LP;
          for (int i = gw * 64 + lane; i < rows_pad * S::W * PPX; i += 64 * NWR) {
            const int px = i / PPX, q = i - px * PPX;
            const int y = H + px / S::W, c = px % S::W;
            const int ty = y / C::TH, tx = c / C::TW;
            const int off = (((ty * (S::W / C::TW) + tx) * (C::TH * C::TW)) + (y - ty * C::TH) * C::TW + (c - tx * C::TW)) * C::LP + q * 8;
            stream_store(reinterpret_cast<u32x4*>(ts_img + off), z);
          }
        }
      }
    }
    __syncthreads();
  };
  // end of a round: this wave's x piece of the next band has landed (vmcnt retires in issue order: the piece was issued before
  // the round's `nst` global stores, which stay in flight), then the barrier
  auto end_round = [&](int nst) {
    if (nst >= 6) wait_vmcnt<6>();
    else wait_vmcnt<0>();
    __syncthreads();
  };

  if (!is_b) {
    RwA<C> rwa;
    RwB<C> none;                                         // (never read)
    rwa.load(PARK + blk * R::W_ELEMS, lane);
    after_weights();
    const int lag = blk ? 3 : 0;
#pragma unroll 1
    for (int i = 0; i < GB + 5; ++i) {
      if (i + 1 < GB) stage_x(i + 1);
      const int ga = i - lag;
      const bool has = ga >= 0 && ga < GB;
      if (has) {
        int ba = 0;
        const int na = image_of(ga, ba);
        const __bf16* const a_in = In + (ga & 3) * BAND_ELEMS;
        __bf16* const a_ts = SAVE_T ? ts + (size_t)na * tiles_img * (C::TH * C::TW) * C::LP : nullptr;
        StPipe<S, SAVE_T> pp;
        pp.ones = ONES; pp.zrow = ZROW; pp.H = H; pp.lane = lane;
        pp.ca = {a_in, Tr, cl, a_ts, (ga - ba) * S::BR, ba};
        pp.template run2<ST_JA, ST_JA>(gw, gw + NWR, rwa, none);
      }
      end_round((SAVE_T && has) ? 6 : 0);
    }
  } else {
    RwA<C> none;
    RwB<C> rwb;
    rwb.load(PARK + blk * R::W_ELEMS, lane);
    after_weights();
    const int lag = blk ? 5 : 2;
#pragma unroll 1
    for (int i = 0; i < GB + 5; ++i) {
      if (i + 1 < GB) stage_x(i + 1);
      const int gb = i - lag;
      const bool has = gb >= 0 && gb < GB;
      bool stores = false;
      if (has) {
        int bb = 0;
        const int nb = image_of(gb, bb);
        const __bf16* const b_res = In + (gb & 3) * BAND_ELEMS;
        __bf16* const b_next = blk ? nullptr : Y0 + (gb & 3) * BAND_ELEMS;
        __bf16* const b_out = blk ? yb + (size_t)nb * img_elems : (ya ? ya + (size_t)nb * img_elems : nullptr);
        stores = b_out != nullptr;
        StPipe<S, SAVE_T> pp;
        pp.ones = ONES; pp.zrow = ZROW; pp.H = H; pp.lane = lane;
        pp.cb = {Tr, b_res, b_next, b_out, (gb - bb) * S::BR, bb};
        pp.template run2<ST_JB, ST_JB>(gw, gw + NWR, none, rwb);
      }
      end_round(stores ? 6 : 0);
    }
  }
}
