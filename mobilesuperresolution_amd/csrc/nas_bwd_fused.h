// NAS supernet block, backward: the pointwise backward (nas_pw_bwd_kernel) and the depthwise weight gradients
// (nas_dw_wgrad3_kernel) of a block from ONE launch (bf16, one tile per workgroup).
//
// The weight-gradient kernel stages exactly what the pointwise backward has just produced -- the core tiles of GZ_0..2 -- plus
// the halo'd m1 * yin tile.  Here every item of the pointwise part overwrites its 32-pixel tile of V_k in LDS with the GZ_k it
// computed (after its last read of that tile; no other wave touches it), the m1 * yin tile is staged at the start with the V
// tiles (its loads are in flight together with theirs), and after the pointwise epilogue the 83 taps run over the LDS images
// as in nas_dw_wgrad3_kernel.  Saved per block: a launch and its gap, the 14 MB read of GZ and one staging latency.
// The pointwise epilogue's slab copies cannot overlay the V tiles any more (they hold GZ now): the six copies of the waves
// grp 0, 1 go to the scratch / weight region, the waves grp 2, 3 add theirs (loads first, then stores), two copies per
// branch are summed on the way out.  Same arithmetic as the two kernels: GZ, dWdw bit-identical; the pointwise slab sums
// in a different order.
#pragma once
#include "nas_block.h"

template <int F>
__global__ __launch_bounds__(768) void nas_block_bwd_a_kernel(const __bf16* __restrict__ yin, const __bf16* __restrict__ V,
                                                              const __bf16* __restrict__ gy, __bf16* __restrict__ GZ,
                                                              const __bf16* __restrict__ frags, const float* __restrict__ tabs,
                                                              const float* __restrict__ scal, const float* __restrict__ dwp,
                                                              float* __restrict__ part_pw, float* __restrict__ part_dw, int N, int H,
                                                              int W, int tiles_x, int tiles_per_img, long vstride) {
  typedef __bf16 T;
  typedef NasCfg<F> C;
  typedef typename FragOf<T>::type FragT;
  typedef typename FragOf<T>::half_type HalfT;
  constexpr int GW = 4, NTHREADS = 768, SCR = 33 * 32, X_ELEMS = (C::NP3 + 2) * 32;
  constexpr int SCR_OFF = 3 * C::VT_ELEMS * 2;                       // bytes
  constexpr int WL_OFF = SCR_OFF + 3 * GW * SCR * 2;                 // 12 weight fragments
  constexpr int TB_OFF = WL_OFF + 12 * 512 * 2;                      // bp[3][32] | ms | mg (C-init layout)
  constexpr int X1_OFF = TB_OFF + 160 * 4;                           // m1 * yin with a 3-pixel halo, [NP3 + 2][32]
  constexpr int LDS_BYTES = X1_OFF + X_ELEMS * 2;
  static_assert(X1_OFF % 16 == 0, "fragment stores");
  static_assert(SCR_OFF + (6 * C::PWB_K + 16) * 4 <= TB_OFF, "the slab copies overlay the scratch and weight regions only");
  __shared__ __attribute__((aligned(16))) char smem_raw[LDS_BYTES];
  T* const VT = reinterpret_cast<T*>(smem_raw);                      // V_k tiles, then GZ_k tiles: [3][NPXC + 1][32]
  const T* const WL = reinterpret_cast<const T*>(smem_raw + WL_OFF);
  const float* const TB = reinterpret_cast<const float*>(smem_raw + TB_OFF);
  T* const X1 = reinterpret_cast<T*>(smem_raw + X1_OFF);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  T* const scr = reinterpret_cast<T*>(smem_raw + SCR_OFF) + wave * SCR;
  const int k = wave / GW, grp = wave - k * GW;                      // 3 x GW waves: (branch, pixel-tile residue)
  T* const VTk = VT + k * C::VT_ELEMS;
  const int t = blockIdx.x;
  const bool has_tile = t < N * tiles_per_img;
  const int n = has_tile ? t / tiles_per_img : 0, tile = has_tile ? t - n * tiles_per_img : 0;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const size_t img = (size_t)n * H * W * F;
  float* const out_pw = part_pw + (size_t)blockIdx.x * C::PWB_SLAB;
  float* const out_dw = part_dw + (size_t)blockIdx.x * C::DWB_SLAB;
  if (!has_tile) {                                                   // (uniform) a workgroup without a tile: zero slabs
    for (int i = tid; i < C::PWB_SLAB; i += NTHREADS) out_pw[i] = 0.f;
    for (int i = tid; i < 83 * 32; i += NTHREADS) out_dw[i] = 0.f;
    return;
  }

  // ---- stage: V tiles, m1 * yin halo tile, weights (backward fragments scaled by c_k = p_k beta2 ms), tables ----
  {
    constexpr int TOTAL = (C::NP3 + 2) * 4, IT = (TOTAL + NTHREADS - 1) / NTHREADS;
    FragT f[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NTHREADS, hp = idx >> 2, c = idx & 3;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[it][j] = (T)0.f;
      if (hp < C::NP3 && c < C::FC) {
        const int hy = hp / C::PW, hx = hp - hy * C::PW;
        const int Y = ty0 - 3 + hy, X = tx0 - 3 + hx;
        if (Y >= 0 && Y < H && X >= 0 && X < W) f[it] = *reinterpret_cast<const FragT*>(yin + img + ((size_t)Y * W + X) * F + c * 8);
      }
    }
    nas_stage_vt3<T, C, NTHREADS>(VT, V + img, vstride, H, W, ty0, tx0, tid);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = tid + it * NTHREADS, c = idx & 3;
      if (idx < TOTAL) {
        FragT v = f[it];
        if (c < C::FC) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (T)((float)v[j] * dwp[C::M1 + c * 8 + j]);
        }
        *reinterpret_cast<FragT*>(X1 + idx * 8) = v;
      }
    }
    const float b2 = scal[3];
    for (int i = tid; i < 12 * 64; i += NTHREADS) {
      FragT v = reinterpret_cast<const FragT*>(frags)[i];
      const int f_ = i >> 6, l = i & 63;
      if (f_ >= 6) {
        const int kk = (f_ - 6) >> 1, st = (f_ - 6) & 1;
        const float pk = (kk == 0 ? scal[0] : (kk == 1 ? scal[1] : scal[2])) * b2;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (T)((float)v[j] * pk * tabs[96 + (l >> 5) * 16 + 8 * st + j]);
      }
      reinterpret_cast<FragT*>(smem_raw + WL_OFF)[i] = v;
    }
    if (tid < 160) reinterpret_cast<float*>(smem_raw + TB_OFF)[tid] = tabs[tid];
  }
  __syncthreads();

  // ---- pointwise backward (nas_pw_bwd_kernel's arithmetic) ----
  f32x16 dW = zero16();
  float db[16], rk[16], sxy = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { db[i] = 0.f; rk[i] = 0.f; }
  for (int idx = tid; idx < C::NPXC * C::FC; idx += NTHREADS) {      // sxy = sum gy mg yin
    const int pcx = idx / C::FC, c = idx - pcx * C::FC;
    const int Y = ty0 + pcx / C::TW, X = tx0 + pcx % C::TW;
    if (Y < H && X < W) {
      const size_t oo = img + ((size_t)Y * W + X) * F + c * 8;
      const FragT a = *reinterpret_cast<const FragT*>(gy + oo), b = *reinterpret_cast<const FragT*>(yin + oo);
#pragma unroll
      for (int j = 0; j < 8; ++j) sxy += (float)a[j] * TB[128 + (j >> 2) * 16 + 4 * c + (j & 3)] * (float)b[j];
    }
  }
#pragma unroll 1
  for (int ot = (grp + GW - k % GW) % GW; ot < C::NPT_O; ot += GW) {
    const int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
    const int oy = toy + (r >> 3), ox = tox + (r & 7);
    const int pc = oy * C::TW + ox;
    const int Y = ty0 + oy, X = tx0 + ox;
    const bool valid = Y < H && X < W;
    const size_t o = img + ((size_t)(valid ? Y : 0) * W + (valid ? X : 0)) * F;
    f32x16 acc = load_cinit(TB + k * 32, hh);
#pragma unroll
    for (int s = 0; s < 2; ++s) acc = mma16<T>(load_wfrag<T>(WL, 2 * k + s, lane), lds_chunk<T>(VTk, pc * 32 + (2 * s + hh) * 8), acc);
    float g[16];
    nas_load_rows<T, F>(g, gy + o, hh);
    if (!valid) {
#pragma unroll
      for (int i = 0; i < 16; ++i) g[i] = 0.f;
    }
    f32x16 gu;                                         // m = gy 1(u > 0); relu(u) gy = m u
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      gu[i] = acc[i] > 0.f ? g[i] : 0.f;
      rk[i] += gu[i] * acc[i];
      db[i] += gu[i];
    }
    const FragT gu0 = acc_to_frag<T, 0>(gu), gu1 = acc_to_frag<T, 1>(gu);
    scratch_store<T>(scr, gu, true, r, hh);
    f32x16 gv = zero16();
    gv = mma16<T>(load_wfrag<T>(WL, 6 + 2 * k, lane), gu0, gv);
    gv = mma16<T>(load_wfrag<T>(WL, 6 + 2 * k + 1, lane), gu1, gv);
    HalfT z[4];                                        // GZ_k of this pixel (zero outside the image: gy was zeroed)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const HalfT vv = *reinterpret_cast<const HalfT*>(VTk + pc * 32 + gq * 8 + hh * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) z[gq][j] = (float)vv[j] > 0.f ? (T)gv[4 * gq + j] : (T)0.f;
      if (valid && gq < C::FC) *reinterpret_cast<HalfT*>(GZ + k * vstride + o + gq * 8 + hh * 4) = z[gq];
    }
    auto rows = [](int p) { return p * 32; };
    auto rowv = [=](int p) { return ((toy + (p >> 3)) * C::TW + tox + (p & 7)) * 32; };
    dW = mma16<T>(tr_frag<T>(scr, 0, lane, rows), tr_frag<T>(VTk, 0, lane, rowv), dW);
    dW = mma16<T>(tr_frag<T>(scr, 1, lane, rows), tr_frag<T>(VTk, 1, lane, rowv), dW);
    // this tile of V_k is through: it becomes the GZ_k tile the weight gradients read (same wave, LDS in order)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<HalfT*>(VTk + pc * 32 + gq * 8 + hh * 4) = z[gq];
  }
  {
    const f32x16 ms = load_cinit(TB + 96, hh);
    const float ck = scal[k] * scal[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float c = ck * ms[i]; dW[i] *= c; db[i] *= c; }
  }
  const float red = half_sum32(db, rk, r);             // lane r: db[r] (r < 16) or rk[r - 16], summed over the half's pixels
  const float sxy_w = wave_sum(sxy);
  const int red_at = 1024 + 2 * (r & 16) + ((r & 15) & 3) + 8 * ((r & 15) >> 2) + 4 * hh;
  __syncthreads();                                     // every item done: GZ tiles complete, scratch / weights / tables free
  float* const S = reinterpret_cast<float*>(smem_raw + SCR_OFF);
  float* const sw = S + (k * 2 + (grp & 1)) * C::PWB_K;
  float* const sx = S + 6 * C::PWB_K;
  if (grp < 2) {
#pragma unroll
    for (int i = 0; i < 16; ++i) sw[i * 64 + lane] = dW[i];
    sw[red_at] = red;
  }
  if (lane == 0) sx[wave] = sxy_w;
  __syncthreads();
  if (grp >= 2) {
    float old[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) old[i] = sw[i * 64 + lane];
    const float oldr = sw[red_at];
#pragma unroll
    for (int i = 0; i < 16; ++i) sw[i * 64 + lane] = old[i] + dW[i];
    sw[red_at] = oldr + red;
  }
  __syncthreads();
  for (int i = tid; i < 3 * C::PWB_K; i += NTHREADS) {
    const int kk = i / C::PWB_K, e = i - kk * C::PWB_K;
    out_pw[i] = S[(kk * 2) * C::PWB_K + e] + S[(kk * 2 + 1) * C::PWB_K + e];
  }
  if (tid < 4) {
    float v = 0.f;
    if (tid == 0)
      for (int w = 0; w < 3 * GW; ++w) v += sx[w];
    out_pw[3 * C::PWB_K + tid] = v;
  }

  // ---- depthwise weight gradients (nas_dw_wgrad3_kernel's loop) over the LDS images ----
  constexpr int NSLOT = 7, NTAP = 83;
  f32x16 acc[NSLOT];
  int xoff[NSLOT], goff[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    acc[i] = zero16();
    const int tp = wave + 12 * i;
    const int kq = tp < 9 ? 0 : (tp < 34 ? 1 : 2), lt = tp - (kq == 0 ? 0 : (kq == 1 ? 9 : 34)), ks = 3 + 2 * kq, off = 3 - ks / 2;
    const int ty = lt / ks, tx = lt - ty * ks;
    xoff[i] = __builtin_amdgcn_readfirstlane(((off + ty) * C::PW + off + tx) * 32);
    goff[i] = __builtin_amdgcn_readfirstlane(kq * C::VT_ELEMS);
  }
#pragma unroll 1
  for (int ot = 0; ot < C::NPT_O; ++ot) {
    const int toy = (ot / (C::TW / 8)) * 4, tox = (ot % (C::TW / 8)) * 8;
    const int gbase = (toy * C::TW + tox) * 32, xbase = (toy * C::PW + tox) * 32;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      FragT a;
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) {
        if (wave + 12 * i < NTAP) {
          if (i == 0 || goff[i] != goff[i - 1]) {
            const T* gq = VT + goff[i] + gbase;
            a = tr_frag<T>(gq, s, lane, [](int p) { return ((p >> 3) * C::TW + (p & 7)) * 32; });
          }
          const T* x = X1 + xoff[i] + xbase;
          acc[i] = mma16<T>(a, tr_frag<T>(x, s, lane, [](int p) { return ((p >> 3) * C::PW + (p & 7)) * 32; }), acc[i]);
        }
      }
    }
  }
  // diagonal of every tile: accumulator register q of lane (r, hh) is row (q & 3) + 8 (q >> 2) + 4 hh, column r
  const bool mine = hh == ((r >> 2) & 1);
  const int isel = (r & 3) + 4 * (r >> 3);
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int tp = wave + 12 * i;
    if (tp < NTAP) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) v = (q == isel) ? acc[i][q] : v;
      if (mine) out_dw[tp * 32 + r] = v;
    }
  }
}
