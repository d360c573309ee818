// 7x7 convolutions of SPyNet's BasicModule (gfx950, bf16 MFMA).  Reference op: models/spynet_arch.py:17-22 -- five
// nn.Conv2d(k=7, pad=3) 8 -> 32 -> 64 -> 32 -> 16 -> 2 with ReLU between, run on six pyramid levels (:63-77).  These are the
// one place on the path where the contraction is deep (K = 49 Cin = 392 ... 3136): an implicit GEMM on the matrix cores,
// output channels in rows, 32 pixels of one image row on the lanes, K walked tap by tap.
//
// One workgroup = 8 waves = an 8 x 32 pixel output tile (one row per wave); the (8 + 6) x (32 + 6) input halo tile sits in LDS
// in NHWC order (all Cin channels of a pixel contiguous), so a lane's B fragment of k-step (ky, kx, 16-channel chunk) is ONE
// 16-byte read at a compile-time offset from its pixel.  The weights are packed on the host as MFMA A fragments in exactly the
// order the loop consumes them and streamed through LDS one kernel row at a time by LDS-DMA (<= 28 KB per row, double
// buffered: row ky + 1 lands while row ky is multiplied).  Cin = 8 (first layer) packs two horizontally adjacent taps into
// one 16-deep k-step.  Both operands come from LDS, so the kernel is LDS-bandwidth bound at about half the matrix rate -- the
// register-resident-weights form of wdsr_fwd_rs.h does not apply (a layer's weights are 25 - 200 KB).
#pragma once
#include "wdsr_fwd_rs.h"

template <int CIN, int COUT> struct Conv7Cfg {
  static_assert(CIN == 8 || CIN % 16 == 0, "input channels: 8 or a multiple of 16");
  static constexpr int TH = 8, TW = 32, HH = TH + 6, HW = TW + 6;
  static constexpr int MT = (COUT + 31) / 32;                      // 32-row output-channel tiles
  static constexpr int KPR = CIN == 8 ? 4 : 7 * (CIN / 16);        // k-steps per kernel row
  static constexpr int FR_ROW = KPR * MT;                          // weight fragments per kernel row
  static constexpr int X_ELEMS = (HH * HW + 2) * CIN;              // (+2 pixels: the paired tap of the last column reads one past)
  static constexpr int LDS_BYTES = X_ELEMS * 2 + 2 * FR_ROW * 1024;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// x: [N][H][W][CIN] bf16; w: packed fragments [7][KPR][MT][64 lanes][8]; bias: [MT * 32] fp32 (zero past COUT);
// y: [N][H][W][COUT] bf16 (OUT_F32: fp32).  grid = (tiles_x * tiles_y, N), 512 threads.
template <int CIN, int COUT, bool RELU, bool OUT_F32>
__global__ __launch_bounds__(512) void conv7_fwd_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ w,
                                                        const float* __restrict__ bias, void* __restrict__ yv, int H, int W,
                                                        int tiles_x) {
  typedef Conv7Cfg<CIN, COUT> K;
  __shared__ __attribute__((aligned(16))) char smem_raw[K::LDS_BYTES];
  __bf16* const Xs = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* const Ws = Xs + K::X_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y, tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * K::TH, tx0 = (tile % tiles_x) * K::TW;
  const __bf16* xin = x + (size_t)n * H * W * CIN;

  auto stage_w = [&](int ky) {                          // kernel row ky: FR_ROW fragments of 1 KB, as they lie
    const char* src = reinterpret_cast<const char*>(w) + (size_t)ky * K::FR_ROW * 1024;
    const unsigned dst = lds_addr(Ws) + (ky & 1) * K::FR_ROW * 1024;
#pragma unroll 1
    for (int p = wave; p < K::FR_ROW; p += 8) dma_piece16(src + p * 1024 + lane * 16, dst + p * 1024);
  };
  stage_w(0);
  {                                                     // input halo tile, zero outside the image
    constexpr int CH = CIN / 8, TOTAL = (K::HH * K::HW + 2) * CH;
    for (int idx = tid; idx < TOTAL; idx += 512) {
      const int p = idx / CH, c = idx - p * CH;
      const int py = p / K::HW, px = p - py * K::HW;
      const int Y = ty0 - 3 + py, X = tx0 - 3 + px;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
      if (p < K::HH * K::HW && Y >= 0 && Y < H && X >= 0 && X < W) v = *reinterpret_cast<const bf16x8*>(xin + ((size_t)Y * W + X) * CIN + c * 8);
      *reinterpret_cast<bf16x8*>(Xs + idx * 8) = v;
    }
  }
  f32x16 acc[K::MT];
#pragma unroll
  for (int m = 0; m < K::MT; ++m) {                     // bias: row (i & 3) + 8 (i >> 2) + 4 hh of tile m
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = bias[m * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh];
  }
  const __bf16* const xrow = Xs + (wave * K::HW + r) * CIN;       // this lane's pixel at tap (0, 0)
#pragma unroll 1
  for (int ky = 0; ky < 7; ++ky) {
    wait_vmcnt<0>();                                    // this wave's pieces of row ky (and, the first time, nothing else)
    __syncthreads();                                    // ... everybody's; the previous row's reads are over
    if (ky + 1 < 7) stage_w(ky + 1);
    const __bf16* const wl = Ws + (ky & 1) * K::FR_ROW * 512;
    const __bf16* const xk = xrow + ky * K::HW * CIN;
#pragma unroll
    for (int s = 0; s < K::KPR; ++s) {
      // B fragment: Cin >= 16: tap kx = s / (CIN / 16), channels 16 (s % (CIN / 16)) + 8 hh ..; Cin = 8: tap kx = 2 s + hh, all 8
      const int off = CIN == 8 ? (2 * s + hh) * 8 : (s / (CIN / 16)) * CIN + (s % (CIN / 16)) * 16 + hh * 8;
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(xk + off);
#pragma unroll
      for (int m = 0; m < K::MT; ++m)
        acc[m] = mma16<__bf16>(lds_chunk<__bf16>(wl, ((s * K::MT + m) * 64 + lane) * 8), b, acc[m]);
    }
  }
  const int Y = ty0 + wave, X = tx0 + r;
  if (Y < H && X < W) {
    const size_t pix = ((size_t)n * H + Y) * W + X;
#pragma unroll
    for (int m = 0; m < K::MT; ++m) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = m * 32 + g * 8 + hh * 4;         // channels ch .. ch + 3 in regs 4 g .. 4 g + 3
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = RELU ? fmaxf(acc[m][4 * g + k], 0.f) : acc[m][4 * g + k];
        if constexpr (OUT_F32) {
          float* y = reinterpret_cast<float*>(yv) + pix * COUT;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (ch + k < COUT) y[ch + k] = v[k];
        } else {
          __bf16* y = reinterpret_cast<__bf16*>(yv) + pix * COUT;
          if (ch + 4 <= COUT) {
            __bf16 o0, o1, o2, o3;
            cvt_pair<__bf16>(o0, o1, v[0], v[1]);
            cvt_pair<__bf16>(o2, o3, v[2], v[3]);
            bf16x4 o;
            o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
            *reinterpret_cast<bf16x4*>(y + ch) = o;
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (ch + k < COUT) y[ch + k] = (__bf16)v[k];
          }
        }
      }
    }
  }
}
