// psnr / psnr_y of the evaluation loop on device (gfx950).  Reference ops replaced: common/metrics.py:10-19 (psnr) and
// :22-38 (psnr_y), called by utils/estimate.py:123-128 on `model(lr).to('cpu')`: here the SR image never leaves HBM.
//   psnr   : sr -> round-half-even(sr * 255) clamped to [0, 255], / 255, clamp [0, 1]; d = sr - hr
//   psnr_y : sr -> clamp [0, 1] only (the reference computes the quantised copy `r` and never uses it, :24-25);
//            d = 0.257 dR + 0.504 dG + 0.098 dB when there are 3 channels (the luma filter on the DIFFERENCE, :29-33)
// both: shave the border, mse over (C, H', W') per image, -10 log10(mse) SUMMED over the batch (:19, :38).
#pragma once
#include "sr_common.h"

// LUMA 1: psnr_y with the luma filter; 0: psnr (quantised); -1: psnr_y on a tensor without 3 channels (no filter, no quantisation)
template <int LUMA>
__global__ __launch_bounds__(256) void sr_sqdiff_kernel(const float* __restrict__ sr, const float* __restrict__ hr,
                                                        float* __restrict__ partial, int C, int H, int W, int shave) {
  __shared__ float red[4];
  const int n = blockIdx.y, hs = H - 2 * shave, ws = W - 2 * shave;
  const size_t plane = (size_t)H * W;
  const float* s = sr + (size_t)n * C * plane;
  const float* h = hr + (size_t)n * C * plane;
  float acc = 0.f;
  if (hs > 0 && ws > 0) {
    const long npx = (long)hs * ws, total = LUMA == 1 ? npx : npx * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      const long px = LUMA == 1 ? i : i % npx;
      const int c = LUMA == 1 ? 0 : (int)(i / npx);
      const int y = (int)(px / ws) + shave, x = (int)(px % ws) + shave;
      const size_t o = (size_t)y * W + x;
      float d;
      if (LUMA == 1) {
        const float d0 = fminf(fmaxf(s[o], 0.f), 1.f) - h[o];
        const float d1 = fminf(fmaxf(s[o + plane], 0.f), 1.f) - h[o + plane];
        const float d2 = fminf(fmaxf(s[o + 2 * plane], 0.f), 1.f) - h[o + 2 * plane];
        d = 0.257f * d0 + 0.504f * d1 + 0.098f * d2;
      } else {
        float v = s[o + (size_t)c * plane];
        if (LUMA == 0) v = fminf(fmaxf(rintf(v * 255.f), 0.f), 255.f) / 255.f;
        d = fminf(fmaxf(v, 0.f), 1.f) - h[o + (size_t)c * plane];
      }
      acc += d * d;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)n * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(64) void sr_psnr_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int N, int wgs,
                                                            double count) {
  double sum = 0.0;
  for (int n = threadIdx.x; n < N; n += 64) {
    double sq = 0.0;
    for (int w = 0; w < wgs; ++w) sq += (double)partial[(size_t)n * wgs + w];
    sum += -10.0 * log10(sq / count);                // count == 0 (everything shaved): nan, as the reference's mean of nothing
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
  if (threadIdx.x == 0) out[0] = (float)sum;
}
