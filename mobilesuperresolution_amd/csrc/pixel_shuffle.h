// Standalone PixelShuffle (nn.PixelShuffle(r), models/basic_wdsr_b.py:80-83 / basicvsr_arch_origin.py:37,87-88):
//   out[n, c, h r + i, w r + j] = in[n, c r^2 + i r + j, h, w]       NCHW fp32, pure index permutation (bit-exact).
// The WDSR tail never runs this (sr_tail_fwd writes the shuffled HR tile straight from its accumulator); it serves the
// BasicVSR-origin upsampler, whose convolutions produce far more channels than the 3 r^2 the fused tail handles.
// Forward: one thread per 4 consecutive output pixels (one 16-byte store; the 4 reads of neighbouring lanes fall into
// the same input rows).  The inverse (the op's backward) mirrors it: one 16-byte store of input-side gradients.
#pragma once
#include "sr_common.h"

template <bool INVERSE>
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H,
                                                            int W, int r, long total4) {
  // C = channels of the SHUFFLED (HR) tensor, H x W = LR size
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total4) return;
  const int HR = H * r, WR = W * r;
  if constexpr (!INVERSE) {
    const int wq = (WR + 3) / 4;
    const int ox0 = (int)(t % wq) * 4;
    long rest = t / wq;
    const int oy = (int)(rest % HR);
    rest /= HR;
    const int c = (int)(rest % C), n = (int)(rest / C);
    const int h = oy / r, i = oy - h * r;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ox = ox0 + k;
      const int w = ox / r, j = ox - w * r;
      v[k] = ox < WR ? src[(((long)n * C * r * r + (long)c * r * r + i * r + j) * H + h) * W + w] : 0.f;
    }
    float* o = dst + (((long)n * C + c) * HR + oy) * WR + ox0;
    if (ox0 + 3 < WR && (WR & 3) == 0) {
      f32x4 q = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(o) = q;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ox0 + k < WR) o[k] = v[k];
    }
  } else {
    const int wq = (W + 3) / 4;
    const int w0 = (int)(t % wq) * 4;
    long rest = t / wq;
    const int h = (int)(rest % H);
    rest /= H;
    const int cc = (int)(rest % (C * r * r)), n = (int)(rest / (C * r * r));
    const int c = cc / (r * r), ij = cc - c * r * r, i = ij / r, j = ij - i * r;
    const float* s = src + (((long)n * C + c) * HR + (h * r + i)) * WR + j;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = w0 + k < W ? s[(long)(w0 + k) * r] : 0.f;
    float* o = dst + (((long)n * C * r * r + cc) * H + h) * W + w0;
    if (w0 + 3 < W && (W & 3) == 0) {
      f32x4 q = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(o) = q;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (w0 + k < W) o[k] = v[k];
    }
  }
}
