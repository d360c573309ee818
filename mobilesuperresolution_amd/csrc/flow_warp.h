// flow_warp (gfx950): bilinear gather at grid + flow, zeros padding, align_corners=True.
// Reference op replaced: flow_warp, models/spynet_arch.py:98-129 (vendored BasicSR copy of the mmedit
// function used at models/basicvsr_arch.py:74,85 and mvvsr_arch.py:79,90).  With align_corners=True the
// normalise / un-normalise pair cancels: the sample position of output pixel (y, x) is (x + fx, y + fy).
// x, out: NCHW fp32 (the recurrent state of the reference); flow: (N, H, W, 2) with [...,0] = dx, [...,1] = dy.
#pragma once
#include "sr_common.h"

struct WarpTaps { int x0, y0; float wx, wy; bool vx0, vx1, vy0, vy1; };

SR_DEV WarpTaps warp_taps(float px, float py, int H, int W) {
  WarpTaps t;
  const float fx = floorf(px), fy = floorf(py);
  t.x0 = (int)fx; t.y0 = (int)fy;
  t.wx = px - fx; t.wy = py - fy;
  t.vx0 = t.x0 >= 0 && t.x0 < W; t.vx1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
  t.vy0 = t.y0 >= 0 && t.y0 < H; t.vy1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
  return t;
}

// the reference normalises with max(w - 1, 1) and grid_sample un-normalises with (w - 1): reproduce the
// round trip in fp32 so that sample positions agree to the last bit for w > 1
SR_DEV float warp_pos(float g, int size) {
  const float d = (float)(size - 1 > 1 ? size - 1 : 1);
  const float v = 2.0f * g / d - 1.0f;
  return ((v + 1.0f) * 0.5f) * (float)(size - 1);
}

// One wave = 16 consecutive pixels x 4 channel phases (channels phase, phase + 4, ...): 4x the waves of a
// thread-per-pixel mapping and 4x shorter per-thread load chains; the flow gradient is reduced over the phases
// with two shuffles.
__global__ __launch_bounds__(64) void flow_warp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                           float* __restrict__ out, int C, int H, int W) {
  const int n = blockIdx.y, phase = threadIdx.x >> 4;
  const size_t plane = (size_t)H * W;
  for (int p = blockIdx.x * 16 + (threadIdx.x & 15); p < (int)plane; p += gridDim.x * 16) {
    const int y = p / W, xx = p - y * W;
    const float2 f = *reinterpret_cast<const float2*>(flow + ((size_t)n * plane + p) * 2);
    const WarpTaps t = warp_taps(warp_pos((float)xx + f.x, W), warp_pos((float)y + f.y, H), H, W);
    const float w00 = (1.f - t.wx) * (1.f - t.wy), w01 = t.wx * (1.f - t.wy), w10 = (1.f - t.wx) * t.wy, w11 = t.wx * t.wy;
    const size_t o00 = (size_t)t.y0 * W + t.x0;
#pragma unroll 8
    for (int c = phase; c < C; c += 4) {
      const float* xp = x + ((size_t)n * C + c) * plane;
      float v = 0.f;
      if (t.vy0 && t.vx0) v += w00 * xp[o00];
      if (t.vy0 && t.vx1) v += w01 * xp[o00 + 1];
      if (t.vy1 && t.vx0) v += w10 * xp[o00 + W];
      if (t.vy1 && t.vx1) v += w11 * xp[o00 + W + 1];
      out[((size_t)n * C + c) * plane + p] = v;
    }
  }
}

// dx must be zero-filled by the caller (scatter-add of the four corners); dflow (N, H, W, 2)
__global__ __launch_bounds__(64) void flow_warp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                           const float* __restrict__ gout, float* __restrict__ dx,
                                                           float* __restrict__ dflow, int C, int H, int W) {
  const int n = blockIdx.y, phase = threadIdx.x >> 4;
  const size_t plane = (size_t)H * W;
  const int iters = ((int)plane + gridDim.x * 16 - 1) / (gridDim.x * 16);      // uniform trip count: shuffles below
  for (int it = 0; it < iters; ++it) {
    const int p = (it * gridDim.x + blockIdx.x) * 16 + (threadIdx.x & 15);
    const bool live = p < (int)plane;
    const int pc = live ? p : 0;
    const int y = pc / W, xx = pc - y * W;
    const float2 f = *reinterpret_cast<const float2*>(flow + ((size_t)n * plane + pc) * 2);
    const WarpTaps t = warp_taps(warp_pos((float)xx + f.x, W), warp_pos((float)y + f.y, H), H, W);
    const float w00 = (1.f - t.wx) * (1.f - t.wy), w01 = t.wx * (1.f - t.wy), w10 = (1.f - t.wx) * t.wy, w11 = t.wx * t.wy;
    const size_t o00 = (size_t)t.y0 * W + t.x0;
    float gx = 0.f, gy = 0.f;
    if (live) {
#pragma unroll 8
      for (int c = phase; c < C; c += 4) {
        const size_t base = ((size_t)n * C + c) * plane;
        const float g = gout[base + p];
        const float v00 = (t.vy0 && t.vx0) ? x[base + o00] : 0.f, v01 = (t.vy0 && t.vx1) ? x[base + o00 + 1] : 0.f;
        const float v10 = (t.vy1 && t.vx0) ? x[base + o00 + W] : 0.f, v11 = (t.vy1 && t.vx1) ? x[base + o00 + W + 1] : 0.f;
        gx += g * ((v01 - v00) * (1.f - t.wy) + (v11 - v10) * t.wy);
        gy += g * ((v10 - v00) * (1.f - t.wx) + (v11 - v01) * t.wx);
        if (dx) {
          if (t.vy0 && t.vx0) unsafeAtomicAdd(dx + base + o00, w00 * g);
          if (t.vy0 && t.vx1) unsafeAtomicAdd(dx + base + o00 + 1, w01 * g);
          if (t.vy1 && t.vx0) unsafeAtomicAdd(dx + base + o00 + W, w10 * g);
          if (t.vy1 && t.vx1) unsafeAtomicAdd(dx + base + o00 + W + 1, w11 * g);
        }
      }
    }
    gx += __shfl_xor(gx, 16); gx += __shfl_xor(gx, 32);
    gy += __shfl_xor(gy, 16); gy += __shfl_xor(gy, 32);
    if (dflow && live && phase == 0) {
      // d(position)/d(flow) = (2 / max(size-1,1)) * ((size-1) / 2): 1 for size > 1, 0 for size == 1
      float2 o;
      o.x = W > 1 ? gx : 0.f;
      o.y = H > 1 ? gy : 0.f;
      *reinterpret_cast<float2*>(dflow + ((size_t)n * plane + p) * 2) = o;
    }
  }
}
