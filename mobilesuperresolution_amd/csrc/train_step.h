// Loss / optimizer epilogue of the training step.  Reference: pretrain.py:73-80 (L1 loss, backward,
// Adam step; Adam hyper-parameters pretrain.py:137) and train_video_superresolution.py:43-53 (Charbonnier).
#pragma once
#include "sr_common.h"

// One Adam step over the flat parameter buffer with the arithmetic of torch.optim.Adam's default (foreach)
// implementation, op for op and rounding for rounding (tests/test_gpu_train_step.py checks bit equality):
//   m = lerp(m, g, 1 - beta1)                        torch._foreach_lerp_   : m + w (g - m), fused multiply-add
//   v = v * beta2;  v = v + (1 - beta2) * (g * g)    _foreach_mul_, _foreach_addcmul_
//   d = sqrt(v) / sqrt(1 - beta2^t) + eps            _foreach_sqrt, _foreach_div_, _foreach_add_
//   p = p + (-lr / (1 - beta1^t)) * (m / d)          _foreach_addcdiv_
// The step-dependent scalars are computed by the caller in double and rounded to float, as torch does when it
// hands Python floats to its kernels.  Block 0 also folds the tail kernel's per-workgroup loss partial sums
// into the loss value (mean over the HR tensor times the loss weight), so reading the loss costs no launch.
struct AdamArgs {
  float w_lerp, beta2, one_minus_beta2, bc2_sqrt, eps, neg_step_size;
};
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n, AdamArgs a,
                                                        const float* __restrict__ loss_part, int n_loss, float loss_scale,
                                                        float* __restrict__ loss_out) {
  const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 + 3 < n) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i0);
    f32x4 mv = *reinterpret_cast<const f32x4*>(m + i0), vv = *reinterpret_cast<const f32x4*>(v + i0);
    f32x4 pv = *reinterpret_cast<const f32x4*>(p + i0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mv[j] = __builtin_fmaf(a.w_lerp, gv[j] - mv[j], mv[j]);
      const float vb = vv[j] * a.beta2;
      vv[j] = __builtin_fmaf(a.one_minus_beta2, gv[j] * gv[j], vb);
      const float d = __builtin_sqrtf(vv[j]) / a.bc2_sqrt + a.eps;
      pv[j] = __builtin_fmaf(a.neg_step_size, mv[j] / d, pv[j]);
    }
    *reinterpret_cast<f32x4*>(m + i0) = mv;
    *reinterpret_cast<f32x4*>(v + i0) = vv;
    *reinterpret_cast<f32x4*>(p + i0) = pv;
  } else {
    for (long i = i0; i < n; ++i) {
      const float gi = g[i];
      const float mi = __builtin_fmaf(a.w_lerp, gi - m[i], m[i]);
      const float vi = __builtin_fmaf(a.one_minus_beta2, gi * gi, v[i] * a.beta2);
      const float d = __builtin_sqrtf(vi) / a.bc2_sqrt + a.eps;
      m[i] = mi;
      v[i] = vi;
      p[i] = __builtin_fmaf(a.neg_step_size, mi / d, p[i]);
    }
  }
  if (loss_out && blockIdx.x == 0 && threadIdx.x < 64) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n_loss; i += 64) s += loss_part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) *loss_out = s * loss_scale;
  }
}

// loss value alone (the autograd route reads it without stepping an optimizer)
__global__ __launch_bounds__(64) void loss_sum_kernel(const float* __restrict__ loss_part, int n_loss, float loss_scale,
                                                      float* __restrict__ loss_out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n_loss; i += 64) s += loss_part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) *loss_out = s * loss_scale;
}

// y = x * (*scale), the product formed in fp32 (a folded loss's data gradient times whatever the trainer multiplied the loss
// with, which autograd hands over as a device scalar)
template <typename T>
__global__ __launch_bounds__(256) void scale_by_kernel(T* __restrict__ y, const T* __restrict__ x, long n, const float* __restrict__ scale) {
  const float s = *scale;
  const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i0 + 8 <= n) {
    typedef T V __attribute__((ext_vector_type(8)));
    V v = *reinterpret_cast<const V*>(x + i0);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (T)((float)v[j] * s);
    *reinterpret_cast<V*>(y + i0) = v;
  } else {
    for (long i = i0; i < n; ++i) y[i] = (T)((float)x[i] * s);
  }
}

// Weight-norm backward and the Adam step in ONE launch (single-process training step): a wave that has just produced the
// gradient of its channel's v row and g (or a thread its bias entry's) applies the Adam update to exactly those
// parameters -- every parameter of the network belongs to exactly one table row, so the separate pass over the flat
// buffer and its launch go away.  Same arithmetic as wn_bwd_kernel followed by adam_step_kernel, element for element
// (tests/test_gpu_train_step.py: parameters bit-identical to torch.optim.Adam's); the gradient is still written out.
SR_DEV void adam_elem(float& p, float g, float& m, float& v, const AdamArgs& a) {
  m = __builtin_fmaf(a.w_lerp, g - m, m);
  const float vb = v * a.beta2;
  v = __builtin_fmaf(a.one_minus_beta2, g * g, vb);
  const float d = __builtin_sqrtf(v) / a.bc2_sqrt + a.eps;
  p = __builtin_fmaf(a.neg_step_size, m / d, p);
}

__global__ __launch_bounds__(256) void wn_bwd_adam_kernel(float* __restrict__ flat, const float* __restrict__ dsrc,
                                                          float* __restrict__ gflat, float* __restrict__ em, float* __restrict__ ev,
                                                          const int4* __restrict__ chan_tab, int n_chan,
                                                          const int* __restrict__ bias_tab, int n_bias, int chan_blocks, AdamArgs a,
                                                          const float* __restrict__ loss_part, int n_loss, float loss_scale,
                                                          float* __restrict__ loss_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (loss_out && blockIdx.x == gridDim.x - 1 && threadIdx.x >= 192) {   // the last (bias) block's last wave also folds the loss
    float s = 0.f;
    for (int i = lane; i < n_loss; i += 64) s += loss_part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) *loss_out = s * loss_scale;
  }
  if ((int)blockIdx.x < chan_blocks) {
    const int c = blockIdx.x * 4 + wave;
    if (c >= n_chan) return;
    const int4 t = chan_tab[c];
    float* v = flat + t.x;
    const float* dw = dsrc + t.w;
    float ss = 0.f, dot = 0.f;
    for (int k = lane; k < t.z; k += 64) { const float x = v[k]; ss += x * x; dot += x * dw[k]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); dot += __shfl_xor(dot, o); }
    const float n = sqrtf(ss), g = flat[t.y];
    const float s1 = g / n, s2 = dot / ss;
    for (int k = lane; k < t.z; k += 64) {
      float gk = s1 * (dw[k] - v[k] * s2);
      asm volatile("" : "+v"(gk));                  // the gradient as wn_bwd_kernel rounds it: no contraction into the Adam arithmetic
      gflat[t.x + k] = gk;
      float pk = v[k], mk = em[t.x + k], vk = ev[t.x + k];
      adam_elem(pk, gk, mk, vk, a);
      v[k] = pk; em[t.x + k] = mk; ev[t.x + k] = vk;
    }
    if (lane == 0) {
      float gg = dot / n;
      asm volatile("" : "+v"(gg));
      gflat[t.y] = gg;
      float pg = g, mg = em[t.y], vg = ev[t.y];
      adam_elem(pg, gg, mg, vg, a);
      flat[t.y] = pg; em[t.y] = mg; ev[t.y] = vg;
    }
  } else {
    const int i = (blockIdx.x - chan_blocks) * 256 + threadIdx.x;
    if (i >= n_bias) return;
    const int pa = bias_tab[3 * i], pb = bias_tab[3 * i + 1], d = bias_tab[3 * i + 2];
    const float gval = dsrc[d];
    gflat[pa] = gval;
    { float pp = flat[pa], mm = em[pa], vv = ev[pa]; adam_elem(pp, gval, mm, vv, a); flat[pa] = pp; em[pa] = mm; ev[pa] = vv; }
    if (pb >= 0) {
      gflat[pb] = gval;
      float pp = flat[pb], mm = em[pb], vv = ev[pb];
      adam_elem(pp, gval, mm, vv, a);
      flat[pb] = pp; em[pb] = mm; ev[pb] = vv;
    }
  }
}

