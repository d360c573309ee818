// Micro-benchmark (diagnostic, not product): cycles per MFMA of issue patterns a lone wave (or two waves per
// SIMD) can sustain.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_sched tools/ubench/mfma_sched.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define DEV __device__ __forceinline__
DEV f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
DEV bf16x8 cvt8(const f32x16& a, int base) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; j += 2) { f32x2 v = {a[base + j], a[base + j + 1]}; bf16x2 p = __builtin_convertvector(v, bf16x2); f[j] = p[0]; f[j + 1] = p[1]; }
  return f;
}
DEV bf16x8 relu8(bf16x8 f) { s16x8 v = __builtin_bit_cast(s16x8, f); s16x8 z = {0,0,0,0,0,0,0,0}; return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(v, z)); }

// MODE 0: one dependent MFMA chain.  MODE 1: two independent chains.  MODE 2: the conv1 -> relu -> conv2 pattern of
// rs_t_tile (4 MFMA + 16 VALU per e-tile, weights in registers).  MODE 3: MODE 2 without sched_group_barrier.
template <int MODE>
__global__ __launch_bounds__(1024) void k(const bf16x8* w, const bf16x8* x, float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 w1[10], w2[9];
#pragma unroll
  for (int i = 0; i < 10; ++i) w1[i] = w[i * 64 + lane];
#pragma unroll
  for (int i = 0; i < 9; ++i) w2[i] = w[(10 + i) * 64 + lane];
  bf16x8 xb[2] = {x[lane], x[64 + lane]};
  f32x16 tacc;
#pragma unroll
  for (int i = 0; i < 16; ++i) tacc[i] = 0.f;
  f32x16 z = tacc;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 19; ++i) tacc = mma(i < 10 ? w1[i] : w2[i - 10], xb[i & 1], tacc);
    } else if constexpr (MODE == 1) {
      f32x16 a = tacc, b = z;
#pragma unroll
      for (int i = 0; i < 19; ++i) { if (i & 1) a = mma(w1[i % 10], xb[0], a); else b = mma(w2[i % 9], xb[1], b); }
#pragma unroll
      for (int i = 0; i < 16; ++i) tacc[i] = a[i] + b[i];
    } else {
      f32x16 h = mma(w1[0], xb[0], z);
      h = mma(w1[1], xb[1], h);
      bf16x8 f1prev = {};
#pragma unroll
      for (int et = 0; et < 5; ++et) {
        const bool more = et + 1 < 5;
        f32x16 hn = h;
        if (more) hn = mma(w1[2 * et + 2], xb[0], z);
        bf16x8 f0 = cvt8(h, 0);
        if (more) hn = mma(w1[2 * et + 3], xb[1], hn);
        f0 = relu8(f0);
        if (et > 0) tacc = mma(w2[2 * et - 1], f1prev, tacc);
        bf16x8 f1 = cvt8(h, 8);
        tacc = mma(w2[2 * et], f0, tacc);
        f1prev = relu8(f1);
        h = hn;
        if constexpr (MODE == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); }
        }
      }
      // (last half e-tile of conv2 omitted: KS2 = 9 -> f1 of e-tile 4 unused)
      asm volatile("" :: "v"(f1prev));
      xb[0][0] = (__bf16)tacc[0];          // loop-carried dependence so that iterations cannot be merged
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += tacc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, int threads, int blocks, const bf16x8* w, const bf16x8* x, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, w, x, out, cyc, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, w, x, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double m = 0; for (auto v : h) m += (double)v; m /= h.size();
  const int mf = MODE >= 2 ? 19 : 19;
  const double flops = (double)blocks * (threads / 64) * iters * mf * 32768.0;
  printf("%-28s waves/WG %2d blocks %4d: %8.1f cycles/iter  %6.2f cycles/MFMA/wave  wall %.3f ms  %.0f TFLOP/s  clock %.2f GHz\n", name, threads / 64, blocks, m / iters, m / iters / mf, ms, flops / ms * 1e-9, m / (ms * 1e6));
}

int main() {
  bf16x8 *w, *x; float* out; unsigned long long* cyc;
  hipMalloc(&w, 19 * 64 * 16); hipMalloc(&x, 128 * 16); hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 1024 * 8 * 8);
  std::vector<unsigned short> hw(19 * 64 * 8), hx(128 * 8);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3c00 + (i * 7919u) % 512;      // ~0.008..0.03 magnitudes, mixed
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = ((i * 104729u) & 1 ? 0xbf00 : 0x3f00) + (i * 31u) % 128;
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  for (int blocks : {256}) {
    for (int threads : {256, 512, 768, 1024}) {
      run<0>("one MFMA chain", threads, blocks, w, x, out, cyc);
      run<1>("two MFMA chains", threads, blocks, w, x, out, cyc);
      run<2>("conv1-relu-conv2 sched", threads, blocks, w, x, out, cyc);
      run<3>("conv1-relu-conv2 nosched", threads, blocks, w, x, out, cyc);
    }
  }
  return 0;
}
