// Micro-benchmark (diagnostic): cost of VALU fillers in the gaps of a dependent v_mfma_f32_32x32x16_bf16 chain, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// KIND 0: none; 1: N x v_cvt_pk_bf16_f32; 2: N x v_pk_max_i16; 3: N x v_add_f32; 4: N x v_cndmask; 5: N x v_mov
template <int KIND, int N>
__global__ __launch_bounds__(256) void k(const bf16x8* w, float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a = w[lane], b = w[64 + lane];
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float f0 = lane, f1 = lane * 2.f, f2 = 1.f, f3 = 3.f;
  unsigned u0 = lane, u1 = lane + 7, u2 = 3, u3 = 9;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < N; ++q) {
        if constexpr (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(f0), "v"(f1));
        if constexpr (KIND == 2) asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(u1) : "v"(u2));
        if constexpr (KIND == 3) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f2) : "v"(f0), "v"(f1));
        if constexpr (KIND == 4) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u3) : "v"(u2), "v"(u1));
        if constexpr (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(u3) : "v"(u2));
        if constexpr (KIND == 6) { if (q & 1) asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(u1) : "v"(u2)); else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(f0), "v"(f1)); }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[threadIdx.x >> 6] = t1 - t0;
  float s = f2 + u0 + u1 + u3;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[threadIdx.x] = s;
}
template <int KIND, int N> void run(const char* name, const bf16x8* w, float* out, unsigned long long* cyc) {
  const int iters = 1000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, N>), dim3(1), dim3(256), 0, 0, w, out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[4];
  (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
  printf("%-18s x%d per gap: %6.2f cycles/MFMA\n", name, N, (double)h[0] / iters / 16);
}
int main() {
  bf16x8* w; float* out; unsigned long long* cyc;
  (void)hipMalloc(&w, 128 * 16); (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 64);
  std::vector<unsigned short> hw(128 * 8);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3c00 + (i * 7919u) % 512;
  (void)hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  run<0, 0>("none", w, out, cyc);
#define R4(K, NAME) run<K, 2>(NAME, w, out, cyc); run<K, 4>(NAME, w, out, cyc); run<K, 5>(NAME, w, out, cyc); run<K, 6>(NAME, w, out, cyc); run<K, 8>(NAME, w, out, cyc);
  R4(1, "v_cvt_pk_bf16_f32") R4(2, "v_pk_max_i16") R4(3, "v_add_f32") R4(4, "v_cndmask_b32") R4(5, "v_mov_b32") R4(6, "cvt/max mix")
  return 0;
}
