// Micro-benchmark (diagnostic): what a PERSISTENT single-block forward could sustain per tile: loop { phase A; barrier;
// phase B; barrier } with both phases' weights resident in registers, for (a) one 8-wave workgroup per CU and (b) two
// independent 4-wave workgroups per CU (their phases drift apart, so one's prologue / epilogue / barrier wait sits under
// the other's MFMAs).  No global staging here: it measures the compute schedule only.
#include "../../mobilesuperresolution_amd/csrc/wdsr_fwd_rs.h"
#include <cstdio>
#include <vector>
typedef BlockCfg<24, 144, 20> C;
typedef RsCfg<24, 144, 20, 1> R;

template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void k(const __bf16* w, const __bf16* x, __bf16* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char smem_raw[R::LDS_BYTES];
  __bf16* const X0 = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* const TT = X0 + R::X0_ELEMS;
  __bf16* const WL = TT + R::TT_ELEMS;
  __bf16* const ONES = WL + R::W_ELEMS;
  float* const CL = reinterpret_cast<float*>(ONES + R::ONES_ELEMS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < R::X0_ELEMS; i += 64 * NW) X0[i] = x[i % 4096];
  for (int i = tid; i < R::TT_ELEMS; i += 64 * NW) TT[i] = x[(i * 7) % 4096];
  for (int i = tid; i < R::W_ELEMS; i += 64 * NW) WL[i] = w[i];
  if (tid < 64) CL[tid] = 0.01f * tid;
  if (tid < 8) ONES[tid] = tid == 0 ? (__bf16)1.f : (__bf16)0.f;
  __syncthreads();
  RwA<C> rwa;
  RwB<C> rwb;
  rwa.load(WL, lane);
  rwb.load(WL, lane);
  __bf16* yout = out + (size_t)blockIdx.x * 48 * 48 * 24;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    rw_phase_a<C, R::KXL, R::rw(0), R::np(0), 1, NW, false>(X0, ONES, TT, rwa, CL, nullptr, 48, 48, 12, 24, wave, lane, [](int, int) {}, [] {});
    __syncthreads();
    rw_phase_b<C, R::KXL, C::TW, C::TH * C::TW, 0, NW>(TT, X0, nullptr, yout, rwb, 48, 48, 12, 24, wave, lane, [](int, int) {});
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * NW + wave] = t1 - t0;
}

template <int NW> void run(int blocks, const __bf16* w, const __bf16* x, __bf16* out, unsigned long long* cyc) {
  const int iters = 200;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NW>), dim3(blocks), dim3(64 * NW), 0, 0, w, x, out, cyc, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NW>), dim3(blocks), dim3(64 * NW), 0, 0, w, x, out, cyc, iters);
  (void)hipEventRecord(e1, 0);
  if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return; }
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * NW);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double m = 0; for (auto v : h) m += (double)v; m /= h.size();
  const double tiles = (double)blocks * iters;
  printf("waves/WG %d, %4d workgroups: %7.0f cycles per tile-block per WG; wall %.3f ms -> %.3f us per tile per CU, clock %.2f GHz, "
         "algorithmic %.0f GB/s = %.3f of 8 TB/s\n", NW, blocks, m / iters, ms, ms * 1e3 / (tiles / 256.0), m / (ms * 1e6),
         tiles * 27648.0 / (ms * 1e-3) * 1e-9, tiles * 27648.0 / (ms * 1e-3) / 8e12);
}

int main() {
  __bf16 *w, *x, *out; unsigned long long* cyc;
  (void)hipMalloc(&w, R::W_ELEMS * 2); (void)hipMalloc(&x, 4096 * 2); (void)hipMalloc(&out, (size_t)1024 * 48 * 48 * 24 * 2); (void)hipMalloc(&cyc, 1024 * 8 * 8);
  std::vector<unsigned short> hw(R::W_ELEMS), hx(4096);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3c00 + (i * 7919u) % 512;
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = ((i * 104729u) & 1 ? 0xbf00 : 0x3f00) + (i * 31u) % 128;
  (void)hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  run<8>(256, w, x, out, cyc);
  run<4>(256, w, x, out, cyc);
  run<4>(512, w, x, out, cyc);
  run<6>(512, w, x, out, cyc);
  return 0;
}
