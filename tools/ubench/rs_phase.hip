// Micro-benchmark (diagnostic): the real phase functions of csrc/wdsr_fwd_rs.h in a loop, one role at a time, to see
// what a lone A or B wave sustains outside the kernel's staging / barriers.
#include "../../mobilesuperresolution_amd/csrc/wdsr_fwd_rs.h"
#include <cstdio>
#include <vector>
typedef BlockCfg<24, 144, 20> C;
typedef RsCfg<24, 144, 20, 2> R;

template <int ROLE, int NWAVES, int VAR>
__global__ __launch_bounds__(64 * NWAVES) void k(const __bf16* w, const __bf16* x, __bf16* out, unsigned long long* cyc, int iters, int H, int W, int ty0, int tx0) {
  __shared__ __attribute__((aligned(16))) char smem_raw[R::LDS_BYTES];
  __bf16* const X0 = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* const TT = X0 + R::X0_ELEMS;
  __bf16* const X1 = TT + R::TT_ELEMS;
  __bf16* const WL = X1 + R::X1_ELEMS;
  __bf16* const ONES = WL + 2 * R::W_ELEMS;
  float* const CL = reinterpret_cast<float*>(ONES + R::ONES_ELEMS);
  if (threadIdx.x < 8) ONES[threadIdx.x] = threadIdx.x == 0 ? (__bf16)1.f : (__bf16)0.f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < R::X0_ELEMS; i += 64 * NWAVES) X0[i] = x[i % 4096];
  for (int i = tid; i < R::TT_ELEMS; i += 64 * NWAVES) TT[i] = x[(i * 7) % 4096];
  for (int i = tid; i < R::X1_ELEMS; i += 64 * NWAVES) X1[i] = x[(i * 3) % 4096];
  for (int i = tid; i < R::W_ELEMS; i += 64 * NWAVES) WL[i] = w[i];
  if (tid < 64) CL[tid] = 0.01f * tid;
  __syncthreads();
  unsigned long long t0, t1;
  if constexpr (ROLE == 0) {
    RwA<C> rwa;
    rwa.load(WL, lane);
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      rw_phase_a<C, R::KXL, R::rw(1), R::np(1), 1, NWAVES, false>(X1, ONES, TT, rwa, CL, nullptr, H, W, ty0, tx0, wave, lane, [](int, int) {}, [] {});
      if constexpr (VAR & 4) __syncthreads();
    }
    t1 = __builtin_amdgcn_s_memtime();
  } else {
    RwB<C> rwb;
    RwA<C> rwa;
    rwb.load(WL, lane);
    auto pf_a = [&](int lo, int hi) {
#pragma unroll
      for (int i = 0; i < RwA<C>::N; ++i)
        if (i >= lo && i < hi) rwa.load_one(WL, lane, i);
    };
    __bf16* yo = (VAR & 1) ? out + (size_t)blockIdx.x * 48 * 48 * 24 : nullptr;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      if constexpr (VAR & 2) rw_phase_b<C, R::KXL, R::rw(1), R::np(1), 1, NWAVES>(TT, X0, X1, yo, rwb, H, W, ty0, tx0, wave, lane, pf_a);
      else rw_phase_b<C, R::KXL, R::rw(1), R::np(1), 1, NWAVES>(TT, X0, X1, yo, rwb, H, W, ty0, tx0, wave, lane, [](int, int) {});
      if constexpr (VAR & 4) __syncthreads();
    }
    t1 = __builtin_amdgcn_s_memtime();
    if constexpr (VAR & 2) { asm volatile("" :: "v"(rwa.w1[0]), "v"(rwa.w2[8])); }
  }
  if (lane == 0) cyc[blockIdx.x * NWAVES + wave] = t1 - t0;
  out[blockIdx.x * 64 * NWAVES + tid] = TT[tid] + X1[tid];
}

template <int ROLE, int NWAVES, int VAR> void run(const char* name, int blocks, const __bf16* w, const __bf16* x, __bf16* out, unsigned long long* cyc) {
  const int iters = 500;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<ROLE, NWAVES, VAR>), dim3(blocks), dim3(64 * NWAVES), 0, 0, w, x, out, cyc, iters, 48, 48, 12, 24);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", name); return; }
  std::vector<unsigned long long> h(blocks * NWAVES);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  printf("%-22s waves %d blocks %4d: cycles per phase call, per wave:", name, NWAVES, blocks);
  for (int i = 0; i < NWAVES; ++i) printf(" %7.0f", (double)h[i] / iters);
  printf("   (12 tiles: waves 0-3 two, 4-7 one; A tile = 19 MFMA, B tile = 15)\n");
}

int main() {
  __bf16 *w, *x, *out; unsigned long long* cyc;
  (void)hipMalloc(&w, R::W_ELEMS * 2); (void)hipMalloc(&x, 4096 * 2); (void)hipMalloc(&out, (size_t)256 * 48 * 48 * 24 * 2); (void)hipMalloc(&cyc, 1024 * 8 * 8);
  std::vector<unsigned short> hw(R::W_ELEMS), hx(4096);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3c00 + (i * 7919u) % 512;
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = ((i * 104729u) & 1 ? 0xbf00 : 0x3f00) + (i * 31u) % 128;
  (void)hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  for (int blocks : {256}) {
    run<0, 8, 0>("phase A", blocks, w, x, out, cyc);
    run<0, 8, 4>("phase A +barrier", blocks, w, x, out, cyc);
    run<1, 8, 0>("phase B", blocks, w, x, out, cyc);
    run<1, 8, 1>("phase B +global store", blocks, w, x, out, cyc);
    run<1, 8, 2>("phase B +prefetch", blocks, w, x, out, cyc);
    run<1, 8, 4>("phase B +barrier", blocks, w, x, out, cyc);
    run<1, 8, 7>("phase B +all", blocks, w, x, out, cyc);
  }
  return 0;
}
