// Micro-benchmark (diagnostic, not product): what a neighbour halo exchange through global memory costs inside ONE
// launch of 256 co-resident workgroups (one per CU): per iteration every workgroup writes its 12x24 tile (48 B per
// pixel), publishes a flag, waits (BOUNDED spin) for the flags of its up-to-8 neighbours in the image and reads the
// 1-pixel ring around its tile that they wrote.  Reported: cycles per iteration with and without a stand-in for the
// compute between exchanges, and the number of ring pixels that did not carry the neighbour's value.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/halo_xchg tools/ubench/halo_xchg.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int TH = 12, TW = 24, H = 48, W = 48, TY = H / TH, TX = W / TW, TPI = TY * TX, NT = 512;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned tag(int img, int y, int x, int it) { return (unsigned)(((img * 48 + y) * 48 + x) * 131 + it * 7 + 1); }

// MODE 0: agent-scope fence + flag (what the memory model asks for).  MODE 1: write-through stores (nontemporal builtin)
// and plain flag atomics without fences (to see what the fences cost; correctness is then checked, not assumed).
template <int MODE>
__global__ __launch_bounds__(NT) void k(u32x4* ybuf0, u32x4* ybuf1, unsigned* flags, unsigned long long* cyc, unsigned* bad,
                                        unsigned* timeouts, int iters, int work_sleep, int flag_base) {
  __shared__ u32x4 ring[80 * 3];
  __shared__ int give_up;
  const int tile = blockIdx.x, img = tile / TPI, ty = (tile % TPI) / TX, tx = tile % TX, tid = threadIdx.x;
  if (tid == 0) give_up = 0;
  __syncthreads();
  unsigned nbad = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    u32x4* yb = (it & 1) ? ybuf1 : ybuf0;
    for (int s = 0; s < work_sleep; ++s) __builtin_amdgcn_s_sleep(8);                 // stand-in for the block's MFMA phases
    // 1. write the tile: 288 pixels x 3 chunks
    for (int c = tid; c < TH * TW * 3; c += NT) {
      const int p = c / 3, ch = c - p * 3, y = ty * TH + p / TW, x = tx * TW + p % TW;
      const unsigned v = tag(img, y, x, flag_base + it);
      u32x4 val = {v, v + (unsigned)ch, v ^ 0x5a5au, (unsigned)ch};
      u32x4* dst = yb + ((size_t)(img * H + y) * W + x) * 3 + ch;
      if (MODE == 0) *dst = val;
      else if (MODE == 1) __builtin_nontemporal_store(val, dst);
      else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(val) : "memory");   // agent-scope write-through
    }
    // 2. publish
    if (MODE == 0) __threadfence();
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (MODE == 0) __hip_atomic_store(&flags[tile], (unsigned)(flag_base + it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else { if (MODE == 1) __threadfence(); __hip_atomic_store(&flags[tile], (unsigned)(flag_base + it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    // 3. wait for the neighbours (bounded: a workgroup that is not co-resident must not hang the others)
    if (tid < 9 && tid != 4) {
      const int ny = ty + tid / 3 - 1, nx = tx + tid % 3 - 1;
      if (ny >= 0 && ny < TY && nx >= 0 && nx < TX) {
        const unsigned* f = &flags[img * TPI + ny * TX + nx];
        int spins = 0;
        while ((int)(__hip_atomic_load(f, MODE == 0 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)(flag_base + it + 1)) < 0) {
          if (++spins > (1 << 16)) { give_up = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    __syncthreads();
    if (give_up) { if (tid == 0) atomicAdd(timeouts, 1u); break; }
    if (MODE == 0) __threadfence();
    // 4. read the ring (76 pixels x 3 chunks) the neighbours wrote; pixels outside the image are skipped
    for (int c = tid; c < 76 * 3; c += NT) {
      const int p = c / 3, ch = c - p * 3;
      int ry, rx;                                    // ring walk: top row (26), bottom row (26), left col (12), right col (12)
      if (p < 26) { ry = -1; rx = p - 1; } else if (p < 52) { ry = TH; rx = p - 27; } else if (p < 64) { ry = p - 52; rx = -1; } else { ry = p - 64; rx = TW; }
      const int y = ty * TH + ry, x = tx * TW + rx;
      if (y >= 0 && y < H && x >= 0 && x < W) {
        const u32x4* src = yb + ((size_t)(img * H + y) * W + x) * 3 + ch;
        u32x4 v;
        if (MODE == 0) v = *src;
        else if (MODE == 1) v = __builtin_nontemporal_load(src);
        else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
        ring[c] = v;
        if (v.x != tag(img, y, x, flag_base + it) || v.w != (unsigned)ch) ++nbad;
      }
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (nbad) atomicAdd(bad, nbad);
  if (tid == 0) cyc[tile] = t1 - t0;
}

template <int MODE> void run(int work_sleep, u32x4* y0, u32x4* y1, unsigned* flags, unsigned long long* cyc, unsigned* bad, unsigned* to, int& base) {
  const int iters = 64, tiles = 256;
  (void)hipMemset(bad, 0, 4); (void)hipMemset(to, 0, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE>), dim3(tiles), dim3(NT), 0, 0, y0, y1, flags, cyc, bad, to, iters, work_sleep, base);
  (void)hipEventRecord(e1, 0);
  base += iters;
  if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return; }
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(tiles);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  unsigned hb = 0, ht = 0; (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&ht, to, 4, hipMemcpyDeviceToHost);
  double m = 0; for (auto v : h) m += (double)v; m /= h.size();
  printf("mode %d work_sleep %3d: %8.0f memtime ticks (100 MHz) per iteration = %.3f us; wall %.3f us per iteration; wrong ring chunks %u, timeouts %u\n",
         MODE, work_sleep, m / iters, m / iters / 100.0, ms * 1e3 / iters, hb, ht);
}

int main() {
  u32x4 *y0, *y1; unsigned *flags, *bad, *to; unsigned long long* cyc;
  const size_t ybytes = (size_t)32 * H * W * 3 * 16;
  (void)hipMalloc(&y0, ybytes); (void)hipMalloc(&y1, ybytes); (void)hipMalloc(&flags, 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
  (void)hipMalloc(&bad, 4); (void)hipMalloc(&to, 4);
  (void)hipMemset(flags, 0, 256 * 4); (void)hipMemset(y0, 0, ybytes); (void)hipMemset(y1, 0, ybytes);
  int base = 0;
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(0, y0, y1, flags, cyc, bad, to, base);
    //run<0>(5, y0, y1, flags, cyc, bad, to, base);
    //run<0>(10, y0, y1, flags, cyc, bad, to, base);
    run<1>(0, y0, y1, flags, cyc, bad, to, base);
    //run<1>(5, y0, y1, flags, cyc, bad, to, base);
    run<1>(10, y0, y1, flags, cyc, bad, to, base);
    run<2>(0, y0, y1, flags, cyc, bad, to, base);
    run<2>(5, y0, y1, flags, cyc, bad, to, base);
    run<2>(10, y0, y1, flags, cyc, bad, to, base);
  }
  return 0;
}
