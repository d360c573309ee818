// Parameter plumbing kernels: weight-norm (w = g v / ||v||, reference models/basic_wdsr_b.py:23 via
// torch.nn.utils.weight_norm), fragment packing, and their backward.  They turn the model's single
// flat fp32 parameter buffer into the packed MFMA fragment blobs the compute kernels read, and the
// weight-gradient slabs back into a flat gradient, in a handful of tiny launches.
#pragma once
#include "sr_common.h"

SR_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// chan_tab[c] = {v_off, g_off, K, dst_off}; one wave per output channel.
// bias_tab[i] = {src_a, src_b (-1: none), dst}; value = flat[src_a] + flat[src_b] + bias_const[i].
__global__ __launch_bounds__(256) void wn_src_kernel(const float* __restrict__ flat, float* __restrict__ src,
                                                     const int4* __restrict__ chan_tab, int n_chan,
                                                     const int* __restrict__ bias_tab,
                                                     const float* __restrict__ bias_const, int n_bias,
                                                     int chan_blocks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x < chan_blocks) {
    const int c = blockIdx.x * 4 + wave;
    if (c >= n_chan) return;
    const int4 t = chan_tab[c];
    const float* v = flat + t.x;
    float ss = 0.f;
    for (int k = lane; k < t.z; k += 64) { const float a = v[k]; ss += a * a; }
    ss = wave_sum(ss);
    const float scale = flat[t.y] / sqrtf(ss);
    float* d = src + t.w;
    for (int k = lane; k < t.z; k += 64) d[k] = v[k] * scale;
  } else {
    const int i = (blockIdx.x - chan_blocks) * 256 + threadIdx.x;
    if (i >= n_bias) return;
    const int a = bias_tab[3 * i], b = bias_tab[3 * i + 1], d = bias_tab[3 * i + 2];
    src[d] = flat[a] + (b >= 0 ? flat[b] : 0.f) + bias_const[i];
  }
}

// out[rep][i] = (T) src[rep * src_stride + idx[i]]
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                   T* __restrict__ out, int n, long src_stride) {
  const int rep = blockIdx.y;
  const float* s = src + (size_t)rep * src_stride;
  T* o = out + (size_t)rep * n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) o[i] = (T)s[idx[i]];
}

// dsrc[rep * dst_stride + dst[i]] = sum_w partial[(rep * wgs + w) * slab + sidx[i]]
// block = 256 threads = 64 elements x 4 slab groups (a quarter of the workgroup slabs each), LDS-reduced.
__global__ __launch_bounds__(256) void unpack_sum_kernel(const float* __restrict__ partial, int wgs, long slab,
                                                         const int* __restrict__ sidx, const int* __restrict__ dst,
                                                         float* __restrict__ dsrc, int n, long dst_stride) {
  __shared__ float red[4][64];
  const int rep = blockIdx.y, e = threadIdx.x & 63, q = threadIdx.x >> 6;
  const float* p = partial + (size_t)rep * wgs * slab;
  float* d = dsrc + (size_t)rep * dst_stride;
  for (int base = blockIdx.x * 64; base < n; base += gridDim.x * 64) {
    const int i = base + e;
    float acc = 0.f;
    if (i < n) {
      const int s = sidx[i];
#pragma unroll 4
      for (int w = q; w < wgs; w += 4) acc += p[(size_t)w * slab + s];
    }
    red[q][e] = acc;
    __syncthreads();
    if (q == 0 && i < n) d[dst[i]] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
    __syncthreads();
  }
}

// weight-norm backward: dv = (g/n)(dw - v (v.dw)/n^2), dg = (v.dw)/n; biases copy through
__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* __restrict__ flat, const float* __restrict__ dsrc,
                                                     float* __restrict__ gflat, const int4* __restrict__ chan_tab,
                                                     int n_chan, const int* __restrict__ bias_tab, int n_bias,
                                                     int chan_blocks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x < chan_blocks) {
    const int c = blockIdx.x * 4 + wave;
    if (c >= n_chan) return;
    const int4 t = chan_tab[c];
    const float* v = flat + t.x;
    const float* dw = dsrc + t.w;
    float ss = 0.f, dot = 0.f;
    for (int k = lane; k < t.z; k += 64) { const float a = v[k]; ss += a * a; dot += a * dw[k]; }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    const float n = sqrtf(ss), g = flat[t.y];
    const float s1 = g / n, s2 = dot / ss;
    float* dv = gflat + t.x;
    for (int k = lane; k < t.z; k += 64) dv[k] = s1 * (dw[k] - v[k] * s2);
    if (lane == 0) gflat[t.y] = dot / n;
  } else {
    const int i = (blockIdx.x - chan_blocks) * 256 + threadIdx.x;
    if (i >= n_bias) return;
    const int a = bias_tab[3 * i], b = bias_tab[3 * i + 1], d = bias_tab[3 * i + 2];
    const float gval = dsrc[d];
    gflat[a] = gval;
    if (b >= 0) gflat[b] = gval;
  }
}

// ---- fused forms: every packing / slab-gather segment of the network in ONE launch each ----
struct PackSeg { const int* idx; void* out; long src_off, src_stride; int n, reps, as_float, block0; };
struct PackSegs { PackSeg s[4]; int nseg; };

template <typename T>
__global__ __launch_bounds__(256) void pack_all_kernel(const float* __restrict__ src, PackSegs segs) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if (i < segs.nseg && (int)blockIdx.x >= segs.s[i].block0) k = i;
  const PackSeg sg = segs.s[k];
  const int per_rep = (sg.n + 255) / 256;
  const int b = blockIdx.x - sg.block0, rep = b / per_rep, i = (b - rep * per_rep) * 256 + threadIdx.x;
  if (rep >= sg.reps || i >= sg.n) return;
  const float v = src[sg.src_off + (size_t)rep * sg.src_stride + sg.idx[i]];
  if (sg.as_float) reinterpret_cast<float*>(sg.out)[(size_t)rep * sg.n + i] = v;
  else reinterpret_cast<T*>(sg.out)[(size_t)rep * sg.n + i] = (T)v;
}

// (rep_wgs: slabs between two reps' first slab when a segment sums only part of a rep's slabs -- two trunks in one launch; 0 = wgs)
struct UnpackSeg { const float* partial; const int* sidx; const int* dst; long dst_off, dst_stride, slab; int wgs, n, reps, block0, rep_wgs; };
struct UnpackSegs { UnpackSeg s[4]; int nseg; };

constexpr int UNPACK_Q = 16;                             // slab groups per block: 16 partial sums per element
// Elements per block.  Few slabs per element (the body layers: 16): a thread has ONE load per element, so it takes four elements
// (four independent loads in flight; one element per thread was 3 600 blocks of one dependent load each at C2).  Many slabs (tail,
// head: 256): a thread's 16 loads of one element are independent already and the segment has few elements -- 64 per block
// spreads it over four times the CUs.
constexpr int UNPACK_JMAX = 4;
constexpr int unpack_j(int wgs) { return wgs > 4 * UNPACK_Q ? 1 : UNPACK_JMAX; }
constexpr int unpack_blocks(int n, int wgs) { return (n + 64 * unpack_j(wgs) - 1) / (64 * unpack_j(wgs)); }

template <int J>
__device__ __forceinline__ void unpack_body(float* __restrict__ dsrc, const UnpackSeg& sg, float (*red)[64 * UNPACK_JMAX]) {
  constexpr int E = 64 * J;
  const int per_rep = (sg.n + E - 1) / E;
  const int b = blockIdx.x - sg.block0, rep = b / per_rep;
  if (rep >= sg.reps) return;                          // (block-uniform; the launch has exactly reps * per_rep blocks per segment)
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6, i0 = (b - rep * per_rep) * E + e;
  float acc[J];
  const float* p[J];
  const float* const base = sg.partial + (size_t)rep * (sg.rep_wgs ? sg.rep_wgs : sg.wgs) * sg.slab;
  // every load UNCONDITIONAL (an element past the segment's end re-reads the last one and is dropped at the store): a load
  // under a per-lane condition is not hoisted, and the loop then waits for each one before issuing the next
#pragma unroll
  for (int j = 0; j < J; ++j) {
    acc[j] = 0.f;
    p[j] = base + sg.sidx[min(i0 + 64 * j, sg.n - 1)];
  }
  constexpr int U = J == 1 ? 16 : 4;
  int w = q;
  for (; w + (U - 1) * UNPACK_Q < sg.wgs; w += U * UNPACK_Q) {
    float v[U][J];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < J; ++j) v[u][j] = p[j][(size_t)(w + u * UNPACK_Q) * sg.slab];
#pragma unroll
    for (int u = 0; u < U; ++u)                        // (summed in slab order: the order does not depend on U)
#pragma unroll
      for (int j = 0; j < J; ++j) acc[j] += v[u][j];
  }
  for (; w < sg.wgs; w += UNPACK_Q) {
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] += p[j][(size_t)w * sg.slab];
  }
#pragma unroll
  for (int j = 0; j < J; ++j) red[q][e + 64 * j] = acc[j];
  __syncthreads();
  const int t = threadIdx.x;
  const int i = (b - rep * per_rep) * E + t;
  if (t < E && i < sg.n) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < UNPACK_Q; ++j) v += red[j][t];
    dsrc[sg.dst_off + (size_t)rep * sg.dst_stride + sg.dst[i]] = v;
  }
}

__global__ __launch_bounds__(64 * UNPACK_Q) void unpack_all_kernel(float* __restrict__ dsrc, UnpackSegs segs) {
  __shared__ float red[UNPACK_Q][64 * UNPACK_JMAX];
  int k = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if (i < segs.nseg && (int)blockIdx.x >= segs.s[i].block0) k = i;
  const UnpackSeg sg = segs.s[k];
  if (unpack_j(sg.wgs) == 1) unpack_body<1>(dsrc, sg, red);
  else unpack_body<UNPACK_JMAX>(dsrc, sg, red);
}
