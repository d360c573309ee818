// Device-side building blocks shared by every kernel of the SR hot path (gfx950 / CDNA4 only).
//
// Conventions (wave64, lane l: r = l & 31, hh = l >> 5), all on v_mfma_f32_32x32x16_bf16 or, in the
// exact-fp32 parity mode, on 8 x v_mfma_f32_32x32x2_f32 per 16-deep k-step:
//   * a "fragment" is 8 elements per lane.  As the A operand lane (r,hh) holds A[row r][k = 8*hh + j],
//     as the B operand it holds B[k = 8*hh + j][col r], j = 0..7.  The fp32 form keeps the same
//     (lane, j) -> (row|col, k) meaning; MFMA #j contracts k in {j, 8 + j}.
//   * an accumulator tile is 32x32 fp32: reg i of lane (r,hh) holds D[row (i&3) + 8*(i>>2) + 4*hh][col r].
//   * an accumulator tile used as the next product's B operand ("chained" k order): k-step s takes
//     regs 8s..8s+7, i.e. element j of lane half hh is row 16s + 8*(j>>2) + 4*hh + (j&3); the other
//     operand is packed on the host in that same order (mobilesuperresolution_amd/packing.py).
//   * swapping the two operand registers of an MFMA yields the transposed tile, so the same packed
//     weight fragments serve "channels in rows, pixels on lanes" and "pixels in rows, channels on lanes".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define SR_DEV __device__ __forceinline__

// In-kernel time stamps exist only in the diagnostic build (-DSR_DEBUG_STAMPS -> libsr_hotpath_dbg.so, see
// build.py); in the product library the macros are empty and no kernel carries a stamp pointer or a stamp load.
// Layout: buf[workgroup][16 waves][16 stamps][2] = {s_memrealtime (100 MHz), s_memtime (shader clock)}, written by
// lane 0 of every wave: the pair gives both the wall time of a phase and the clock the chip held during it.
#ifdef SR_DEBUG_STAMPS
__device__ unsigned long long* g_sr_stamps = nullptr;
__device__ unsigned long long g_sr_stamp_wgs = 0;      // workgroups the buffer has room for: launches with more do not stamp past it
#define SR_STAMP_DECL int stamp_i_ = 0
#define SR_STAMP() do { unsigned long long* sp_ = g_sr_stamps; const size_t wg_ = (size_t)blockIdx.y * gridDim.x + blockIdx.x; \
  if (sp_ && wg_ < g_sr_stamp_wgs && (threadIdx.x & 63) == 0 && stamp_i_ < 16 && (threadIdx.x >> 6) < 16) { \
    unsigned long long* q_ = sp_ + ((wg_ * 16 + (threadIdx.x >> 6)) * 16 + stamp_i_) * 2; \
    q_[0] = __builtin_amdgcn_s_memrealtime(); q_[1] = __builtin_amdgcn_s_memtime(); } ++stamp_i_; } while (0)
// a stamp with an explicit slot, for device functions that have no running stamp index (inner phases; diagnostic build only)
#define SR_STAMP_AT(i_) do { unsigned long long* sp_ = g_sr_stamps; const size_t wg_ = (size_t)blockIdx.y * gridDim.x + blockIdx.x; \
  if (sp_ && wg_ < g_sr_stamp_wgs && (threadIdx.x & 63) == 0 && (i_) < 16 && (threadIdx.x >> 6) < 16) { \
    unsigned long long* q_ = sp_ + ((wg_ * 16 + (threadIdx.x >> 6)) * 16 + (i_)) * 2; \
    q_[0] = __builtin_amdgcn_s_memrealtime(); q_[1] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define SR_STAMP_DECL do {} while (0)
#define SR_STAMP() do {} while (0)
#define SR_STAMP_AT(i_) do {} while (0)
#endif

template <typename T> struct FragOf;
template <> struct FragOf<__bf16> { typedef bf16x8 type; typedef bf16x4 half_type; };
template <> struct FragOf<float>  { typedef f32x8 type;  typedef f32x4 half_type; };

// D = A*B + C over one 16-deep k-step.
template <typename T> SR_DEV f32x16 mma16(typename FragOf<T>::type a, typename FragOf<T>::type b, f32x16 c);
template <> SR_DEV f32x16 mma16<__bf16>(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <> SR_DEV f32x16 mma16<float>(f32x8 a, f32x8 b, f32x16 c) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
  return c;
}

// two fp32 -> one packed pair.  bf16: ONE v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN kept); element-wise
// casts make hipcc emit one conversion per value plus a v_perm_b32 per pair (3 VALU ops instead of 1).
template <typename T> SR_DEV void cvt_pair(T& lo, T& hi, float a, float b) {
  if constexpr (sizeof(T) == 2) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    const f32x2_ v = {a, b};
    const bf16x2_ p = __builtin_convertvector(v, bf16x2_);
    lo = p[0];
    hi = p[1];
  } else {
    lo = a;
    hi = b;
  }
}

// accumulator regs 8s..8s+7 -> fragment (chained k order)
template <typename T, int S> SR_DEV typename FragOf<T>::type acc_to_frag(const f32x16& acc) {
  typename FragOf<T>::type f;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    T lo, hi;
    cvt_pair<T>(lo, hi, acc[8 * S + j], acc[8 * S + j + 1]);
    f[j] = lo;
    f[j + 1] = hi;
  }
  return f;
}

// relu(accumulator regs 8s..8s+7) as a fragment.  bf16: convert first, then a packed signed-integer max
// with 0 on the bf16 bit patterns (sign-magnitude: negative <=> negative int16; -0 -> +0): half the VALU
// ops of fmaxf on fp32, and none of the canonicalising v_max hipcc puts in front of fmaxf on MFMA output.
template <typename T, int S> SR_DEV typename FragOf<T>::type acc_to_frag_relu(const f32x16& acc) {
  if constexpr (sizeof(T) == 2) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const bf16x8 f = acc_to_frag<T, S>(acc);
    s16x8 v = __builtin_bit_cast(s16x8, f);
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    v = __builtin_elementwise_max(v, z);
    return __builtin_bit_cast(bf16x8, v);
  } else {
    typename FragOf<T>::type f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = fmaxf(acc[8 * S + j], 0.f);
    return f;
  }
}

// accumulator regs 4g..4g+3 (rows 8g + 4hh + 0..3) -> 4 packed elements
template <typename T> SR_DEV typename FragOf<T>::half_type acc_group(const f32x16& acc, int g) {
  typename FragOf<T>::half_type v;
#pragma unroll
  for (int j = 0; j < 4; j += 2) {
    T lo, hi;
    cvt_pair<T>(lo, hi, acc[4 * g + j], acc[4 * g + j + 1]);
    v[j] = lo;
    v[j + 1] = hi;
  }
  return v;
}

// activation stores bypass the per-XCD L2 write-back path: nothing of ours re-reads them before the kernel
// ends, and the end-of-kernel release then has no dirty lines left to flush
template <typename V> SR_DEV void stream_store(V* p, const V& v) { __builtin_nontemporal_store(v, p); }

SR_DEV f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// C-init table: float[2][16] per tile (lane half hh, reg i) -> accumulator
SR_DEV f32x16 load_cinit(const float* __restrict__ tab, int hh) {
  const f32x4* p = reinterpret_cast<const f32x4*>(tab + hh * 16);
  f32x16 c;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = p[q];
    c[4 * q + 0] = v[0]; c[4 * q + 1] = v[1]; c[4 * q + 2] = v[2]; c[4 * q + 3] = v[3];
  }
  return c;
}

// packed weight fragment #idx (64 lanes x 8 elements, lane-major) from global memory
template <typename T> SR_DEV typename FragOf<T>::type load_wfrag(const T* __restrict__ w, int idx, int lane) {
  return *reinterpret_cast<const typename FragOf<T>::type*>(w + ((size_t)idx * 64 + lane) * 8);
}

// Weight fragments are loop-invariant, so hipcc hoists their loads out of the tile loop until it runs
// out of registers and spills (and ROCm 7.2 miscompiles a partially spilled fragment: the tail dword
// parked in an AGPR is never copied back).  Where the fragments do not fit, launder the base pointer
// once per tile so the loads stay inside the loop (they hit L1/L2).  The build rejects any spill.
template <bool HOIST, typename T> SR_DEV const T* weights_for_tile(const T* w) {
  if constexpr (!HOIST) {
    int z = 0;
    asm volatile("" : "+v"(z));
    return w + z;
  } else {
    return w;
  }
}

// Where a kernel's packed weight fragments come from: an LDS copy staged once per workgroup (bf16 mode:
// one conflict-free ds_read_b128 per fragment) or global memory (fp32 parity mode: the fragments are
// twice as large and speed is not the point), re-laundered per tile so the loads are never hoisted.
template <typename T, bool INLDS> struct WSrc;
template <typename T> struct WSrc<T, true> {
  const T* p;
  SR_DEV void tile() {}
  SR_DEV typename FragOf<T>::type get(int idx, int lane) const {
    return *reinterpret_cast<const typename FragOf<T>::type*>(p + (idx * 64 + lane) * 8);
  }
};
template <typename T> struct WSrc<T, false> {
  const T* p0;
  const T* p;
  SR_DEV void tile() { p = weights_for_tile<false>(p0); }
  SR_DEV typename FragOf<T>::type get(int idx, int lane) const { return load_wfrag<T>(p, idx, lane); }
};

// copy `nfrag` packed fragments (512 elements each) global -> LDS by LDS-DMA (global_load_lds_dwordx4: one
// wave-instruction moves 64 lanes x 16 B to M0 + lane * 16, exactly the lane-major fragment layout).  No
// registers, no wait: every piece is in flight at once; the data is readable after the issuing wave's
// s_waitcnt vmcnt and a workgroup barrier (__syncthreads() emits both).  A register-staged copy loop compiles
// to load / vmcnt(0) / ds_write per iteration, i.e. one L2 round trip per 16 bytes per thread.
template <typename T, int NTHREADS> SR_DEV void stage_weights(T* dst, const T* __restrict__ src, int nfrag, int tid) {
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr int NW = NTHREADS / 64, PIECES = (int)sizeof(T) * 512 / 1024;   // 1 KiB pieces per fragment
  const int lane = tid & 63, wave = tid >> 6;
  const char* s = reinterpret_cast<const char*>(src);
  char* d = reinterpret_cast<char*>(dst);
  for (int p = wave; p < nfrag * PIECES; p += NW)
    __builtin_amdgcn_global_load_lds((gptr_t)(s + (size_t)p * 1024 + lane * 16), (lptr_t)(d + p * 1024), 16, 0, 0);
}

// 8 consecutive elements from an LDS image (16-byte aligned element offset)
template <typename T> SR_DEV typename FragOf<T>::type lds_chunk(const T* img, int elem_off) {
  return *reinterpret_cast<const typename FragOf<T>::type*>(img + elem_off);
}

// ---- transposed fragment: element j = img[px(j)][ch] for 8 pixel rows and this lane's channel ----
// bf16: ds_read_b64_tr_b16.  Within each 16-lane group the instruction reads a 4-row x 16-column block:
// lane 4q+p of the group supplies the address of row q, columns 4p..4p+3; lane i receives column i of
// the 4 rows.  Our lane (r,hh): group = r>>4 picks channels 16*(r>>4) .. +15; rows are 4 pixels.
// `rowaddr_lo/hi` = element offsets of THIS lane's supplied row (q = (lane>>2)&3) for the two
// 4-pixel groups, pointing at channel 16*(r>>4) + 4*(lane&3).
SR_DEV bf16x8 lds_tr_frag(const __bf16* img, int rowaddr_lo, int rowaddr_hi) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + rowaddr_lo));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + rowaddr_hi));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

template <typename T> SR_DEV T relu(T x) { return x > (T)0 ? x : (T)0; }

#define SR_HIP_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

// Transposed fragment for a pixel-contraction (weight-gradient) product: element j of lane (r,hh) =
// img[rowbase(p_j) + r] with p_j = 16 s + 8 (j>>2) + 4 hh + (j&3) (chained order), i.e. this lane's
// channel r for 8 pixel rows.  bf16: two ds_read_b64_tr_b16 (EXEC must be all ones); fp32: 8 ds_read_b32.
template <typename T, typename RowBase>
SR_DEV typename FragOf<T>::type tr_frag(const T* img, int s, int lane, RowBase rowbase) {
  typename FragOf<T>::type f;
  const int hh = lane >> 5;
  if constexpr (sizeof(T) == 2) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int q = (lane >> 2) & 3, p = lane & 3, grp = (lane >> 4) & 1;
    const int plo = 16 * s + 4 * hh + q;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + rowbase(plo) + 16 * grp + 4 * p));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + rowbase(plo + 8) + 16 * grp + 4 * p));
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  } else {
    const int r = lane & 31;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = img[rowbase(16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)) + r];
  }
  return f;
}

// sum over the 32 lanes of each wave half (lanes that share hh); every lane of the half gets the total
SR_DEV float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// 32 per-lane partial sums (a[0..15], b[0..15]) -> their totals over the 32 lanes of each wave half, one per lane:
// lane r of a half ends up with total #r (r < 16: a[r], else b[r - 16]).  A halving butterfly: at distance 16, 8, .. 1
// a lane keeps the half of the values its bit selects and hands the other half to its partner: 31 exchanges instead
// of the 160 of 32 separate half_sum() calls (each a ds_bpermute, ~5 us per wave at the end of a kernel).
template <int N> SR_DEV void half_sum32_step(float (&v)[32], int r) {
  const bool up = (r & N) != 0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const float keep = up ? v[j + N] : v[j], send = up ? v[j] : v[j + N];
    v[j] = keep + __shfl_xor(send, N);
  }
}
SR_DEV float half_sum32(const float (&a)[16], const float (&b)[16], int r) {
  float v[32];
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = a[i]; v[16 + i] = b[i]; }
  half_sum32_step<16>(v, r);
  half_sum32_step<8>(v, r);
  half_sum32_step<4>(v, r);
  half_sum32_step<2>(v, r);
  half_sum32_step<1>(v, r);
  return v[0];
}

// accumulator tile -> [reg i][lane] floats of a slab (LDS or global), plain stores.  LDS float atomics
// (ds_add_f32) measured ~1500 cycles per wave-instruction on gfx950: never reduce through them.
SR_DEV void slab_store_tile(float* slab, int tile, const f32x16& acc, int lane) {
#pragma unroll
  for (int i = 0; i < 16; ++i) slab[(tile * 16 + i) * 64 + lane] = acc[i];
}

// accumulator tile -> [reg i][lane] floats in an LDS slab (atomic add: several waves share the slab)
SR_DEV void slab_add_tile(float* slab, int tile, const f32x16& acc, int lane) {
#pragma unroll
  for (int i = 0; i < 16; ++i) atomicAdd(slab + (tile * 16 + i) * 64 + lane, acc[i]);
}
