// C-ABI entry points of libsr_hotpath.so (declared in include/sr_hotpath.h).
#include <algorithm>
#include <cstdlib>
#include "../../include/sr_hotpath.h"
#include "wdsr_block.h"
#include "wdsr_fwd_rs.h"
#include "wdsr_fwd_stream.h"
#include "wdsr_bwd_rs.h"
#include "wdsr_bwd_pair_lds.h"
#include "wdsr_wgrad_rs.h"
#include "wdsr_ends.h"
#include "wdsr_prep.h"
#include "conv3x3.h"
#include "spynet_conv.h"
#include "nas_block.h"
#include "nas_dw_lc.h"
#include "nas_bwd_fused.h"
#include "flow_warp.h"
#include "metrics.h"
#include "patches.h"
#include "train_step.h"
#include "pixel_shuffle.h"

extern "C" int sr_abi_version(void) { return 13; }

// A/B switches between a kernel and the one it replaced are live in the diagnostic build only (build.py --debug); in the product
// library they are the constant false, and the kernels only they reach are not instantiated.  (SR_NAS_FWD_SPLIT / SR_NAS_BWD_SPLIT /
// SR_C3_ONE_BLOCK_PER_LAUNCH stay: the parity tests chain the fused kernels to the separately tested ones through them.)
#ifdef SR_DEBUG_STAMPS
#define SR_AB(name) (getenv(name) != nullptr)
#else
#define SR_AB(name) false
#endif

namespace {

template <typename T, int F, int E, int L>
int launch_block_fwd(const void* x, void* y, const void* wblob, const float* cinit, int N, int H, int W,
                     hipStream_t st, void* tsave = nullptr) {
  typedef BlockCfg<F, E, L> C;
  const int tiles_x = (W + C::TW - 1) / C::TW, tiles_y = (H + C::TH - 1) / C::TH;
  dim3 grid(tiles_x * tiles_y, N), block(64 * C::NPT_H);
  hipLaunchKernelGGL((wdsr_block_fwd_kernel<T, F, E, L>), grid, block, 0, st, (const T*)x, (T*)y, (const T*)wblob,
                     cinit, H, W, tiles_x, (T*)tsave);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

template <typename T, int F, int E, int L>
int launch_block_bwd_data(const void* x, const void* dy, void* dx, const void* wblob, const float* cinit, int N,
                          int H, int W, hipStream_t st, void* dtsave = nullptr) {
  typedef BlockCfg<F, E, L> C;
  const int tiles_x = (W + C::TW - 1) / C::TW, tiles_y = (H + C::TH - 1) / C::TH;
  dim3 grid(tiles_x * tiles_y, N), block(64 * C::NPT_O);
  hipLaunchKernelGGL((wdsr_block_bwd_data_kernel<T, F, E, L>), grid, block, 0, st, (const T*)x, (const T*)dy, (T*)dx,
                     (const T*)wblob, cinit, H, W, tiles_x, (T*)dtsave);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

template <typename T, int F, int E, int L>
int launch_block_wgrad(const void* x, const void* dy, const void* wblob, const float* cinit, float* pa, float* pb,
                       int layers, int wgs, int N, int H, int W, long x_ls, long dy_ls, long w_ls, long c_ls,
                       hipStream_t st) {
  typedef BlockCfg<F, E, L> C;
  const int tiles_x = (W + C::TW - 1) / C::TW, tiles_y = (H + C::TH - 1) / C::TH;
  dim3 grid(wgs, layers);
  hipLaunchKernelGGL((wdsr_block_wgrad_kernel<T, F, E, L, 0>), grid, dim3(64 * WgradCfg<F, E, L, 0>::NWAVES), 0, st,
                     (const T*)x, (const T*)dy, (const T*)wblob, cinit, pa, N, H, W, tiles_x, tiles_x * tiles_y, x_ls,
                     dy_ls, w_ls, c_ls);
  hipLaunchKernelGGL((wdsr_block_wgrad_kernel<T, F, E, L, 1>), grid, dim3(64 * WgradCfg<F, E, L, 1>::NWAVES), 0, st,
                     (const T*)x, (const T*)dy, (const T*)wblob, cinit, pb, N, H, W, tiles_x, tiles_x * tiles_y, x_ls,
                     dy_ls, w_ls, c_ls);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// one image per workgroup pays from one image per CU on, when the last round of workgroups is at least ~70 % full
static bool fwd_stream_applies(long N, int H, int W) {
  if (W != 48 || H % 4 != 0 || H < 8 || H > 4096 || N < 256) return false;
  const long rounds = (N + 255) / 256;
  return N * 10 >= rounds * 256 * 7;
}

// the sixteen-wave form (weights read from LDS at use): the route of the 32-unit network, whose weight sets do not fit the
// register-resident kernels; at 24 units in variant / diagnostic builds only (A/B timing)
template <int F, int E, int L, int NBLK>
static int launch_fwd_rs16(const void* x, void* ya, void* yb, const void* wa, const void* wb, const float* cia, const float* cib,
                           void* tsa, void* tsb, int N, int H, int W, hipStream_t st) {
  typedef BlockCfg<F, E, L> C;
  typedef __bf16 T;
  const int tiles_x = (W + C::TW - 1) / C::TW, tiles_y = (H + C::TH - 1) / C::TH;
  if (tsa && (NBLK == 1 || tsb))
    hipLaunchKernelGGL((wdsr_fwd_rs16_kernel<F, E, L, NBLK, true>), dim3(tiles_x * tiles_y, N), dim3(1024), 0, st, (const T*)x, (T*)ya,
                       (T*)yb, (const T*)wa, (const T*)wb, cia, cib, (T*)tsa, (T*)tsb, H, W, tiles_x);
  else
    hipLaunchKernelGGL((wdsr_fwd_rs16_kernel<F, E, L, NBLK, false>), dim3(tiles_x * tiles_y, N), dim3(1024), 0, st, (const T*)x, (T*)ya,
                       (T*)yb, (const T*)wa, (const T*)wb, cia, cib, (T*)tsa, (T*)tsb, H, W, tiles_x);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

// forward with register-resident weights (csrc/wdsr_fwd_rs.h): nblk = 1 (x -> yb) or 2 (x -> ya -> yb)
template <int F, int E, int L, int NBLK>
static int launch_fwd_rs(const void* x, void* ya, void* yb, const void* wa, const void* wb, const float* cia, const float* cib,
                         void* tsa, void* tsb, int N, int H, int W, hipStream_t st) {
  typedef BlockCfg<F, E, L> C;
  typedef __bf16 T;
  const int tiles_x = (W + C::TW - 1) / C::TW, tiles_y = (H + C::TH - 1) / C::TH;
  constexpr int persist_from = 768;                    // measured crossover: 3 tiles per CU
  const long total = (long)N * tiles_x * tiles_y;
  if constexpr (NBLK == 2) {
    // whole 48-wide images, at least one per CU and the last round of workgroups reasonably full: the streaming kernel
    // (csrc/wdsr_fwd_stream.h: no halo recompute, one barrier per band), with or without the saved t images
    if (fwd_stream_applies(N, H, W)) {
#if defined(SR_FORCE_STREAM8) || defined(SR_DEBUG_STAMPS)   // (variant / diagnostic builds: the eight-wave form, one block's whole weight set per wave)
#ifdef SR_FORCE_STREAM8
      static const bool eight = true;
#else
      static const bool eight = SR_AB("SR_STREAM8");
#endif
      if (eight) {
        if (tsa && tsb)
          hipLaunchKernelGGL((wdsr_fwd_stream_kernel<F, E, L, true>), dim3(256), dim3(512), 0, st, (const T*)x, (T*)ya, (T*)yb, (const T*)wa,
                             (const T*)wb, cia, cib, (T*)tsa, (T*)tsb, N, H);
        else
          hipLaunchKernelGGL((wdsr_fwd_stream_kernel<F, E, L, false>), dim3(256), dim3(512), 0, st, (const T*)x, (T*)ya, (T*)yb, (const T*)wa,
                             (const T*)wb, cia, cib, (T*)nullptr, (T*)nullptr, N, H);
        SR_HIP_CHECK_LAUNCH();
        return 0;
      }
#endif
      if (tsa && tsb)
        hipLaunchKernelGGL((wdsr_fwd_stream12_kernel<F, E, L, true>), dim3(256), dim3(768), 0, st, (const T*)x, (T*)ya, (T*)yb, (const T*)wa,
                           (const T*)wb, cia, cib, (T*)tsa, (T*)tsb, N, H);
      else
        hipLaunchKernelGGL((wdsr_fwd_stream12_kernel<F, E, L, false>), dim3(256), dim3(768), 0, st, (const T*)x, (T*)ya, (T*)yb, (const T*)wa,
                           (const T*)wb, cia, cib, (T*)nullptr, (T*)nullptr, N, H);
      SR_HIP_CHECK_LAUNCH();
      return 0;
    }
  }
  if (total >= persist_from && total < (1L << 31)) {   // many tiles per CU: the persistent form (one workgroup per CU)
    const int wgs = 256;
    const bool save = tsa && (NBLK == 1 || tsb);
    if (save && NBLK == 1) {
      hipLaunchKernelGGL((wdsr_fwd_rs_persist_kernel<F, E, L, 1, true>), dim3(wgs), dim3(512), 0, st, (const T*)x, (T*)ya, (T*)yb,
                         (const T*)wa, (const T*)wb, cia, cib, (T*)tsa, (T*)tsb, N, H, W, tiles_x, tiles_x * tiles_y);
      SR_HIP_CHECK_LAUNCH();
      return 0;
    }
    if constexpr (NBLK == 2) {
      if (!save) {                                     // two blocks, nothing saved: the two-role pipeline
        hipLaunchKernelGGL((wdsr_fwd_rs_pipe_kernel<F, E, L>), dim3(wgs), dim3(512), 0, st, (const T*)x, (T*)ya, (T*)yb, (const T*)wa,
                           (const T*)wb, cia, cib, N, H, W, tiles_x, tiles_x * tiles_y);
        SR_HIP_CHECK_LAUNCH();
        return 0;
      }
    } else if (!save) {
      hipLaunchKernelGGL((wdsr_fwd_rs_persist_kernel<F, E, L, 1, false>), dim3(wgs), dim3(512), 0, st, (const T*)x, (T*)ya, (T*)yb,
                         (const T*)wa, (const T*)wb, cia, cib, (T*)tsa, (T*)tsb, N, H, W, tiles_x, tiles_x * tiles_y);
      SR_HIP_CHECK_LAUNCH();
      return 0;
    }
  }
#if defined(SR_FORCE_RS16) || defined(SR_DEBUG_STAMPS)  // (tools/build_variant.sh / diagnostic build: the sixteen-wave form, A/B timing only)
#ifdef SR_FORCE_RS16
  static const bool rs16 = true;
#else
  static const bool rs16 = SR_AB("SR_RS16");
#endif
  if (rs16) return launch_fwd_rs16<F, E, L, NBLK>(x, ya, yb, wa, wb, cia, cib, tsa, tsb, N, H, W, st);
#endif
  if (tsa && (NBLK == 1 || tsb))
    hipLaunchKernelGGL((wdsr_fwd_rs_kernel<F, E, L, NBLK, true>), dim3(tiles_x * tiles_y, N), dim3(512), 0, st, (const T*)x, (T*)ya,
                       (T*)yb, (const T*)wa, (const T*)wb, cia, cib, (T*)tsa, (T*)tsb, H, W, tiles_x);
  else
    hipLaunchKernelGGL((wdsr_fwd_rs_kernel<F, E, L, NBLK, false>), dim3(tiles_x * tiles_y, N), dim3(512), 0, st, (const T*)x, (T*)ya,
                       (T*)yb, (const T*)wa, (const T*)wb, cia, cib, (T*)tsa, (T*)tsb, H, W, tiles_x);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
extern "C" int sr_wdsr_fwd_rs(const void* x, void* ya, void* yb, const void* wa, const void* wb, const float* cia,
                              const float* cib, void* tsa, void* tsb, int nblk, int N, int H, int W, int F, int dtype,
                              sr_stream_t stream) {
  if (!x || !yb || !wa || !cia || N <= 0 || H <= 0 || W <= 0 || N > 65535 || (nblk == 2 && (!wb || !cib))) return -2;
  if (dtype != SR_DTYPE_BF16) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (F == 24 && nblk == 2) return launch_fwd_rs<24, 144, 20, 2>(x, ya, yb, wa, wb, cia, cib, tsa, tsb, N, H, W, st);
  if (F == 24 && nblk == 1) return launch_fwd_rs<24, 144, 20, 1>(x, nullptr, yb, wa, nullptr, cia, nullptr, tsa, nullptr, N, H, W, st);
  if (F == 32 && nblk == 2) return launch_fwd_rs16<32, 192, 26, 2>(x, ya, yb, wa, wb, cia, cib, tsa, tsb, N, H, W, st);
  if (F == 32 && nblk == 1) return launch_fwd_rs16<32, 192, 26, 1>(x, nullptr, yb, wa, nullptr, cia, nullptr, tsa, nullptr, N, H, W, st);
  return -1;
}
extern "C" int sr_wdsr_block2_fwd(const void* x, void* ya, void* yb, const void* wa, const void* wb, const float* cia,
                                  const float* cib, void* tsa, void* tsb, int N, int H, int W, int F, int dtype,
                                  sr_stream_t stream) {
  if (F != 24 && F != 32) return (!x || !yb || !wa || !wb || !cia || !cib || N <= 0 || H <= 0 || W <= 0 || N > 65535) ? -2 : -1;
  return sr_wdsr_fwd_rs(x, ya, yb, wa, wb, cia, cib, tsa, tsb, 2, N, H, W, F, dtype, stream);
}
extern "C" int sr_wdsr_fwd_rs_repeat(void* x, void* ya, void* yb, const void* wa, const void* wb, const float* cia,
                                     const float* cib, void* tsa, void* tsb, int nblk, int N, int H, int W, int F, int dtype,
                                     int reps, sr_stream_t stream) {
  for (int i = 0; i < reps; ++i) {
    const int rc = (i & 1) ? sr_wdsr_fwd_rs(yb, ya, x, wa, wb, cia, cib, tsa, tsb, nblk, N, H, W, F, dtype, stream)
                           : sr_wdsr_fwd_rs(x, ya, yb, wa, wb, cia, cib, tsa, tsb, nblk, N, H, W, F, dtype, stream);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int sr_wdsr_block_fwd_repeat(void* x, void* y, const void* wblob, const float* cinit, int N, int H, int W,
                                        int F, int dtype, int reps, sr_stream_t stream) {
  for (int i = 0; i < reps; ++i) {
    const int rc = (i & 1) ? sr_wdsr_block_fwd(y, x, wblob, cinit, N, H, W, F, dtype, stream)
                           : sr_wdsr_block_fwd(x, y, wblob, cinit, N, H, W, F, dtype, stream);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int sr_wdsr_block2_fwd_repeat(void* x, void* ya, void* yb, const void* wa, const void* wb, const float* cia,
                                         const float* cib, int N, int H, int W, int F, int dtype, int reps,
                                         sr_stream_t stream) {
  for (int i = 0; i < reps; ++i) {
    const int rc = (i & 1) ? sr_wdsr_block2_fwd(yb, ya, x, wa, wb, cia, cib, nullptr, nullptr, N, H, W, F, dtype, stream)
                           : sr_wdsr_block2_fwd(x, ya, yb, wa, wb, cia, cib, nullptr, nullptr, N, H, W, F, dtype, stream);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int sr_wdsr_block_bwd_data(const void* x, const void* dy, void* dx, const void* wblob,
                                      const float* cinit, int N, int H, int W, int F, int dtype,
                                      sr_stream_t stream) {
  if (!x || !dy || !dx || !wblob || !cinit || N <= 0 || H <= 0 || W <= 0 || N > 65535) return -2;
  hipStream_t st = (hipStream_t)stream;
  if (F == 24 && dtype == SR_DTYPE_BF16) return launch_block_bwd_data<__bf16, 24, 144, 20>(x, dy, dx, wblob, cinit, N, H, W, st);
  if (F == 24 && dtype == SR_DTYPE_F32) return launch_block_bwd_data<float, 24, 144, 20>(x, dy, dx, wblob, cinit, N, H, W, st);
  if (F == 32 && dtype == SR_DTYPE_BF16) return launch_block_bwd_data<__bf16, 32, 192, 26>(x, dy, dx, wblob, cinit, N, H, W, st);
  if (F == 32 && dtype == SR_DTYPE_F32) return launch_block_bwd_data<float, 32, 192, 26>(x, dy, dx, wblob, cinit, N, H, W, st);
  return -1;
}

extern "C" int sr_wdsr_block2_bwd_data(const void* xa, const void* xb, const void* dyb, void* dxb, void* dxa,
                                       const void* wa, const void* wb, const float* cia, const float* cib, void* dta,
                                       void* dtb, int N, int H, int W, int F, int dtype, sr_stream_t stream) {
  if (!xa || !xb || !dyb || !dxb || !dxa || !wa || !wb || !cia || !cib || N <= 0 || H <= 0 || W <= 0 || N > 65535)
    return -2;
  if ((F != 24 && F != 32) || dtype != SR_DTYPE_BF16) return -1;
  typedef BlockCfg<24, 144, 20> C;
  const int tiles_x = (W + C::TW - 1) / C::TW, tiles_y = (H + C::TH - 1) / C::TH;
  if (F == 32) {                                       // 32 units: twelve waves, fragments from LDS at use (csrc/wdsr_bwd_pair_lds.h)
    typedef BwdPairCfg<32, 192, 26> R;
    hipLaunchKernelGGL((wdsr_bwd_pair_lds_kernel<32, 192, 26>), dim3(tiles_x * tiles_y, N), dim3(R::NTHREADS), 0, (hipStream_t)stream,
                       (const __bf16*)xa, (const __bf16*)xb, (const __bf16*)dyb, (__bf16*)dxb, (__bf16*)dxa, (const __bf16*)wa,
                       (const __bf16*)wb, cia, cib, (__bf16*)dta, (__bf16*)dtb, H, W, tiles_x);
    SR_HIP_CHECK_LAUNCH();
    return 0;
  }
  if (!SR_AB("SR_BWD2_OLD")) {                         // round 3: 8 waves, register-resident weights (csrc/wdsr_bwd_rs.h)
    hipLaunchKernelGGL((wdsr_bwd_rs_kernel<24, 144, 20>), dim3(tiles_x * tiles_y, N), dim3(512), 0, (hipStream_t)stream, (const __bf16*)xa,
                       (const __bf16*)xb, (const __bf16*)dyb, (__bf16*)dxb, (__bf16*)dxa, (const __bf16*)wa, (const __bf16*)wb, (__bf16*)dta,
                       (__bf16*)dtb, H, W, tiles_x);
    SR_HIP_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL((wdsr_block2_bwd_data_kernel<__bf16, 24, 144, 20>), dim3(tiles_x * tiles_y, N),
                     dim3(64 * C::NPT_H), 0, (hipStream_t)stream, (const __bf16*)xa, (const __bf16*)xb,
                     (const __bf16*)dyb, (__bf16*)dxb, (__bf16*)dxa, (const __bf16*)wa, (const __bf16*)wb, cia, cib,
                     (__bf16*)dta, (__bf16*)dtb, H, W, tiles_x);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

extern "C" int sr_wdsr_block_wgrad(const void* x, const void* dy, const void* wblob, const float* cinit,
                                   float* pa, float* pb, int layers, int wgs, int N, int H, int W, int F,
                                   int dtype, long x_ls, long dy_ls, long w_ls, long c_ls, sr_stream_t stream) {
  if (!x || !dy || !wblob || !cinit || !pa || !pb || layers <= 0 || wgs <= 0 || N <= 0 || H <= 0 || W <= 0 ||
      layers > 65535)
    return -2;
  hipStream_t st = (hipStream_t)stream;
#define SR_WG(T, F_, E_, L_) launch_block_wgrad<T, F_, E_, L_>(x, dy, wblob, cinit, pa, pb, layers, wgs, N, H, W, x_ls, dy_ls, w_ls, c_ls, st)
  if (F == 24 && dtype == SR_DTYPE_BF16) return SR_WG(__bf16, 24, 144, 20);
  if (F == 24 && dtype == SR_DTYPE_F32) return SR_WG(float, 24, 144, 20);
  if (F == 32 && dtype == SR_DTYPE_BF16) return SR_WG(__bf16, 32, 192, 26);
  if (F == 32 && dtype == SR_DTYPE_F32) return SR_WG(float, 32, 192, 26);
#undef SR_WG
  return -1;
}

namespace {
template <int F, int E, int L>
int launch_wgrad_saved(const void* x, const void* dy, const void* tsave, const void* dtsave, const void* wblob,
                       const float* cinit, float* pa, float* pb, int layers, int wgs, int N, int H, int W, long x_ls,
                       long dy_ls, long side_ls, long w_ls, long c_ls, hipStream_t st) {
  typedef BlockCfg<F, E, L> C;
  typedef __bf16 T;
  const int tiles_x = (W + C::TW - 1) / C::TW, tiles_y = (H + C::TH - 1) / C::TH;
  dim3 grid(wgs, layers);
  static const bool env_a15 = SR_AB("SR_WGRAD_A15");        // the 15-wave kernel, kept for A/B measurements
  const bool old_a = env_a15 || (long)N * tiles_x * tiles_y * C::TH * C::TW >= (1L << 31);   // (the 8-wave kernel indexes pixels in 32 bits)
  if constexpr (F == 24) {
    if (!old_a)
      hipLaunchKernelGGL((wdsr_wgrad_a8_kernel<F, E, L>), grid, dim3(WgradA8Cfg<F, E, L>::NTHREADS), 0, st, (const T*)x, (const T*)dtsave,
                         (const T*)wblob, pa, N, H, W, tiles_x, tiles_x * tiles_y, x_ls, side_ls, w_ls);
  }
  if (F != 24 || old_a)
    hipLaunchKernelGGL((wdsr_block_wgrad_saved_kernel<T, F, E, L, 0>), grid, dim3(64 * WgradSavedCfg<F, E, L, 0>::NWAVES), 0, st,
                       (const T*)x, (const T*)dtsave, (const T*)wblob, cinit, pa, N, H, W, tiles_x, tiles_x * tiles_y, x_ls,
                       side_ls, w_ls, c_ls);
  static const bool old_b = SR_AB("SR_WGRAD_B9");           // the tap-per-wave kernel, kept for A/B measurements
  if (!old_b)
    hipLaunchKernelGGL((wdsr_wgrad_b8_kernel<F, E, L>), grid, dim3(WgradB8Cfg<F, E, L>::NTHREADS), 0, st, (const T*)dy, (const T*)tsave,
                       pb, N, H, W, tiles_x, tiles_x * tiles_y, dy_ls, side_ls);
  if (old_b)
    hipLaunchKernelGGL((wdsr_block_wgrad_saved_kernel<T, F, E, L, 1>), grid, dim3(64 * WgradSavedCfg<F, E, L, 1>::NWAVES), 0, st,
                       (const T*)dy, (const T*)tsave, (const T*)wblob, cinit, pb, N, H, W, tiles_x, tiles_x * tiles_y, dy_ls,
                       side_ls, w_ls, c_ls);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" int sr_wdsr_block_wgrad_saved(const void* x, const void* dy, const void* tsave, const void* dtsave,
                                         const void* wblob, const float* cinit, float* pa, float* pb, int layers, int wgs,
                                         int N, int H, int W, int F, int dtype, long x_ls, long dy_ls, long side_ls,
                                         long w_ls, long c_ls, sr_stream_t stream) {
  if (!x || !dy || !tsave || !dtsave || !wblob || !cinit || !pa || !pb || layers <= 0 || wgs <= 0 || N <= 0 || H <= 0 ||
      W <= 0 || layers > 65535)
    return -2;
  if (dtype != SR_DTYPE_BF16) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (F == 24) return launch_wgrad_saved<24, 144, 20>(x, dy, tsave, dtsave, wblob, cinit, pa, pb, layers, wgs, N, H, W, x_ls, dy_ls, side_ls, w_ls, c_ls, st);
  if (F == 32) return launch_wgrad_saved<32, 192, 26>(x, dy, tsave, dtsave, wblob, cinit, pa, pb, layers, wgs, N, H, W, x_ls, dy_ls, side_ls, w_ls, c_ls, st);
  return -1;
}

extern "C" int sr_wdsr_block_slab_sizes(int F, int* slab_a, int* slab_b) {
  if (!slab_a || !slab_b) return -2;
  if (F == 24) { *slab_a = BwdCfg<BlockCfg<24, 144, 20>>::SLAB_A; *slab_b = BwdCfg<BlockCfg<24, 144, 20>>::SLAB_B; return 0; }
  if (F == 32) { *slab_a = BwdCfg<BlockCfg<32, 192, 26>>::SLAB_A; *slab_b = BwdCfg<BlockCfg<32, 192, 26>>::SLAB_B; return 0; }
  return -1;
}

extern "C" int sr_wdsr_block_fwd(const void* x, void* y, const void* wblob, const float* cinit, int N, int H,
                                 int W, int F, int dtype, sr_stream_t stream) {
  if (!x || !y || !wblob || !cinit || N <= 0 || H <= 0 || W <= 0 || N > 65535) return -2;
  hipStream_t st = (hipStream_t)stream;
  if (F == 24 && dtype == SR_DTYPE_BF16) return launch_block_fwd<__bf16, 24, 144, 20>(x, y, wblob, cinit, N, H, W, st);
  if (F == 24 && dtype == SR_DTYPE_F32) return launch_block_fwd<float, 24, 144, 20>(x, y, wblob, cinit, N, H, W, st);
  if (F == 32 && dtype == SR_DTYPE_BF16) return launch_block_fwd<__bf16, 32, 192, 26>(x, y, wblob, cinit, N, H, W, st);
  if (F == 32 && dtype == SR_DTYPE_F32) return launch_block_fwd<float, 32, 192, 26>(x, y, wblob, cinit, N, H, W, st);
  return -1;
}

// ------------------------------------------------------------------------------------------
// head / tail
// ------------------------------------------------------------------------------------------
namespace {
template <typename E> dim3 tile_grid(int N, int H, int W, int* tiles_x) {
  *tiles_x = (W + E::TW - 1) / E::TW;
  return dim3(*tiles_x * ((H + E::TH - 1) / E::TH), N);
}
#define SR_DISPATCH_TF(CALL)                                                     \
  if (F == 24 && dtype == SR_DTYPE_BF16) { CALL(__bf16, 24) }                    \
  else if (F == 24 && dtype == SR_DTYPE_F32) { CALL(float, 24) }                 \
  else if (F == 32 && dtype == SR_DTYPE_BF16) { CALL(__bf16, 32) }               \
  else if (F == 32 && dtype == SR_DTYPE_F32) { CALL(float, 32) }                 \
  else return -1;
#define SR_DISPATCH_R(CALL, T, F_)                                               \
  if (R == 4) { CALL(T, F_, 4) } else if (R == 2) { CALL(T, F_, 2) } else if (R == 3) { CALL(T, F_, 3) } else return -1;
}  // namespace

extern "C" int sr_head_fwd(const float* x, void* y, const void* wblob, float mean, int N, int H, int W, int F,
                           int dtype, sr_stream_t stream) {
  if (!x || !y || !wblob || N <= 0 || H <= 0 || W <= 0 || N > 65535) return -2;
  hipStream_t st = (hipStream_t)stream;
#define CALL(T, F_) { typedef EndsCfg<F_, 4> E; int tx; dim3 g = tile_grid<E>(N, H, W, &tx); \
    hipLaunchKernelGGL((sr_head_fwd_kernel<T, F_>), g, dim3(256), 0, st, x, (T*)y, (const T*)wblob, mean, H, W, tx); }
  SR_DISPATCH_TF(CALL)
#undef CALL
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

extern "C" int sr_tail_fwd(const void* feat, const float* x, float* out, const void* wblob, float mean, int N,
                           int H, int W, int F, int R, int dtype, sr_stream_t stream) {
  if (!feat || !x || !out || !wblob || N <= 0 || H <= 0 || W <= 0 || N > 65535) return -2;
  hipStream_t st = (hipStream_t)stream;
#define CALLR(T, F_, R_) { typedef EndsCfg<F_, R_> E; int tx; dim3 g = tile_grid<E>(N, H, W, &tx); \
    hipLaunchKernelGGL((sr_tail_fwd_kernel<T, F_, R_>), g, dim3(SR_TAIL_FWD_THREADS), 0, st, (const T*)feat, x, out, (const T*)wblob, mean, H, W, tx); }
#define CALL(T, F_) SR_DISPATCH_R(CALLR, T, F_)
  SR_DISPATCH_TF(CALL)
#undef CALL
#undef CALLR
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

namespace {
// LOSS = 0: `dout` is d(loss)/d(out).  LOSS = 1 (L1) / 2 (Charbonnier): `dout` is the network output, `hr` the target,
// gscale = upstream gradient / numel; the wgrad / fused kernels also write one loss partial sum per workgroup.
template <int LOSS>
int tail_bwd_data_t(const float* dout, const float* hr, float gscale, void* dfeat, const void* wblob, int N, int H, int W, int F,
                    int R, int dtype, hipStream_t st) {
  const LossIn li{hr, gscale};
#define CALLR(T, F_, R_) { typedef EndsCfg<F_, R_> E; int tx; dim3 g = tile_grid<E>(N, H, W, &tx); \
    hipLaunchKernelGGL((sr_tail_bwd_data_kernel<T, F_, R_, LOSS>), g, dim3(256), 0, st, dout, (T*)dfeat, (const T*)wblob, H, W, tx, li); }
#define CALL(T, F_) SR_DISPATCH_R(CALLR, T, F_)
  SR_DISPATCH_TF(CALL)
#undef CALL
#undef CALLR
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
template <int LOSS>
int tail_wgrad_t(const float* dout, const float* hr, float gscale, float* loss_part, const void* feat, const float* x, float mean,
                 float* partial, int wgs, int N, int H, int W, int F, int R, int dtype, hipStream_t st) {
  const LossIn li{hr, gscale};
#define CALLR(T, F_, R_) { typedef EndsCfg<F_, R_> E; const int tx = (W + E::TW - 1) / E::TW, tpi = tx * ((H + E::TH - 1) / E::TH); \
    hipLaunchKernelGGL((sr_tail_wgrad_kernel<T, F_, R_, LOSS>), dim3(wgs), dim3(896), 0, st, dout, (const T*)feat, x, mean, partial, N, H, W, tx, tpi, li, loss_part); }
#define CALL(T, F_) SR_DISPATCH_R(CALLR, T, F_)
  SR_DISPATCH_TF(CALL)
#undef CALL
#undef CALLR
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
template <int LOSS>
int tail_bwd_t(const float* dout, const float* hr, float gscale, float* loss_part, const void* feat, const float* x, float mean,
               const void* wblob, void* dfeat, float* partial, int wgs, int N, int H, int W, int F, int R, hipStream_t st) {
  const LossIn li{hr, gscale};
  typedef __bf16 TB;
#define CALLR(T, F_, R_) { typedef EndsCfg<F_, R_> E; const int tx = (W + E::TW - 1) / E::TW, tpi = tx * ((H + E::TH - 1) / E::TH); \
    hipLaunchKernelGGL((sr_tail_bwd_kernel<T, F_, R_, LOSS>), dim3(wgs), dim3(896), 0, st, dout, (const T*)feat, x, mean, (const T*)wblob, (T*)dfeat, partial, N, H, W, tx, tpi, li, loss_part); }
  if (F == 24) { SR_DISPATCH_R(CALLR, TB, 24) } else if (F == 32) { SR_DISPATCH_R(CALLR, TB, 32) } else return -1;
#undef CALLR
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" int sr_tail_bwd_data(const float* dout, void* dfeat, const void* wblob, int N, int H, int W, int F,
                                int R, int dtype, sr_stream_t stream) {
  if (!dout || !dfeat || !wblob || N <= 0 || H <= 0 || W <= 0 || N > 65535) return -2;
  return tail_bwd_data_t<0>(dout, nullptr, 0.f, dfeat, wblob, N, H, W, F, R, dtype, (hipStream_t)stream);
}

extern "C" int sr_tail_wgrad(const float* dout, const void* feat, const float* x, float mean, float* partial,
                             int wgs, int N, int H, int W, int F, int R, int dtype, sr_stream_t stream) {
  if (!dout || !feat || !x || !partial || wgs <= 0 || N <= 0 || H <= 0 || W <= 0) return -2;
  return tail_wgrad_t<0>(dout, nullptr, 0.f, nullptr, feat, x, mean, partial, wgs, N, H, W, F, R, dtype, (hipStream_t)stream);
}

extern "C" int sr_tail_bwd(const float* dout, const void* feat, const float* x, float mean, const void* wblob,
                          void* dfeat, float* partial, int wgs, int N, int H, int W, int F, int R, int dtype,
                          sr_stream_t stream) {
  if (!dout || !feat || !x || !wblob || !dfeat || !partial || wgs <= 0 || N <= 0 || H <= 0 || W <= 0) return -2;
  if (dtype != SR_DTYPE_BF16) return -1;
  return tail_bwd_t<0>(dout, nullptr, 0.f, nullptr, feat, x, mean, wblob, dfeat, partial, wgs, N, H, W, F, R, (hipStream_t)stream);
}

// tail backward with the loss folded in: `sr` = network output, `hr` = target (both NCHW fp32 HR), loss_kind 1 = L1,
// 2 = Charbonnier; writes dfeat, the weight-gradient slabs and loss_part[wgs] (sums of |sr - hr| resp. sqrt(d^2 + eps))
extern "C" int sr_tail_bwd_loss(const float* sr, const float* hr, int loss_kind, float gscale, float* loss_part, const void* feat,
                                const float* x, float mean, const void* wblob, void* dfeat, float* partial, int wgs, int N,
                                int H, int W, int F, int R, int dtype, sr_stream_t stream) {
  if (!sr || !hr || !loss_part || !feat || !x || !wblob || !dfeat || !partial || wgs <= 0 || N <= 0 || H <= 0 || W <= 0 || N > 65535)
    return -2;
  if (loss_kind != 1 && loss_kind != 2) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SR_DTYPE_BF16)
    return loss_kind == 1 ? tail_bwd_t<1>(sr, hr, gscale, loss_part, feat, x, mean, wblob, dfeat, partial, wgs, N, H, W, F, R, st)
                          : tail_bwd_t<2>(sr, hr, gscale, loss_part, feat, x, mean, wblob, dfeat, partial, wgs, N, H, W, F, R, st);
  int rc = loss_kind == 1 ? tail_bwd_data_t<1>(sr, hr, gscale, dfeat, wblob, N, H, W, F, R, dtype, st)
                          : tail_bwd_data_t<2>(sr, hr, gscale, dfeat, wblob, N, H, W, F, R, dtype, st);
  if (rc) return rc;
  return loss_kind == 1 ? tail_wgrad_t<1>(sr, hr, gscale, loss_part, feat, x, mean, partial, wgs, N, H, W, F, R, dtype, st)
                        : tail_wgrad_t<2>(sr, hr, gscale, loss_part, feat, x, mean, partial, wgs, N, H, W, F, R, dtype, st);
}

extern "C" int sr_head_wgrad(const void* dy0, const float* x, float mean, float* partial, int wgs, int N, int H,
                             int W, int F, int dtype, sr_stream_t stream) {
  if (!dy0 || !x || !partial || wgs <= 0 || N <= 0 || H <= 0 || W <= 0) return -2;
  hipStream_t st = (hipStream_t)stream;
#define CALL(T, F_) { typedef EndsCfg<F_, 4> E; const int tx = (W + E::TW - 1) / E::TW, tpi = tx * ((H + E::TH - 1) / E::TH); \
    hipLaunchKernelGGL((sr_head_wgrad_kernel<T, F_>), dim3(wgs), dim3(576), 0, st, (const T*)dy0, x, mean, partial, N, H, W, tx, tpi); }
  SR_DISPATCH_TF(CALL)
#undef CALL
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// BasicVSR trunk convolutions
// ------------------------------------------------------------------------------------------
namespace {
struct C3Grid { int tx, tpi; dim3 grid; };
inline C3Grid c3_grid(int N, int H, int W) {
  C3Grid g;
  g.tx = (W + C3Cfg::TW - 1) / C3Cfg::TW;
  g.tpi = g.tx * ((H + C3Cfg::TH - 1) / C3Cfg::TH);
  g.grid = dim3(g.tpi, N);
  return g;
}
template <typename T> C3WarpSrc<T> c3_warp_src(const sr_c3_warp_t* w) {
  C3WarpSrc<T> s{};
  if (w) { s.frame = w->frame; s.state = (const T*)w->state; s.flow = w->flow; s.frame_bs = w->frame_bs; s.flow_bs = w->flow_bs; s.x0_save = (T*)w->x0_save; }
  return s;
}
template <typename T>
int c3_fwd_t(const void* x, const void* res, void* y, const void* w, int N, int H, int W, int CI, int act, hipStream_t st,
             const sr_c3_warp_t* warp = nullptr, C3Dir dir = C3Dir{0, 0}) {
  const C3Grid g = c3_grid(N, H, W);
  const dim3 blk(64 * C3Cfg::NPT_O);
  const C3WarpSrc<T> ws = c3_warp_src<T>(warp);
#define L(CI_, ONES_, ACT_, ADD_) hipLaunchKernelGGL((c3_fwd_kernel<T, CI_, ONES_, ACT_, ADD_>), g.grid, blk, 0, st, (const T*)x, (const T*)res, (T*)y, (const T*)w, H, W, g.tx, ws, dir)
  if (warp) {
    if (CI != 32 || act != 2 || res) return -1;
    hipLaunchKernelGGL((c3_fwd_kernel<T, 32, 27, 2, false, true>), g.grid, blk, 0, st, (const T*)nullptr, (const T*)nullptr, (T*)y,
                       (const T*)w, H, W, g.tx, ws, dir);
  } else if (CI == 32 && act == 2 && !res) L(32, 27, 2, false);
  else if (CI == 24 && act == 1 && !res) L(24, 24, 1, false);
  else if (CI == 24 && act == 0 && res) L(24, 24, 0, true);
  else if (CI == 24 && act == 0 && !res) L(24, 24, 0, false);
  else return -1;
#undef L
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
template <typename T>
int c3_bwd_t(const void* dA, const void* A, const void* add, void* dx, const void* w, int N, int H, int W, int CI, int act,
             hipStream_t st, C3Dir dir = C3Dir{0, 0}) {
  const C3Grid g = c3_grid(N, H, W);
  const dim3 blk(64 * C3Cfg::NPT_O);
#define L(CI_, ACT_, ADD_) hipLaunchKernelGGL((c3_bwd_data_kernel<T, CI_, ACT_, ADD_>), g.grid, blk, 0, st, (const T*)dA, (const T*)A, (const T*)add, (T*)dx, (const T*)w, H, W, g.tx, dir)
  if (CI == 32 && act == 2 && !add) L(32, 2, false);
  else if (CI == 24 && act == 1 && add) L(24, 1, true);
  else if (CI == 24 && act == 1 && !add) L(24, 1, false);
  else if (CI == 24 && act == 0 && !add) L(24, 0, false);
  else return -1;
#undef L
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
template <typename T>
int c3_wgrad_t(const void* x, const void* dA, const void* A, float* partial, int wgs, int N, int H, int W, int CI, int act,
               hipStream_t st, int layers = 1, long x_ls = 0, long d_ls = 0, long a_ls = 0, long p_ls = 0,
               const sr_c3_warp_t* warp = nullptr, int n_dir = 0) {
  const C3Grid g = c3_grid(N, H, W);
  const C3WarpSrc<T> ws = c3_warp_src<T>(warp);
#define L(CI_, ONES_, ACT_) hipLaunchKernelGGL((c3_wgrad_kernel<T, CI_, ONES_, ACT_>), dim3(wgs, layers), dim3(576), 0, st, (const T*)x, (const T*)dA, (const T*)A, partial, N, H, W, g.tx, g.tpi, x_ls, d_ls, a_ls, p_ls, ws, n_dir)
  if (warp) {
    if (CI != 32 || act != 2 || layers != 1) return -1;
    hipLaunchKernelGGL((c3_wgrad_kernel<T, 32, 27, 2, true>), dim3(wgs, 1), dim3(576), 0, st, (const T*)nullptr, (const T*)dA,
                       (const T*)A, partial, N, H, W, g.tx, g.tpi, x_ls, d_ls, a_ls, p_ls, ws, n_dir);
  } else if (CI == 32 && act == 2) L(32, 27, 2);
  else if (CI == 24 && act == 1) L(24, 24, 1);
  else if (CI == 24 && act == 0) L(24, 24, 0);
  else return -1;
#undef L
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" int sr_c3_fwd(const void* x, const void* res, void* y, const void* wblob, int N, int H, int W, int CI,
                         int act, int dtype, sr_stream_t stream) {
  if (!x || !y || !wblob || N <= 0 || H <= 0 || W <= 0 || N > 65535) return -2;
  return dtype == SR_DTYPE_BF16 ? c3_fwd_t<__bf16>(x, res, y, wblob, N, H, W, CI, act, (hipStream_t)stream)
                                : c3_fwd_t<float>(x, res, y, wblob, N, H, W, CI, act, (hipStream_t)stream);
}
extern "C" int sr_c3_bwd_data(const void* dA, const void* A, const void* add, void* dx, const void* wblob, int N,
                              int H, int W, int CI, int act, int dtype, sr_stream_t stream) {
  if (!dA || !dx || !wblob || (act != 0 && !A) || N <= 0 || H <= 0 || W <= 0 || N > 65535) return -2;
  if (!A) A = dA;
  return dtype == SR_DTYPE_BF16 ? c3_bwd_t<__bf16>(dA, A, add, dx, wblob, N, H, W, CI, act, (hipStream_t)stream)
                                : c3_bwd_t<float>(dA, A, add, dx, wblob, N, H, W, CI, act, (hipStream_t)stream);
}
extern "C" int sr_c3_wgrad(const void* x, const void* dA, const void* A, float* partial, int wgs, int N, int H,
                           int W, int CI, int act, int dtype, sr_stream_t stream) {
  if (!x || !dA || !partial || (act != 0 && !A) || wgs <= 0 || N <= 0 || H <= 0 || W <= 0) return -2;
  if (!A) A = dA;
  return dtype == SR_DTYPE_BF16 ? c3_wgrad_t<__bf16>(x, dA, A, partial, wgs, N, H, W, CI, act, (hipStream_t)stream)
                                : c3_wgrad_t<float>(x, dA, A, partial, wgs, N, H, W, CI, act, (hipStream_t)stream);
}

// whole propagation trunk (ConvResidualBlocks.forward, models/basicvsr_arch.py:108-147) from one call
template <typename T>
static int c3_trunk_fwd_t(const void* x0, const sr_c3_warp_t* warp, void* acts_, void* mids_, const void* blob_,
                          const long* boff, int nb, int N, int H, int W, int ci0, hipStream_t st, C3Dir dir) {
  const size_t act = (size_t)N * H * W * 24;
  T* acts = (T*)acts_; T* mids = (T*)mids_; const T* blob = (const T*)blob_;
  int rc;
  if ((rc = c3_fwd_t<T>(x0, nullptr, acts, blob + boff[0], N, H, W, ci0, 2, st, warp, dir))) return rc;
  for (int i = 0; i < nb; ++i) {
    if constexpr (sizeof(T) == 2) {
      const C3Grid g = c3_grid(N, H, W);
      const bool single = getenv("SR_C3_ONE_BLOCK_PER_LAUNCH") != nullptr;   // (read per call: the parity test chains the two forms)
      if (!single && i + 1 < nb) {                     // two residual blocks per launch
        hipLaunchKernelGGL((c3_resblock2_fwd_kernel<T>), g.grid, dim3(64 * C3Quad::NWAVES), 0, st, acts + i * act, mids + i * act,
                           acts + (i + 1) * act, mids + (i + 1) * act, acts + (i + 2) * act, blob, boff[1 + 2 * i], boff[2 + 2 * i],
                           boff[3 + 2 * i], boff[4 + 2 * i], H, W, g.tx, dir);
        SR_HIP_CHECK_LAUNCH();
        ++i;
        continue;
      }
      hipLaunchKernelGGL((c3_resblock_fwd_kernel<T>), g.grid, dim3(64 * C3Pair::NPT_H), 0, st, acts + i * act, mids + i * act,
                         acts + (i + 1) * act, blob + boff[1 + 2 * i], blob + boff[2 + 2 * i], H, W, g.tx, dir);
      SR_HIP_CHECK_LAUNCH();
      continue;
    }
    if ((rc = c3_fwd_t<T>(acts + i * act, nullptr, mids + i * act, blob + boff[1 + 2 * i], N, H, W, 24, 1, st, nullptr, dir))) return rc;
    if ((rc = c3_fwd_t<T>(mids + i * act, acts + i * act, acts + (i + 1) * act, blob + boff[2 + 2 * i], N, H, W, 24, 0, st, nullptr, dir)))
      return rc;
  }
  return 0;
}
template <typename T>
static int c3_trunk_bwd_t(const void* x0, const sr_c3_warp_t* warp, const void* acts_, const void* mids_, void* ga_, void* gt_,
                          const void* blob_, const long* boff, float* parts, void* dx0, const sr_c3_unpack_t* up, int nb,
                          int wgs, int N, int H, int W, int ci0, hipStream_t st, C3Dir dir) {
  const size_t act = (size_t)N * H * W * 24;
  const T* acts = (const T*)acts_; const T* mids = (const T*)mids_; const T* blob = (const T*)blob_;
  T* ga = (T*)ga_; T* gt = (T*)gt_;
  const long slot = (long)wgs * 9 * 1024;
  int rc;
  for (int i = nb - 1; i >= 0; --i) {                // a_{i+1} = a_i + conv2(t_i), t_i = relu(conv1(a_i))
    if constexpr (sizeof(T) == 2) {
      const C3Grid g = c3_grid(N, H, W);
      hipLaunchKernelGGL((c3_resblock_bwd_data_kernel<T>), g.grid, dim3(64 * C3Pair::NPT_H), 0, st, ga + (i + 1) * act,
                         mids + i * act, gt + i * act, ga + i * act, blob + boff[1 + 2 * i], blob + boff[2 + 2 * i], H, W, g.tx, dir);
      SR_HIP_CHECK_LAUNCH();
      continue;
    }
    if ((rc = c3_bwd_t<T>(ga + (i + 1) * act, ga + (i + 1) * act, nullptr, gt + i * act, blob + boff[2 + 2 * i], N, H, W, 24, 0, st, dir)))
      return rc;
    if ((rc = c3_bwd_t<T>(gt + i * act, mids + i * act, ga + (i + 1) * act, ga + i * act, blob + boff[1 + 2 * i], N, H, W, 24, 1, st, dir)))
      return rc;
  }
  if (nb > 0) {                                      // every conv2, then every conv1, one launch each
    if ((rc = c3_wgrad_t<T>(mids, ga + act, ga + act, parts + 2 * slot, wgs, N, H, W, 24, 0, st, nb, (long)act, (long)act,
                            (long)act, 2 * slot, nullptr, dir.n_dir)))
      return rc;
    if ((rc = c3_wgrad_t<T>(acts, gt, mids, parts + slot, wgs, N, H, W, 24, 1, st, nb, (long)act, (long)act, (long)act, 2 * slot,
                            nullptr, dir.n_dir)))
      return rc;
  }
  const bool regather = warp && !warp->x0_save;      // the forward kept the gathered input unless told not to
  if ((rc = c3_wgrad_t<T>(warp && !regather ? warp->x0_save : x0, ga, acts, parts, wgs, N, H, W, ci0, 2, st, 1, 0, 0, 0, 0, regather ? warp : nullptr,
                          dir.n_dir)))
    return rc;
  if (dx0 && (rc = c3_bwd_t<T>(ga, acts, nullptr, dx0, blob + boff[0], N, H, W, ci0, 2, st, dir))) return rc;
  if (warp && (warp->dstate || warp->dflow)) {       // flow_warp backward, gather form (no float atomics)
    if (!dx0 || !warp->dstate || (warp->flow && !warp->flow_bound)) return -2;
    const int tx = (W + C3WarpBwd::TS - 1) / C3WarpBwd::TS, ty = (H + C3WarpBwd::TS - 1) / C3WarpBwd::TS;
    hipLaunchKernelGGL((c3_warp_bwd_kernel<T>), dim3(tx * ty, N), dim3(256), 0, st, (const T*)dx0, c3_warp_src<T>(warp),
                       warp->flow_bound, (T*)warp->dstate, warp->dflow, warp->dflow_bs, H, W, tx);
    SR_HIP_CHECK_LAUNCH();
  }
  if (up) {                                          // slabs -> gradient of the flat parameter (of each trunk: its half of the slabs)
    UnpackSegs us;
    const long slab = 9 * 1024;
    const int ndir = dir.n_dir > 0 ? 2 : 1, wd = wgs / ndir;
    const long total = (long)up->n0 + (nb > 0 ? 2L * nb * up->n1 : 0);
    us.nseg = 0;
    int blk = 0;
    for (int d = 0; d < ndir; ++d) {
      us.s[us.nseg++] = UnpackSeg{parts + (size_t)d * wd * slab, up->sidx0, up->dst0, d * total, 0, slab, wd, up->n0, 1, blk, wgs};
      blk += unpack_blocks(up->n0, wd);
      if (nb > 0) {
        // layer l's slabs start at parts + (1 + l) wgs slab: the stride between layers stays wgs slabs, this trunk's start d wd slabs in
        us.s[us.nseg++] = UnpackSeg{parts + (size_t)wgs * slab + (size_t)d * wd * slab, up->sidx1, up->dst1, d * total + (long)up->n0,
                                    (long)up->n1, slab, wd, up->n1, 2 * nb, blk, wgs};
        blk += 2 * nb * unpack_blocks(up->n1, wd);
      }
    }
    hipLaunchKernelGGL(unpack_all_kernel, dim3(blk), dim3(64 * UNPACK_Q), 0, st, up->gflat, us);
    SR_HIP_CHECK_LAUNCH();
  }
  return 0;
}
extern "C" int sr_c3_trunk_fwd(const void* x0, const sr_c3_warp_t* warp, void* acts, void* mids, const void* blob,
                               const long* blob_off, int nb, int N, int H, int W, int ci0, int dtype, int n_dir, long blob_dir_stride,
                               sr_stream_t stream) {
  if ((!x0) == (!warp) || !acts || (nb > 0 && !mids) || !blob || !blob_off || nb < 0 || N <= 0 || H <= 0 || W <= 0 || N > 65535) return -2;
  if (warp && (!warp->frame || ci0 != 32 || (warp->flow && !warp->state))) return -2;
  if (n_dir < 0 || n_dir >= N) return -2;
  const C3Dir dir{blob_dir_stride, n_dir};
  return dtype == SR_DTYPE_BF16 ? c3_trunk_fwd_t<__bf16>(x0, warp, acts, mids, blob, blob_off, nb, N, H, W, ci0, (hipStream_t)stream, dir)
                                : c3_trunk_fwd_t<float>(x0, warp, acts, mids, blob, blob_off, nb, N, H, W, ci0, (hipStream_t)stream, dir);
}
extern "C" int sr_c3_trunk_bwd(const void* x0, const sr_c3_warp_t* warp, const void* acts, const void* mids, void* ga, void* gt,
                               const void* blob, const long* blob_off, float* parts, void* dx0, const sr_c3_unpack_t* unpack,
                               int nb, int wgs, int N, int H, int W, int ci0, int dtype, int n_dir, long blob_dir_stride,
                               sr_stream_t stream) {
  if ((!x0) == (!warp) || !acts || !ga || (nb > 0 && (!mids || !gt)) || !blob || !blob_off || !parts || nb < 0 || wgs <= 0 || N <= 0 ||
      H <= 0 || W <= 0 || N > 65535)
    return -2;
  if (warp && (!warp->frame || ci0 != 32 || (warp->flow && !warp->state))) return -2;
  if (unpack && (!unpack->sidx0 || !unpack->dst0 || !unpack->gflat || unpack->n0 <= 0 ||
                 (nb > 0 && (!unpack->sidx1 || !unpack->dst1 || unpack->n1 <= 0))))
    return -2;
  if (n_dir < 0 || n_dir >= N || (n_dir > 0 && (wgs & 1))) return -2;
  const C3Dir dir{blob_dir_stride, n_dir};
  return dtype == SR_DTYPE_BF16
             ? c3_trunk_bwd_t<__bf16>(x0, warp, acts, mids, ga, gt, blob, blob_off, parts, dx0, unpack, nb, wgs, N, H, W, ci0, (hipStream_t)stream, dir)
             : c3_trunk_bwd_t<float>(x0, warp, acts, mids, ga, gt, blob, blob_off, parts, dx0, unpack, nb, wgs, N, H, W, ci0, (hipStream_t)stream, dir);
}

// ------------------------------------------------------------------------------------------
// flow_warp
// ------------------------------------------------------------------------------------------
extern "C" int sr_flow_warp_fwd(const float* x, const float* flow, float* out, int N, int C, int H, int W,
                                sr_stream_t stream) {
  if (!x || !flow || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || N > 65535) return -2;
  const int blocks = std::min((H * W + 15) / 16, 8192);          // one wave = 16 pixels x 4 channel phases
  hipLaunchKernelGGL(flow_warp_fwd_kernel, dim3(blocks, N), dim3(64), 0, (hipStream_t)stream, x, flow, out, C, H, W);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
extern "C" int sr_flow_warp_bwd(const float* x, const float* flow, const float* gout, float* dx, float* dflow, int N,
                                int C, int H, int W, sr_stream_t stream) {
  if (!x || !flow || !gout || N <= 0 || C <= 0 || H <= 0 || W <= 0 || N > 65535) return -2;
  const int blocks = std::min((H * W + 15) / 16, 8192);
  hipLaunchKernelGGL(flow_warp_bwd_kernel, dim3(blocks, N), dim3(64), 0, (hipStream_t)stream, x, flow, gout, dx, dflow,
                     C, H, W);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// NAS supernet block
// ------------------------------------------------------------------------------------------
#define SR_NAS_DISPATCH(CALL)                                                                   \
  if (F == 24 && dtype == SR_DTYPE_BF16) { CALL(__bf16, 24) } else if (F == 24 && dtype == SR_DTYPE_F32) { CALL(float, 24) } \
  else if (F == 32 && dtype == SR_DTYPE_BF16) { CALL(__bf16, 32) } else if (F == 32 && dtype == SR_DTYPE_F32) { CALL(float, 32) } \
  else return -1;

extern "C" int sr_nas_dw_fwd(const void* yin, void* V, const float* dwp, int N, int H, int W, int F, int dtype,
                             sr_stream_t stream) {
  if (!yin || !V || !dwp || N <= 0 || H <= 0 || W <= 0 || N > 65535) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long vs = (long)N * H * W * F;
  static const bool valu_dw = SR_AB("SR_NAS_DW_VALU");     // the VALU stencils also in bf16 mode (A/B measurements)
  if (dtype == SR_DTYPE_BF16 && !valu_dw && (F == 24 || F == 32)) {      // lane = channel, packed bf16 dot products (csrc/nas_dw_lc.h)
    typedef NasCfg<24> C;
    const int tx = (W + C::TW - 1) / C::TW;
    dim3 g(tx * ((H + C::TH - 1) / C::TH), N);
    if (F == 24) hipLaunchKernelGGL((nas_dw_fwd_lc_kernel<24>), g, dim3(512), 0, st, (const __bf16*)yin, (__bf16*)V, dwp, H, W, tx, vs);
    else hipLaunchKernelGGL((nas_dw_fwd_lc_kernel<32>), g, dim3(512), 0, st, (const __bf16*)yin, (__bf16*)V, dwp, H, W, tx, vs);
    SR_HIP_CHECK_LAUNCH();
    return 0;
  }
#define CALL(T, F_) { typedef NasCfg<F_> C; const int tx = (W + C::TW - 1) / C::TW; dim3 g(tx * ((H + C::TH - 1) / C::TH), N); \
    hipLaunchKernelGGL((nas_dw_fwd_kernel<T, F_>), g, dim3(512), 0, st, (const T*)yin, (T*)V, dwp, H, W, tx, vs); }
  SR_NAS_DISPATCH(CALL)
#undef CALL
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
extern "C" int sr_nas_pw_fwd(const void* yin, const void* V, void* y, const void* frags, const float* tabs,
                             const float* scal, int N, int H, int W, int F, int dtype, sr_stream_t stream) {
  if (!yin || !V || !y || !frags || !tabs || !scal || N <= 0 || H <= 0 || W <= 0 || N > 65535) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long vs = (long)N * H * W * F;
#define CALL(T, F_) { typedef NasCfg<F_> C; const int tx = (W + C::TW - 1) / C::TW; dim3 g(tx * ((H + C::TH - 1) / C::TH), N); \
    hipLaunchKernelGGL((nas_pw_fwd_kernel<T, F_>), g, dim3(576), 0, st, (const T*)yin, (const T*)V, (T*)y, (const T*)frags, tabs, scal, H, W, tx, vs); }
  SR_NAS_DISPATCH(CALL)
#undef CALL
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
extern "C" int sr_nas_pw_bwd(const void* yin, const void* V, const void* gy, void* GZ, const void* frags,
                             const float* tabs, const float* scal, float* partial, int wgs, int N, int H, int W, int F,
                             int dtype, sr_stream_t stream) {
  if (!yin || !V || !gy || !GZ || !frags || !tabs || !scal || !partial || wgs <= 0 || N <= 0 || H <= 0 || W <= 0) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long vs = (long)N * H * W * F;
#define CALL(T, F_) { typedef NasCfg<F_> C; const int tx = (W + C::TW - 1) / C::TW, tpi = tx * ((H + C::TH - 1) / C::TH); \
    hipLaunchKernelGGL((nas_pw_bwd_kernel<T, F_>), dim3(wgs), dim3(NasPwb<T>::THREADS), 0, st, (const T*)yin, (const T*)V, (const T*)gy, (T*)GZ, (const T*)frags, tabs, scal, partial, N, H, W, tx, tpi, vs); }
  SR_NAS_DISPATCH(CALL)
#undef CALL
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
extern "C" int sr_nas_dw_bwd(const void* yin, const void* GZ, const void* gy, void* gyin, const float* dwp,
                             float* partial, int wgs, int N, int H, int W, int F, int dtype, sr_stream_t stream) {
  if (!yin || !GZ || !gy || !gyin || !dwp || !partial || wgs <= 0 || N <= 0 || H <= 0 || W <= 0) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long vs = (long)N * H * W * F;
  static const bool valu_dw = SR_AB("SR_NAS_DW_VALU");
  if (dtype == SR_DTYPE_BF16 && !valu_dw && (F == 24 || F == 32)) {      // lane = channel (csrc/nas_dw_lc.h); the dW part of the slab is sr_nas_dw_wgrad's
    typedef NasCfg<24> C;
    const int tx = (W + C::TW - 1) / C::TW, tpi = tx * ((H + C::TH - 1) / C::TH);
    if (F == 24) hipLaunchKernelGGL((nas_dw_bwd_lc_kernel<24>), dim3(wgs), dim3(NAS_DW_BWD_THREADS), 0, st, (const __bf16*)yin, (const __bf16*)GZ, (const __bf16*)gy, (__bf16*)gyin, dwp, partial, N, H, W, tx, tpi, vs);
    else hipLaunchKernelGGL((nas_dw_bwd_lc_kernel<32>), dim3(wgs), dim3(NAS_DW_BWD_THREADS), 0, st, (const __bf16*)yin, (const __bf16*)GZ, (const __bf16*)gy, (__bf16*)gyin, dwp, partial, N, H, W, tx, tpi, vs);
    SR_HIP_CHECK_LAUNCH();
    return 0;
  }
#define CALL(T, F_) { typedef NasCfg<F_> C; const int tx = (W + C::TW - 1) / C::TW, tpi = tx * ((H + C::TH - 1) / C::TH); \
    hipLaunchKernelGGL((nas_dw_bwd_kernel<T, F_>), dim3(wgs), dim3(512), 0, st, (const T*)yin, (const T*)GZ, (const T*)gy, (T*)gyin, dwp, partial, N, H, W, tx, tpi, vs); }
  SR_NAS_DISPATCH(CALL)
#undef CALL
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

extern "C" int sr_nas_dw_wgrad(const void* yin, const void* GZ, const float* dwp, float* partial, int wgs, int N, int H,
                               int W, int F, int dtype, sr_stream_t stream) {
  if (!yin || !GZ || !dwp || !partial || wgs <= 0 || N <= 0 || H <= 0 || W <= 0) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long vs = (long)N * H * W * F;
  static const bool wgrad_split = SR_AB("SR_NAS_WGRAD_SPLIT");
  if (dtype == SR_DTYPE_BF16 && !wgrad_split && (F == 24 || F == 32)) {   // the three stencils from one workgroup per tile
    typedef NasCfg<24> C;
    const int tx = (W + C::TW - 1) / C::TW, tpi = tx * ((H + C::TH - 1) / C::TH);
    if (F == 24) hipLaunchKernelGGL((nas_dw_wgrad3_kernel<24>), dim3(wgs), dim3(768), 0, st, (const __bf16*)yin, (const __bf16*)GZ, dwp, partial, N, H, W, tx, tpi, vs);
    else hipLaunchKernelGGL((nas_dw_wgrad3_kernel<32>), dim3(wgs), dim3(768), 0, st, (const __bf16*)yin, (const __bf16*)GZ, dwp, partial, N, H, W, tx, tpi, vs);
    SR_HIP_CHECK_LAUNCH();
    return 0;
  }
#define CALL(T, F_) { typedef NasCfg<F_> C; const int tx = (W + C::TW - 1) / C::TW, tpi = tx * ((H + C::TH - 1) / C::TH); \
    hipLaunchKernelGGL((nas_dw_wgrad_kernel<T, F_>), dim3(wgs, 3), dim3(768), 0, st, (const T*)yin, (const T*)GZ, dwp, partial, N, H, W, tx, tpi, vs); }
  SR_NAS_DISPATCH(CALL)
#undef CALL
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// whole network
// ------------------------------------------------------------------------------------------
namespace {
template <typename T> int net_pack(const sr_wdsr_net_t* n, hipStream_t st) {
  const int cb = (n->n_chan + 3) / 4, bb = (n->n_bias + 255) / 256;
  hipLaunchKernelGGL(wn_src_kernel, dim3(cb + bb), dim3(256), 0, st, n->flat, n->src, (const int4*)n->chan_tab,
                     n->n_chan, n->bias_tab, n->bias_const, n->n_bias, cb);
  PackSegs ps;
  ps.nseg = 4;
  int blk = 0;
  auto add = [&](int k, const int* idx, void* out, long off, long stride, int cnt, int reps, int as_float) {
    ps.s[k] = PackSeg{idx, out, off, stride, cnt, reps, as_float, blk};
    blk += reps * ((cnt + 255) / 256);
  };
  add(0, n->idx_head, n->blob_head, n->src_head_off, 0, n->n_idx_head, 1, 0);
  add(1, n->idx_body, n->blob_body, n->src_body_off, n->src_body_stride, n->n_idx_body, n->NB, 0);
  add(2, n->idx_cinit, n->cinit_body, n->src_body_off, n->src_body_stride, n->n_idx_cinit, n->NB, 1);
  add(3, n->idx_tail, n->blob_tail, n->src_tail_off, 0, n->n_idx_tail, 1, 0);
  hipLaunchKernelGGL((pack_all_kernel<T>), dim3(blk), dim3(256), 0, st, n->src, ps);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
}  // namespace

// t / dt of every block are kept for the weight-gradient kernels when every block runs through the
// two-block kernels (bf16, F = 24, even block count) and the caller provided both buffers
static bool net_saves_side_images(const sr_wdsr_net_t* n, bool backward) {
  return (n->F == 24 || n->F == 32) && n->dtype == SR_DTYPE_BF16 && n->tsave && (!backward || n->dtsave);
}
// Two blocks per launch pay while a launch is bound by its fixed costs (about one workgroup per CU); with more
// workgroups the single-block kernels win (two resident per CU, no halo-2 recompute): measured crossover at
// batch 64 of 48x48 patches = 512 workgroups (tools/bench_rows.py).
static long net_tiles(const sr_wdsr_net_t* n) {
  typedef BlockCfg<24, 144, 20> C;
  return (long)n->N * ((n->W + C::TW - 1) / C::TW) * ((n->H + C::TH - 1) / C::TH);
}
static bool net_uses_pairs(const sr_wdsr_net_t* n) { return n->F == 24 && n->dtype == SR_DTYPE_BF16 && net_tiles(n) <= 384; }
static size_t side_image_bytes(const sr_wdsr_net_t* n) {     // one block's [N][tiles][288][LP] image
  typedef BlockCfg<24, 144, 20> C;
  const size_t tiles = (size_t)((n->W + C::TW - 1) / C::TW) * ((n->H + C::TH - 1) / C::TH);
  const int lp = n->F == 24 ? BlockCfg<24, 144, 20>::LP : BlockCfg<32, 192, 26>::LP;
  return (size_t)n->N * tiles * C::TH * C::TW * lp * 2;
}

extern "C" int sr_wdsr_net_forward(const sr_wdsr_net_t* n, int flags, sr_stream_t stream) {
  if (!n || !n->flat || !n->src) return -2;
  if (!(flags & SR_NET_PACK_ONLY) && (!n->x || !n->acts || !n->out)) return -2;
  hipStream_t st = (hipStream_t)stream;
  const size_t esz = n->dtype == SR_DTYPE_BF16 ? 2 : 4;
  const int save_acts = flags & SR_NET_SAVE_ACTS;
  int rc = 0;
  if (!(flags & SR_NET_WEIGHTS_PACKED)) rc = n->dtype == SR_DTYPE_BF16 ? net_pack<__bf16>(n, st) : net_pack<float>(n, st);
  if (rc || (flags & SR_NET_PACK_ONLY)) return rc;
  const size_t act = (size_t)n->N * n->H * n->W * n->F * esz;
  const size_t blob = (size_t)n->n_idx_body * esz;
  char* acts = (char*)n->acts;
  if ((rc = sr_head_fwd(n->x, acts, n->blob_head, n->mean, n->N, n->H, n->W, n->F, n->dtype, stream))) return rc;
  char* cur = acts;
  // inference over many tiles per CU: the persistent two-block launches (csrc/wdsr_fwd_rs.h) beat the single-block ones
  // again (0.29 vs 0.27 of the roof at batch 512); with saved images (training) the large grids stay on single blocks
  // 32 units (round 3): two blocks per forward launch at launch-bound grids as well (the sixteen-wave kernel; SR_F32_ONE_BLOCK=1, read
  // per call for the parity test, keeps the one-block kernels); its backward stays one block per launch
  const bool pairs32 = n->F == 32 && n->dtype == SR_DTYPE_BF16 && net_tiles(n) <= 384 && !getenv("SR_F32_ONE_BLOCK");
  const bool pairs = net_uses_pairs(n) || pairs32 || (n->F == 24 && n->dtype == SR_DTYPE_BF16 &&
                                           ((!save_acts && net_tiles(n) >= 768) || fwd_stream_applies(n->N, n->H, n->W)));
  const bool saved = net_saves_side_images(n, false);
  const size_t side = side_image_bytes(n);
  for (int i = 0; i < n->NB; ++i) {
    if (pairs && i + 1 < n->NB) {
      char* mid = save_acts ? acts + (size_t)(i + 1) * act : nullptr;   // inference keeps nothing
      char* nxt = save_acts ? acts + (size_t)(i + 2) * act : (cur == acts ? acts + act : acts);
      char* ts = (save_acts && saved) ? (char*)n->tsave : nullptr;
      if ((rc = sr_wdsr_block2_fwd(cur, mid, nxt, (char*)n->blob_body + i * blob, (char*)n->blob_body + (i + 1) * blob,
                                   n->cinit_body + (size_t)i * n->n_idx_cinit,
                                   n->cinit_body + (size_t)(i + 1) * n->n_idx_cinit, ts ? ts + (size_t)i * side : nullptr,
                                   ts ? ts + (size_t)(i + 1) * side : nullptr, n->N, n->H, n->W, n->F, n->dtype, stream)))
        return rc;
      cur = nxt;
      ++i;
      continue;
    }
    char* nxt = save_acts ? acts + (size_t)(i + 1) * act : (cur == acts ? acts + act : acts);
    if (pairs) {                                  // odd block count at a launch-bound grid: same kernel family, one block
      char* ts = (save_acts && saved) ? (char*)n->tsave + (size_t)i * side : nullptr;
      if ((rc = sr_wdsr_fwd_rs(cur, nullptr, nxt, (char*)n->blob_body + i * blob, nullptr, n->cinit_body + (size_t)i * n->n_idx_cinit,
                               nullptr, ts, nullptr, 1, n->N, n->H, n->W, n->F, n->dtype, stream)))
        return rc;
      cur = nxt;
      continue;
    }
    if (save_acts && saved) {                     // single-block kernel that also keeps t (bf16)
      rc = n->F == 24 ? launch_block_fwd<__bf16, 24, 144, 20>(cur, nxt, (char*)n->blob_body + i * blob,
                                                              n->cinit_body + (size_t)i * n->n_idx_cinit, n->N, n->H, n->W,
                                                              st, (char*)n->tsave + (size_t)i * side)
                      : launch_block_fwd<__bf16, 32, 192, 26>(cur, nxt, (char*)n->blob_body + i * blob,
                                                              n->cinit_body + (size_t)i * n->n_idx_cinit, n->N, n->H, n->W,
                                                              st, (char*)n->tsave + (size_t)i * side);
      if (rc) return rc;
    } else if ((rc = sr_wdsr_block_fwd(cur, nxt, (char*)n->blob_body + i * blob,
                                       n->cinit_body + (size_t)i * n->n_idx_cinit, n->N, n->H, n->W, n->F, n->dtype, stream)))
      return rc;
    cur = nxt;
  }
  return sr_tail_fwd(cur, n->x, n->out, n->blob_tail, n->mean, n->N, n->H, n->W, n->F, n->R, n->dtype, stream);
}

// Backward in up to two parts so that the gradient of the LATE parameters (blocks [nb_split, NB), tail, skip) is final --
// bucket-ready for a DistributedDataParallel all-reduce -- before the early half (head, blocks [0, nb_split)) runs.
//   part 0: everything (one call);  part 1: late half;  part 2: early half (after part 1).
namespace {
struct FusedAdam { float* m; float* v; AdamArgs a; const float* loss_part; int n_loss; float loss_scale; float* loss_out; };
}
// fa != nullptr (part 0 only): the weight-norm backward also applies the Adam step (wn_bwd_adam_kernel)
static int net_backward_part_impl(const sr_wdsr_net_t* n, int part, sr_stream_t stream, const FusedAdam* fa) {
  if (!n || !n->flat || !n->gflat || !n->dsrc || !n->x || !n->acts || !n->grads || part < 0 || part > 2) return -2;
  if (part != 2 && (n->hr ? (!n->out || !n->loss_part || (n->loss_kind != 1 && n->loss_kind != 2)) : !n->dout)) return -2;
  hipStream_t st = (hipStream_t)stream;
  const size_t esz = n->dtype == SR_DTYPE_BF16 ? 2 : 4;
  const size_t act = (size_t)n->N * n->H * n->W * n->F * esz;
  const size_t blob = (size_t)n->n_idx_body * esz;
  const long act_e = (long)n->N * n->H * n->W * n->F;
  char* acts = (char*)n->acts;
  char* grads = (char*)n->grads;
  const bool pairs = net_uses_pairs(n) ||
                     (n->F == 32 && n->dtype == SR_DTYPE_BF16 && net_tiles(n) <= 384 && !getenv("SR_F32_ONE_BLOCK"));   // (as the forward)
  const bool saved = net_saves_side_images(n, true);
  const size_t side = side_image_bytes(n);
  int split = part == 0 ? 0 : n->nb_split;
  if (part != 0 && (split <= 0 || split >= n->NB || (pairs && ((n->NB - split) & 1)))) return -2;   // (pair launches must not straddle the split)
  const int b0 = part == 1 ? split : 0, b1 = part == 2 ? split : n->NB;      // blocks [b0, b1) handled by this call
  int rc;
  if (part != 2) {
    if (n->hr) {                                    // loss folded into the tail backward: no HR gradient tensor
      if ((rc = sr_tail_bwd_loss(n->out, n->hr, n->loss_kind, n->loss_gscale, n->loss_part, acts + (size_t)n->NB * act, n->x,
                                 n->mean, n->blob_tail, grads + (size_t)n->NB * act, n->part_tail, n->wgs_tail, n->N, n->H, n->W,
                                 n->F, n->R, n->dtype, stream)))
        return rc;
    } else if (n->dtype == SR_DTYPE_BF16) {         // data + weight gradients of the tail in one launch
      if ((rc = sr_tail_bwd(n->dout, acts + (size_t)n->NB * act, n->x, n->mean, n->blob_tail, grads + (size_t)n->NB * act,
                            n->part_tail, n->wgs_tail, n->N, n->H, n->W, n->F, n->R, n->dtype, stream)))
        return rc;
    } else {
      if ((rc = sr_tail_bwd_data(n->dout, grads + (size_t)n->NB * act, n->blob_tail, n->N, n->H, n->W, n->F, n->R,
                                 n->dtype, stream)))
        return rc;
      if ((rc = sr_tail_wgrad(n->dout, acts + (size_t)n->NB * act, n->x, n->mean, n->part_tail, n->wgs_tail, n->N, n->H,
                              n->W, n->F, n->R, n->dtype, stream)))
        return rc;
    }
  }
  for (int i = b1 - 1; i >= b0; --i) {
    if (pairs && i >= b0 + 1) {
      if ((rc = sr_wdsr_block2_bwd_data(acts + (size_t)(i - 1) * act, acts + (size_t)i * act, grads + (size_t)(i + 1) * act,
                                        grads + (size_t)i * act, grads + (size_t)(i - 1) * act,
                                        (char*)n->blob_body + (i - 1) * blob, (char*)n->blob_body + i * blob,
                                        n->cinit_body + (size_t)(i - 1) * n->n_idx_cinit,
                                        n->cinit_body + (size_t)i * n->n_idx_cinit,
                                        saved ? (char*)n->dtsave + (size_t)(i - 1) * side : nullptr,
                                        saved ? (char*)n->dtsave + (size_t)i * side : nullptr, n->N, n->H, n->W, n->F,
                                        n->dtype, stream)))
        return rc;
      --i;
      continue;
    }
    if (saved) {
      rc = n->F == 24
               ? launch_block_bwd_data<__bf16, 24, 144, 20>(acts + (size_t)i * act, grads + (size_t)(i + 1) * act,
                                                            grads + (size_t)i * act, (char*)n->blob_body + i * blob,
                                                            n->cinit_body + (size_t)i * n->n_idx_cinit, n->N, n->H, n->W,
                                                            (hipStream_t)stream, (char*)n->dtsave + (size_t)i * side)
               : launch_block_bwd_data<__bf16, 32, 192, 26>(acts + (size_t)i * act, grads + (size_t)(i + 1) * act,
                                                            grads + (size_t)i * act, (char*)n->blob_body + i * blob,
                                                            n->cinit_body + (size_t)i * n->n_idx_cinit, n->N, n->H, n->W,
                                                            (hipStream_t)stream, (char*)n->dtsave + (size_t)i * side);
      if (rc) return rc;
    } else if ((rc = sr_wdsr_block_bwd_data(acts + (size_t)i * act, grads + (size_t)(i + 1) * act, grads + (size_t)i * act,
                                            (char*)n->blob_body + i * blob, n->cinit_body + (size_t)i * n->n_idx_cinit,
                                            n->N, n->H, n->W, n->F, n->dtype, stream)))
      return rc;
  }
  // weight gradients of blocks [b0, b1): every per-layer pointer advanced by b0 layers
  const int nl = b1 - b0;
  const int wgs_part = nl > 0 ? (n->wgs_body * n->NB) / nl : 0;
  if (nl > 0) {
    char* a0 = acts + (size_t)b0 * act;
    char* g1 = grads + (size_t)(b0 + 1) * act;
    char* wb0 = (char*)n->blob_body + (size_t)b0 * blob;
    const float* ci0 = n->cinit_body + (size_t)b0 * n->n_idx_cinit;
    float* pa0 = n->part_a;                         // a part owns the whole partial-slab buffer while it runs, so a
    float* pb0 = n->part_b;                         // half-depth part spreads each layer over twice the workgroups
    if (saved) {
      if ((rc = sr_wdsr_block_wgrad_saved(a0, g1, (char*)n->tsave + (size_t)b0 * side, (char*)n->dtsave + (size_t)b0 * side, wb0, ci0,
                                          pa0, pb0, nl, wgs_part, n->N, n->H, n->W, n->F, n->dtype, act_e, act_e,
                                          (long)(side / esz), (long)n->n_idx_body, (long)n->n_idx_cinit, stream)))
        return rc;
    } else if ((rc = sr_wdsr_block_wgrad(a0, g1, wb0, ci0, pa0, pb0, nl, wgs_part, n->N, n->H, n->W, n->F, n->dtype, act_e,
                                         act_e, (long)n->n_idx_body, (long)n->n_idx_cinit, stream)))
      return rc;
  }
  if (part != 1)
    if ((rc = sr_head_wgrad(grads, n->x, n->mean, n->part_head, n->wgs_head, n->N, n->H, n->W, n->F, n->dtype, stream)))
      return rc;
  // slabs -> d(effective weights) -> d(flat parameters), for the layers of this part
  {
    UnpackSegs us;
    us.nseg = 0;
    int blk = 0;
    auto add = [&](const float* part_, const int* sidx, const int* dst, long off, long stride, long slab, int wgs, int cnt,
                   int reps) {
      us.s[us.nseg++] = UnpackSeg{part_, sidx, dst, off, stride, slab, wgs, cnt, reps, blk};
      blk += reps * unpack_blocks(cnt, wgs);
    };
    if (nl > 0) {
      add(n->part_a, n->ga_sidx, n->ga_dst, n->src_body_off + b0 * n->src_body_stride, n->src_body_stride, n->slab_a, wgs_part,
          n->n_ga, nl);
      add(n->part_b, n->gb_sidx, n->gb_dst, n->src_body_off + b0 * n->src_body_stride, n->src_body_stride, n->slab_b, wgs_part,
          n->n_gb, nl);
    }
    if (part != 2) add(n->part_tail, n->gt_sidx, n->gt_dst, n->src_tail_off, 0, n->slab_tail, n->wgs_tail, n->n_gt, 1);
    if (part != 1) add(n->part_head, n->gh_sidx, n->gh_dst, n->src_head_off, 0, n->slab_head, n->wgs_head, n->n_gh, 1);
    hipLaunchKernelGGL(unpack_all_kernel, dim3(blk), dim3(64 * UNPACK_Q), 0, st, n->dsrc, us);
  }
  // weight-norm backward over the table rows of this part (rows are in state_dict order: head, body.0 .., tail, skip)
  const int c0 = part == 1 ? n->chan_split : 0, c1 = part == 2 ? n->chan_split : n->n_chan;
  const int d0 = part == 1 ? n->bias_split : 0, d1 = part == 2 ? n->bias_split : n->n_bias;
  const int cb = (c1 - c0 + 3) / 4, bb = (d1 - d0 + 255) / 256;
  if (fa)
    hipLaunchKernelGGL(wn_bwd_adam_kernel, dim3(cb + bb), dim3(256), 0, st, const_cast<float*>(n->flat), n->dsrc, n->gflat, fa->m, fa->v,
                       (const int4*)n->chan_tab + c0, c1 - c0, n->bias_tab + 3 * d0, d1 - d0, cb, fa->a, fa->loss_part, fa->n_loss,
                       fa->loss_scale, fa->loss_out);
  else
    hipLaunchKernelGGL(wn_bwd_kernel, dim3(cb + bb), dim3(256), 0, st, n->flat, n->dsrc, n->gflat, (const int4*)n->chan_tab + c0,
                       c1 - c0, n->bias_tab + 3 * d0, d1 - d0, cb);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
extern "C" int sr_wdsr_net_backward_part(const sr_wdsr_net_t* n, int part, sr_stream_t stream) {
  return net_backward_part_impl(n, part, stream, nullptr);
}

extern "C" int sr_wdsr_net_backward(const sr_wdsr_net_t* n, sr_stream_t stream) { return sr_wdsr_net_backward_part(n, 0, stream); }

// table-driven parameter plumbing for callers other than the BASIC_MODEL net struct (the supernet body)
extern "C" int sr_param_pack(const float* flat, float* src, const int* chan_tab, int n_chan, const int* bias_tab,
                             const float* bias_const, int n_bias, const sr_pack_seg_t* segs, int nseg, int dtype,
                             sr_stream_t stream) {
  if (!flat || !src || !chan_tab || n_chan <= 0 || n_bias < 0 || (n_bias > 0 && (!bias_tab || !bias_const)) || !segs || nseg < 1 ||
      nseg > 4 || (dtype != SR_DTYPE_F32 && dtype != SR_DTYPE_BF16))
    return -2;
  hipStream_t st = (hipStream_t)stream;
  const int cb = (n_chan + 3) / 4, bb = (n_bias + 255) / 256;
  hipLaunchKernelGGL(wn_src_kernel, dim3(cb + bb), dim3(256), 0, st, flat, src, (const int4*)chan_tab, n_chan, bias_tab, bias_const,
                     n_bias, cb);
  PackSegs ps;
  ps.nseg = nseg;
  int blk = 0;
  for (int k = 0; k < nseg; ++k) {
    const sr_pack_seg_t& g = segs[k];
    if (!g.idx || !g.out || g.n <= 0 || g.reps <= 0) return -2;
    ps.s[k] = PackSeg{g.idx, g.out, g.src_off, g.src_stride, g.n, g.reps, g.as_float, blk};
    blk += g.reps * ((g.n + 255) / 256);
  }
  if (dtype == SR_DTYPE_BF16) hipLaunchKernelGGL((pack_all_kernel<__bf16>), dim3(blk), dim3(256), 0, st, src, ps);
  else hipLaunchKernelGGL((pack_all_kernel<float>), dim3(blk), dim3(256), 0, st, src, ps);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

extern "C" int sr_param_grads(const float* flat, float* dsrc, float* gflat, const int* chan_tab, int n_chan, const int* bias_tab,
                              int n_bias, const sr_unpack_seg_t* segs, int nseg, sr_stream_t stream) {
  if (!flat || !dsrc || !gflat || !chan_tab || n_chan <= 0 || n_bias < 0 || (n_bias > 0 && !bias_tab) || !segs || nseg < 1 || nseg > 4)
    return -2;
  hipStream_t st = (hipStream_t)stream;
  UnpackSegs us;
  us.nseg = nseg;
  int blk = 0;
  for (int k = 0; k < nseg; ++k) {
    const sr_unpack_seg_t& g = segs[k];
    if (!g.partial || !g.sidx || !g.dst || g.n <= 0 || g.reps <= 0 || g.wgs <= 0) return -2;
    us.s[k] = UnpackSeg{g.partial, g.sidx, g.dst, g.dst_off, g.dst_stride, g.slab, g.wgs, g.n, g.reps, blk};
    blk += g.reps * unpack_blocks(g.n, g.wgs);
  }
  hipLaunchKernelGGL(unpack_all_kernel, dim3(blk), dim3(64 * UNPACK_Q), 0, st, dsrc, us);
  const int cb = (n_chan + 3) / 4, bb = (n_bias + 255) / 256;
  hipLaunchKernelGGL(wn_bwd_kernel, dim3(cb + bb), dim3(256), 0, st, flat, dsrc, gflat, (const int4*)chan_tab, n_chan, bias_tab,
                     n_bias, cb);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

extern "C" int sr_nas_scalars(const float* mask_w, const float* split_w, const float* alpha, const float* alpha1,
                              const float* alpha2, int nb, int F, float* out, float* src, long src_stride, int off_mg, float* scal,
                              sr_stream_t stream) {
  if (!mask_w || !split_w || !alpha || !alpha1 || !alpha2 || !out || nb <= 0 || F < 8 || (src && (src_stride <= 0 || off_mg < 0)))
    return -2;
  if ((long)(nb + 1) * F > NAS_SCALARS_MAX) return -1;
  hipLaunchKernelGGL(nas_scalars_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask_w, split_w, alpha, alpha1, alpha2, nb,
                     F, out, src, src_stride, off_mg, scal);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

extern "C" int sr_nas_mask_grads(const float* dsrc, long ds, int off_r, int off_sxy, int off_sA, int off_sB, const float* ms,
                                 const float* p, const float* beta, int nb, int F, float* out, sr_stream_t stream) {
  if (!dsrc || !ms || !p || !beta || !out || nb <= 0 || nb > 1024 || F <= 0 || ds <= 0) return -2;
  hipLaunchKernelGGL(nas_mask_grads_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, dsrc, ds, off_r, off_sxy, off_sA, off_sB, ms,
                     p, beta, nb, F, out);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

// every block of the supernet body from one call each way
extern "C" int sr_nas_body_fwd(void* ys, void* V, const float* dwp, long dwp_bs, const void* frags, long frags_bs, const float* tabs,
                               long tabs_bs, const float* scal, long scal_bs, int nb, int N, int H, int W, int F, int dtype,
                               sr_stream_t stream) {
  if (!ys || !V || !dwp || !frags || !tabs || !scal || nb <= 0 || N <= 0 || H <= 0 || W <= 0) return -2;
  const size_t act = (size_t)N * H * W * F * (dtype == SR_DTYPE_BF16 ? 2 : 4);
  // bf16: depthwise + pointwise of a block from one launch (csrc/nas_dw_lc.h nas_block_fwd_kernel); SR_NAS_FWD_SPLIT=1 (read per
  // call, so a test can compare both routes in one process): the two kernels
  const bool fused = dtype == SR_DTYPE_BF16 && (F == 24 || F == 32) && N <= 65535 && !getenv("SR_NAS_FWD_SPLIT") && !SR_AB("SR_NAS_DW_VALU");
  for (int i = 0; i < nb; ++i) {
    char* yi = (char*)ys + (size_t)i * act;
    char* Vi = (char*)V + (size_t)i * 3 * act;
    int rc;
    if (fused) {
      typedef NasCfg<24> C;
      const int tx = (W + C::TW - 1) / C::TW;
      const dim3 g(tx * ((H + C::TH - 1) / C::TH), N);
      const long vs = (long)N * H * W * F;
      const float* dw_i = (const float*)((const char*)dwp + (size_t)i * dwp_bs);
      const __bf16* fr_i = (const __bf16*)((const char*)frags + (size_t)i * frags_bs);
      const float* tb_i = (const float*)((const char*)tabs + (size_t)i * tabs_bs);
      const float* sc_i = (const float*)((const char*)scal + (size_t)i * scal_bs);
      hipStream_t st = (hipStream_t)stream;
      if (F == 24) hipLaunchKernelGGL((nas_block_fwd_kernel<24>), g, dim3(NAS_BLOCK_FWD_THREADS), 0, st, (const __bf16*)yi, (__bf16*)Vi, (__bf16*)(yi + act), dw_i, fr_i, tb_i, sc_i, H, W, tx, vs);
      else hipLaunchKernelGGL((nas_block_fwd_kernel<32>), g, dim3(NAS_BLOCK_FWD_THREADS), 0, st, (const __bf16*)yi, (__bf16*)Vi, (__bf16*)(yi + act), dw_i, fr_i, tb_i, sc_i, H, W, tx, vs);
      SR_HIP_CHECK_LAUNCH();
      continue;
    }
    if ((rc = sr_nas_dw_fwd(yi, Vi, (const float*)((const char*)dwp + (size_t)i * dwp_bs), N, H, W, F, dtype, stream))) return rc;
    if ((rc = sr_nas_pw_fwd(yi, Vi, yi + act, (const char*)frags + (size_t)i * frags_bs, (const float*)((const char*)tabs + (size_t)i * tabs_bs),
                            (const float*)((const char*)scal + (size_t)i * scal_bs), N, H, W, F, dtype, stream)))
      return rc;
  }
  return 0;
}
extern "C" int sr_nas_body_bwd(const void* ys, const void* V, const void* g_out, void* g_tmp0, void* g_tmp1, void* GZ, const float* dwp,
                               long dwp_bs, const void* frags, long frags_bs, const float* tabs, long tabs_bs, const float* scal,
                               long scal_bs, float* part_pw, long pw_bs, float* part_dw, long dw_bs, int wgs, int nb, int N, int H,
                               int W, int F, int dtype, void** g_in, sr_stream_t stream) {
  if (!ys || !V || !g_out || !g_tmp0 || !g_tmp1 || !GZ || !dwp || !frags || !tabs || !scal || !part_pw || !part_dw || !g_in || wgs <= 0 ||
      nb <= 0 || N <= 0 || H <= 0 || W <= 0)
    return -2;
  const size_t act = (size_t)N * H * W * F * (dtype == SR_DTYPE_BF16 ? 2 : 4);
  const void* g = g_out;
  // bf16, one tile per workgroup: csrc/nas_bwd_fused.h; SR_NAS_BWD_SPLIT=1 (read per call, so a test can compare both routes in one
  // process): the separate kernels
  const int tx_f = (W + NasCfg<24>::TW - 1) / NasCfg<24>::TW, tpi_f = tx_f * ((H + NasCfg<24>::TH - 1) / NasCfg<24>::TH);
  const long vs_f = (long)N * H * W * F;
  const bool fused = dtype == SR_DTYPE_BF16 && (F == 24 || F == 32) && (long)N * tpi_f <= wgs && !getenv("SR_NAS_BWD_SPLIT") &&
                     !SR_AB("SR_NAS_WGRAD_SPLIT") && !SR_AB("SR_NAS_DW_VALU");
  for (int i = nb - 1; i >= 0; --i) {
    void* gin = (i & 1) ? g_tmp1 : g_tmp0;
    const char* yi = (const char*)ys + (size_t)i * act;
    const char* Vi = (const char*)V + (size_t)i * 3 * act;
    const float* dw_i = (const float*)((const char*)dwp + (size_t)i * dwp_bs);
    float* pdw = (float*)((char*)part_dw + (size_t)i * dw_bs);
    int rc;
    if (fused) {                                       // pointwise backward + depthwise weight gradients from one launch
      const float* tb_i = (const float*)((const char*)tabs + (size_t)i * tabs_bs);
      const float* sc_i = (const float*)((const char*)scal + (size_t)i * scal_bs);
      const __bf16* fr_i = (const __bf16*)((const char*)frags + (size_t)i * frags_bs);
      float* ppw = (float*)((char*)part_pw + (size_t)i * pw_bs);
      hipStream_t st = (hipStream_t)stream;
      if (F == 24) hipLaunchKernelGGL((nas_block_bwd_a_kernel<24>), dim3(wgs), dim3(768), 0, st, (const __bf16*)yi, (const __bf16*)Vi, (const __bf16*)g, (__bf16*)GZ, fr_i, tb_i, sc_i, dw_i, ppw, pdw, N, H, W, tx_f, tpi_f, vs_f);
      else hipLaunchKernelGGL((nas_block_bwd_a_kernel<32>), dim3(wgs), dim3(768), 0, st, (const __bf16*)yi, (const __bf16*)Vi, (const __bf16*)g, (__bf16*)GZ, fr_i, tb_i, sc_i, dw_i, ppw, pdw, N, H, W, tx_f, tpi_f, vs_f);
      SR_HIP_CHECK_LAUNCH();
      if ((rc = sr_nas_dw_bwd(yi, GZ, g, gin, dw_i, pdw, wgs, N, H, W, F, dtype, stream))) return rc;
      g = gin;
      continue;
    }
    if ((rc = sr_nas_pw_bwd(yi, Vi, g, GZ, (const char*)frags + (size_t)i * frags_bs, (const float*)((const char*)tabs + (size_t)i * tabs_bs),
                            (const float*)((const char*)scal + (size_t)i * scal_bs), (float*)((char*)part_pw + (size_t)i * pw_bs), wgs, N,
                            H, W, F, dtype, stream)))
      return rc;
    if ((rc = sr_nas_dw_bwd(yi, GZ, g, gin, dw_i, pdw, wgs, N, H, W, F, dtype, stream))) return rc;
    if ((rc = sr_nas_dw_wgrad(yi, GZ, dw_i, pdw, wgs, N, H, W, F, dtype, stream))) return rc;
    g = gin;
  }
  *g_in = const_cast<void*>(g);
  return 0;
}

// ------------------------------------------------------------------------------------------
// standalone PixelShuffle
// ------------------------------------------------------------------------------------------
extern "C" int sr_pixel_shuffle(const float* in, float* out, int N, int C, int H, int W, int r, int inverse, sr_stream_t stream) {
  if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || r < 1 || r > 8) return -2;
  const long total4 = inverse ? (long)N * C * r * r * H * ((W + 3) / 4) : (long)N * C * (H * r) * ((W * r + 3) / 4);
  const long blocks = (total4 + 255) / 256;
  if (blocks > 0x7fffffffL) return -2;
  if (inverse)
    hipLaunchKernelGGL((pixel_shuffle_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, C, H, W, r, total4);
  else
    hipLaunchKernelGGL((pixel_shuffle_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, C, H, W, r, total4);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// input pipeline
// ------------------------------------------------------------------------------------------
extern "C" int sr_patch_gather(const unsigned char* cache, const void* recs, float* lr_out, float* hr_out, int B, int P, int scale,
                               sr_stream_t stream) {
  static_assert(sizeof(PatchRec) == 40, "record layout is part of the ABI (sr_patch_rec_t)");
  if (!cache || !recs || (!lr_out && !hr_out) || B <= 0 || B > 65535 || P <= 0 || scale <= 0) return -2;
  hipStream_t st = (hipStream_t)stream;
  if (lr_out) {
    const int blocks = std::min((3 * P * P + 255) / 256, 64);
    hipLaunchKernelGGL(sr_patch_gather_kernel, dim3(blocks, B), dim3(256), 0, st, cache, (const PatchRec*)recs, lr_out, P, scale, 0);
  }
  if (hr_out) {
    const int S = P * scale, blocks = std::min((3 * S * S + 255) / 256, 256);
    hipLaunchKernelGGL(sr_patch_gather_kernel, dim3(blocks, B), dim3(256), 0, st, cache, (const PatchRec*)recs, hr_out, P, scale, 1);
  }
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// evaluation metrics
// ------------------------------------------------------------------------------------------
extern "C" int sr_psnr(const float* sr, const float* hr, float* partial, float* out, int N, int C, int H, int W, int shave,
                       int luma, int wgs, sr_stream_t stream) {
  if (!sr || !hr || !partial || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || shave < 0 || wgs <= 0 || N > 65535 ||
      (luma && C != 3 && luma != -1))
    return -2;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(wgs, N);
  if (luma == 1) hipLaunchKernelGGL((sr_sqdiff_kernel<1>), grid, dim3(256), 0, st, sr, hr, partial, C, H, W, shave);
  else if (luma == -1) hipLaunchKernelGGL((sr_sqdiff_kernel<-1>), grid, dim3(256), 0, st, sr, hr, partial, C, H, W, shave);
  else hipLaunchKernelGGL((sr_sqdiff_kernel<0>), grid, dim3(256), 0, st, sr, hr, partial, C, H, W, shave);
  const long hs = H - 2 * shave, ws = W - 2 * shave;
  const double count = (hs > 0 && ws > 0) ? (double)hs * ws * (luma == 1 ? 1 : C) : 0.0;
  hipLaunchKernelGGL(sr_psnr_finish_kernel, dim3(1), dim3(64), 0, st, partial, out, N, wgs, count);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------
// loss / optimizer epilogue
// ------------------------------------------------------------------------------------------
extern "C" int sr_adam_step(float* p, const float* g, float* m, float* v, long n, const sr_adam_t* a, const float* loss_part,
                            int n_loss, float loss_scale, float* loss_out, sr_stream_t stream) {
  if (!p || !g || !m || !v || !a || n <= 0 || (loss_out && (!loss_part || n_loss <= 0))) return -2;
  const AdamArgs aa{a->w_lerp, a->beta2, a->one_minus_beta2, a->bc2_sqrt, a->eps, a->neg_step_size};
  const long blocks = (n + 1023) / 1024;
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, aa, loss_part, n_loss,
                     loss_scale, loss_out);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
extern "C" int sr_loss_value(const float* loss_part, int n_loss, float loss_scale, float* loss_out, sr_stream_t stream) {
  if (!loss_part || !loss_out || n_loss <= 0) return -2;
  hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_part, n_loss, loss_scale, loss_out);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
extern "C" int sr_scale_by(void* y, const void* x, long n, const float* scale, int dtype, sr_stream_t stream) {
  if (!y || !x || !scale || n <= 0 || ((uintptr_t)y & 15) || ((uintptr_t)x & 15)) return -2;
  const unsigned blocks = (unsigned)((n + 2047) / 2048);
  if (dtype == SR_DTYPE_BF16)
    hipLaunchKernelGGL(scale_by_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (__bf16*)y, (const __bf16*)x, n, scale);
  else if (dtype == SR_DTYPE_F32)
    hipLaunchKernelGGL(scale_by_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)y, (const float*)x, n, scale);
  else return -1;
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
// forward + loss-folded backward + Adam from ONE call: net->hr / loss_kind / loss_gscale / loss_part set by the caller;
// `flat` is updated in place, *loss_out = loss_scale * sum(loss partials)
extern "C" int sr_wdsr_net_train_step(const sr_wdsr_net_t* n, float* m, float* v, long n_params, const sr_adam_t* a,
                                      float loss_scale, float* loss_out, sr_stream_t stream) {
  if (!n || !n->hr || !m || !v || !a || n_params <= 0) return -2;
  int rc;
  if ((rc = sr_wdsr_net_forward(n, SR_NET_SAVE_ACTS, stream))) return rc;
  // every parameter belongs to exactly one row of the weight-norm tables (the caller checks it: n_params == rows' elements),
  // so the Adam update rides on the weight-norm backward; SR_TRAIN_SEPARATE_ADAM=1: the two launches
  static const bool separate = SR_AB("SR_TRAIN_SEPARATE_ADAM");
  if (!separate && n->adam_in_wn_bwd) {
    const FusedAdam fa{m, v, AdamArgs{a->w_lerp, a->beta2, a->one_minus_beta2, a->bc2_sqrt, a->eps, a->neg_step_size}, n->loss_part,
                       n->wgs_tail, loss_scale, loss_out};
    return net_backward_part_impl(n, 0, stream, &fa);
  }
  if ((rc = sr_wdsr_net_backward(n, stream))) return rc;
  return sr_adam_step(const_cast<float*>(n->flat), n->gflat, m, v, n_params, a, n->loss_part, n->wgs_tail, loss_scale, loss_out, stream);
}

// ------------------------------------------------------------------------------------------
// probes
// ------------------------------------------------------------------------------------------
__global__ void probe_mfma_bf16_kernel(const bf16x8* a, const bf16x8* b, float* out) {
  const int l = threadIdx.x;
  f32x16 acc = zero16();
  acc = mma16<__bf16>(a[l], b[l], acc);
#pragma unroll
  for (int i = 0; i < 16; ++i) out[l * 16 + i] = acc[i];
}
__global__ void probe_mfma_f32_kernel(const f32x8* a, const f32x8* b, float* out) {
  const int l = threadIdx.x;
  f32x16 acc = zero16();
  acc = mma16<float>(a[l], b[l], acc);
#pragma unroll
  for (int i = 0; i < 16; ++i) out[l * 16 + i] = acc[i];
}
__global__ void probe_tr_kernel(const __bf16* img, int n, const int* off, bf16x4* out) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[8192];
  for (int i = threadIdx.x; i < n && i < 8192; i += 64) lds[i] = img[i];
  __syncthreads();
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  out[threadIdx.x] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lds + off[threadIdx.x]));
}
__global__ void probe_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

extern "C" int sr_probe_mfma_bf16(const void* a, const void* b, float* out, sr_stream_t stream) {
  hipLaunchKernelGGL(probe_mfma_bf16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16x8*)a,
                     (const bf16x8*)b, out);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
extern "C" int sr_probe_mfma_f32(const float* a, const float* b, float* out, sr_stream_t stream) {
  hipLaunchKernelGGL(probe_mfma_f32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const f32x8*)a,
                     (const f32x8*)b, out);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
extern "C" int sr_probe_tr_read(const void* img, int n, const int* off, void* out, sr_stream_t stream) {
  if (n > 8192) return -2;
  hipLaunchKernelGGL(probe_tr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const __bf16*)img, n, off,
                     (bf16x4*)out);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
extern "C" int sr_probe_copy(const void* src, void* dst, size_t n_bytes, sr_stream_t stream) {
  if (n_bytes % 16) return -2;
  hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src,
                     (u32x4*)dst, n_bytes / 16);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

// launch-floor probe: `reps` dependent launches of a kernel that does nothing but occupy `threads` threads and
// `lds_bytes` of LDS per workgroup (so one workgroup per CU when large) and write one word per workgroup
__global__ void probe_floor_kernel(unsigned* out) {
  extern __shared__ unsigned dyn_lds[];
  if (threadIdx.x == 0) {
    dyn_lds[0] = blockIdx.x;
    out[blockIdx.x + gridDim.x * blockIdx.y] = dyn_lds[0];
  }
}
extern "C" int sr_probe_launch_floor(void* out, int gx, int gy, int threads, int lds_bytes, int reps,
                                     sr_stream_t stream) {
  if (!out || gx <= 0 || gy <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 4 || lds_bytes > 160 * 1024) return -2;
  if (hipFuncSetAttribute((const void*)probe_floor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return -3;
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL(probe_floor_kernel, dim3(gx, gy), dim3(threads), lds_bytes, (hipStream_t)stream, (unsigned*)out);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}

// same chain captured once in a hipGraph and replayed: separates the host's launch rate from the GPU's
// dependent-dispatch cost.  Synchronises; returns microseconds per kernel in *us_per_launch.
extern "C" int sr_probe_launch_floor_graph(void* out, int gx, int gy, int threads, int lds_bytes, int reps, int iters,
                                           float* us_per_launch, float* host_us_per_graph) {
  if (!out || !us_per_launch || gx <= 0 || gy <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 4 ||
      lds_bytes > 160 * 1024 || reps <= 0 || iters <= 0)
    return -2;
  if (hipFuncSetAttribute((const void*)probe_floor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return -3;
  hipStream_t s;
  hipGraph_t g;
  hipGraphExec_t ge;
  hipEvent_t e0, e1;
  int rc = -3;
  if (hipStreamCreate(&s) != hipSuccess) return -3;
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
    for (int i = 0; i < reps; ++i)
      hipLaunchKernelGGL(probe_floor_kernel, dim3(gx, gy), dim3(threads), lds_bytes, s, (unsigned*)out);
    if (hipStreamEndCapture(s, &g) == hipSuccess) {
      if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int i = 0; i < iters; ++i) hipGraphLaunch(ge, s);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        *us_per_launch = ms * 1e3f / ((float)reps * iters);
        if (host_us_per_graph) *host_us_per_graph = (float)(((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) * 1e-3 / iters);
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        hipGraphExecDestroy(ge);
        rc = 0;
      }
      hipGraphDestroy(g);
    }
  }
  hipStreamDestroy(s);
  return rc;
}

// debug: route the in-kernel time stamps of the diagnostic build to `buf` ([n_workgroups][16 waves][16 stamps][2] u64; NULL = off).
// Workgroups beyond n_workgroups do not stamp (a stamped launch with a larger grid must not write past the buffer).
// The product library carries no stamp code: there the call reports "unsupported".
extern "C" int sr_debug_set_stamps(void* buf, long n_workgroups) {
#ifdef SR_DEBUG_STAMPS
  if (buf && n_workgroups <= 0) return -2;
  unsigned long long* p = (unsigned long long*)buf;
  const unsigned long long n = buf ? (unsigned long long)n_workgroups : 0ull;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_sr_stamp_wgs), &n, sizeof(n)) != hipSuccess) return -3;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_sr_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -3;
#else
  (void)buf; (void)n_workgroups;
  return -1;
#endif
}

// ---- SPyNet's 7x7 convolutions (csrc/spynet_conv.h) ----
template <int CIN, int COUT, bool RELU, bool OUT_F32>
static int launch_conv7(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, hipStream_t st) {
  typedef Conv7Cfg<CIN, COUT> K;
  const int tiles_x = (W + K::TW - 1) / K::TW, tiles_y = (H + K::TH - 1) / K::TH;
  hipLaunchKernelGGL((conv7_fwd_kernel<CIN, COUT, RELU, OUT_F32>), dim3(tiles_x * tiles_y, N), dim3(512), 0, st, (const __bf16*)x,
                     (const __bf16*)w, bias, y, H, W, tiles_x);
  SR_HIP_CHECK_LAUNCH();
  return 0;
}
extern "C" int sr_conv7_fwd(const void* x, const void* wpacked, const float* bias, void* y, int N, int H, int W, int CIN, int COUT,
                            int relu, int out_f32, sr_stream_t stream) {
  if (!x || !wpacked || !bias || !y || N <= 0 || H <= 0 || W <= 0 || N > 65535) return -2;
  hipStream_t st = (hipStream_t)stream;
  if (CIN == 8 && COUT == 32 && relu && !out_f32) return launch_conv7<8, 32, true, false>(x, wpacked, bias, y, N, H, W, st);
  if (CIN == 32 && COUT == 64 && relu && !out_f32) return launch_conv7<32, 64, true, false>(x, wpacked, bias, y, N, H, W, st);
  if (CIN == 64 && COUT == 32 && relu && !out_f32) return launch_conv7<64, 32, true, false>(x, wpacked, bias, y, N, H, W, st);
  if (CIN == 32 && COUT == 16 && relu && !out_f32) return launch_conv7<32, 16, true, false>(x, wpacked, bias, y, N, H, W, st);
  if (CIN == 16 && COUT == 2 && !relu && out_f32) return launch_conv7<16, 2, false, true>(x, wpacked, bias, y, N, H, W, st);
  return -1;
}
