/* sr_hotpath.h -- C ABI of libsr_hotpath.so: the MI355X (gfx950) kernels behind the
 * WDSR-B / BasicVSR super-resolution hot path of zhuzhui-2000/mobilesuperresolution.
 *
 * The reference has no FFI of its own: its boundary is the nn.Module surface
 * (models/__init__.py:31-32 get_model, models/basic_wdsr_b.py:85-93 BASIC_MODEL.forward).  This
 * header is the seam *beneath* that surface: the Python mirror in mobilesuperresolution_amd/models
 * binds these entry points with ctypes (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed from the caller (PyTorch allocator); nothing is
 *     allocated, freed or synchronised inside; all work is enqueued on `stream` (a hipStream_t).
 *   - activations are NHWC ("pixel-major, channels innermost"); dtype: 0 = float32 (exact-fp32 MFMA,
 *     parity mode), 1 = bfloat16 storage with fp32 accumulation (throughput mode).
 *   - weights arrive as "packed fragment blobs" built on the host by
 *     mobilesuperresolution_amd/packing.py from the effective (weight-normalised) tensors.
 *   - return value: 0 on success, a hipError_t value on a launch error, -1 unsupported geometry,
 *     -2 bad argument.  Entry points are re-entrant and hold no global mutable state (autograd calls
 *     the backward ones from its own thread).
 */
#ifndef SR_HOTPATH_H
#define SR_HOTPATH_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sr_stream_t; /* hipStream_t */

#define SR_DTYPE_F32 0
#define SR_DTYPE_BF16 1

/* ABI version of this header (bumped on any signature change). */
int sr_abi_version(void);

/* Fused residual block forward.  Replaces Block.forward, models/basic_wdsr_b.py:142-144 (body of
 * :108-138): y = conv3x3(conv1x1(relu(conv1x1(x)))) + x on an N x H x W x F NHWC tensor.
 * Supported (F, E, L): (24,144,20), (32,192,26).  wblob / cinit: packing.block_fwd_tables(). */
int sr_wdsr_block_fwd(const void* x, void* y, const void* wblob, const float* cinit,
                      int N, int H, int W, int F, int dtype, sr_stream_t stream);

/* Two consecutive residual blocks in one launch (bf16, F = 24 or 32; -1 otherwise): x -> ya (block A's
 * output, kept because backward needs every block input; NULL = do not store) -> yb.  Equal to two sr_wdsr_block_fwd
 * calls up to the fp32 summation order of the 3x3 conv (dense-K form: the residual enters the sum first); exists because a
 * single block launch at batch 32 is bound by its fixed costs.  (= sr_wdsr_fwd_rs with nblk = 2.) */
int sr_wdsr_block2_fwd(const void* x, void* ya, void* yb, const void* wblob_a, const void* wblob_b,
                       const float* cinit_a, const float* cinit_b, void* tsave_a, void* tsave_b, int N, int H, int W,
                       int F, int dtype, sr_stream_t stream);
/* tsave_a / tsave_b (NULL = off): keep each block's t = conv1x1(relu(conv1x1(x))) (the 3x3 conv's input,
 * basic_wdsr_b.py:131-137) for the core pixels, tile-local [N][tiles][288][24] bf16, for
 * sr_wdsr_block_wgrad_saved. */

/* Backward-data of two consecutive blocks in one launch (bf16, F = 24 or 32; -1 otherwise): block A feeds
 * block B.  xa / xb = the blocks' inputs, dyb = gradient at B's output; writes dxb (= gradient at A's
 * output, which the weight-gradient kernels read) and dxa.  F = 32: bit-identical to two sr_wdsr_block_bwd_data calls
 * (csrc/wdsr_bwd_pair_lds.h); F = 24: equal to them up to fp32 summation order (csrc/wdsr_bwd_rs.h: the skip term enters the sum
 * first). */
int sr_wdsr_block2_bwd_data(const void* xa, const void* xb, const void* dyb, void* dxb, void* dxa,
                            const void* wblob_a, const void* wblob_b, const float* cinit_a, const float* cinit_b,
                            void* dtsave_a, void* dtsave_b, int N, int H, int W, int F, int dtype, sr_stream_t stream);
/* dtsave_a / dtsave_b (NULL = off): keep each block's dt = conv3x3^T(dy), same layout as tsave.
 *
 * Weight gradients of `layers` blocks from the saved t / dt images instead of recomputing them (bf16 only;
 * -1 otherwise).  Same slabs as sr_wdsr_block_wgrad.  side_ls = elements between two blocks' saved images
 * ([N][tiles][288][LP], LP = 24 for 24 units, 32 for 32 units). */
int sr_wdsr_block_wgrad_saved(const void* x, const void* dy, const void* tsave, const void* dtsave,
                              const void* wblob, const float* cinit, float* partial_a, float* partial_b, int layers,
                              int wgs, int N, int H, int W, int F, int dtype, long x_ls, long dy_ls, long side_ls,
                              long w_ls, long c_ls, sr_stream_t stream);

/* Forward of nblk = 1 or 2 consecutive residual blocks, role-specialised kernels of csrc/wdsr_fwd_rs.h /
 * wdsr_fwd_stream.h (bf16; -1 otherwise).  F = 24: every wave keeps the weights of its current phase in registers, x / weights
 * are staged by LDS-DMA; from 256 whole images of width 48 on, the streaming kernel.  F = 32: sixteen waves, weights read from
 * LDS at use.  The 3x3 conv runs in its dense-K form.  Same arguments as sr_wdsr_block_fwd (nblk = 1: x -> yb, the *_b / ya
 * arguments unused) and sr_wdsr_block2_fwd (nblk = 2); results equal to theirs up to fp32 summation order, and bit-identical
 * across this entry point's kernels (one- and two-block launches, tile, persistent and streaming forms). */
int sr_wdsr_fwd_rs(const void* x, void* ya, void* yb, const void* wblob_a, const void* wblob_b, const float* cinit_a,
                   const float* cinit_b, void* tsave_a, void* tsave_b, int nblk, int N, int H, int W, int F, int dtype,
                   sr_stream_t stream);
/* measurement aid: `reps` back-to-back launches ping-ponging x <-> yb; tsave_a / tsave_b as in sr_wdsr_fwd_rs (NULL: the
 * inference variant; both given: the variant a training step launches, which also keeps the t images) */
int sr_wdsr_fwd_rs_repeat(void* x, void* ya, void* yb, const void* wblob_a, const void* wblob_b, const float* cinit_a,
                          const float* cinit_b, void* tsave_a, void* tsave_b, int nblk, int N, int H, int W, int F, int dtype,
                          int reps, sr_stream_t stream);

/* Measurement aid for bench.py's roofline leg: `reps` back-to-back launches of the same forward kernel,
 * ping-ponging x <-> y, so that HIP events around the call measure the kernel and not the host. */
int sr_wdsr_block_fwd_repeat(void* x, void* y, const void* wblob, const float* cinit,
                             int N, int H, int W, int F, int dtype, int reps, sr_stream_t stream);
/* Same for the two-block kernel: launch i reads x (even i) or yb (odd i), writes ya and the other one. */
int sr_wdsr_block2_fwd_repeat(void* x, void* ya, void* yb, const void* wblob_a, const void* wblob_b,
                              const float* cinit_a, const float* cinit_b, int N, int H, int W, int F, int dtype,
                              int reps, sr_stream_t stream);


/* Fused residual block backward w.r.t. its input.  Replaces autograd's backward of Block.forward
 * (models/basic_wdsr_b.py:142-144): dx = dy + W1^T[1(h>0) * W2^T conv3x3^T(dy)], h recomputed from x.
 * wblob / cinit: packing.block_tables() (forward sections first). */
int sr_wdsr_block_bwd_data(const void* x, const void* dy, void* dx, const void* wblob, const float* cinit,
                           int N, int H, int W, int F, int dtype, sr_stream_t stream);

/* Weight/bias gradients of `layers` residual blocks in one call.  Layer i reads x + i*x_ls,
 * dy + i*dy_ls, wblob + i*w_ls, cinit + i*c_ls (strides in elements).  Every workgroup writes one
 * partial slab: partial_a[layers][wgs_per_layer][slab_a] (dW1, dW2, db1, db2) and
 * partial_b[layers][wgs_per_layer][slab_b] (dW3, db3) in accumulator layout; the caller sums over
 * the workgroup axis and gathers with packing.block_grad_tables().  sr_wdsr_block_slab_sizes returns
 * the slab lengths (floats). */
int sr_wdsr_block_wgrad(const void* x, const void* dy, const void* wblob, const float* cinit,
                        float* partial_a, float* partial_b, int layers, int wgs_per_layer,
                        int N, int H, int W, int F, int dtype,
                        long x_ls, long dy_ls, long w_ls, long c_ls, sr_stream_t stream);
int sr_wdsr_block_slab_sizes(int F, int* slab_a, int* slab_b);

/* Head conv forward.  Replaces `x - image_mean` + self.head(x), models/basic_wdsr_b.py:86-87:
 * x NCHW fp32 [N,3,H,W] in [0,1] -> y NHWC [N,H,W,F].  wblob: packing.ends_tables()["head"]. */
int sr_head_fwd(const float* x_nchw, void* y, const void* wblob, float mean, int N, int H, int W, int F,
                int dtype, sr_stream_t stream);
/* Fused tail: self.tail(y) + self.skip(x - mean) -> PixelShuffle(R) -> + mean, models/basic_wdsr_b.py:90-92.
 * feat NHWC [N,H,W,F], x NCHW fp32, out NCHW fp32 [N,3,R*H,R*W].  R in {2,3,4}.
 * wblob: packing.ends_tables()["tail"] (forward section, then the backward-data section). */
int sr_tail_fwd(const void* feat, const float* x_nchw, float* out, const void* wblob, float mean,
                int N, int H, int W, int F, int R, int dtype, sr_stream_t stream);
/* d(loss)/d(feat) of the tail from d(loss)/d(out) (NCHW fp32, HR). */
int sr_tail_bwd_data(const float* dout, void* dfeat, const void* wblob, int N, int H, int W, int F, int R,
                     int dtype, sr_stream_t stream);
/* Weight/bias gradients of tail + skip: partial[wgs][14*NT*1024] floats in accumulator layout
 * (packing.ends_grad_tables()["tail"]); NT = ceil(3 R^2 / 32). */
int sr_tail_wgrad(const float* dout, const void* feat, const float* x_nchw, float mean, float* partial,
                  int wgs, int N, int H, int W, int F, int R, int dtype, sr_stream_t stream);
/* sr_tail_bwd_data + sr_tail_wgrad in one launch (bf16 only; -1 otherwise): reads the HR gradient once.
 * Same outputs, bit for bit, as the two calls with the same `wgs`. */
int sr_tail_bwd(const float* dout, const void* feat, const float* x, float mean, const void* wblob, void* dfeat,
                float* partial, int wgs, int N, int H, int W, int F, int R, int dtype, sr_stream_t stream);
/* Weight/bias gradient of the head conv from dy0 = d(loss)/d(head output), NHWC: partial[wgs][3*1024]. */
int sr_head_wgrad(const void* dy0, const float* x_nchw, float mean, float* partial, int wgs,
                  int N, int H, int W, int F, int dtype, sr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * BasicVSR propagation trunk: 3x3 convolutions of ConvResidualBlocks / ResidualBlockNoBN,
 * models/basicvsr_arch.py:108-147.  NHWC activations, CI in {24, 32} input channels (the 27-channel
 * concat of frame + state is stored zero-padded to 32), 24 output channels.  act: 0 none, 1 ReLU,
 * 2 LeakyReLU(0.1).  wblob: packing.c3_tables() (forward fragments, then backward-data fragments).
 * ------------------------------------------------------------------------------------------------ */
/* y = act(conv3x3(x) + b) (+ res when res != NULL; only with act == 0). */
int sr_c3_fwd(const void* x, const void* res, void* y, const void* wblob, int N, int H, int W, int CI, int act,
              int dtype, sr_stream_t stream);
/* dx = conv3x3^T(dA * act'(A)) (+ add when add != NULL; only with act == 1).  A = saved post-activation
 * output (ignored for act == 0). */
int sr_c3_bwd_data(const void* dA, const void* A, const void* add, void* dx, const void* wblob, int N, int H,
                   int W, int CI, int act, int dtype, sr_stream_t stream);
/* weight/bias gradient slabs partial[wgs][9*1024] (accumulator layout, packing.c3_grad_table()). */
int sr_c3_wgrad(const void* x, const void* dA, const void* A, float* partial, int wgs, int N, int H, int W,
                int CI, int act, int dtype, sr_stream_t stream);

/* The first conv's input gathered on the fly: what the reference builds per frame and direction as
 * `flow_warp(feat_prop, flow.permute(0,2,3,1))` then `torch.cat([x_i, feat_prop], dim=1)` (models/basicvsr_arch.py:74-76,
 * 85-87; mvvsr_arch.py:79-81,90-92).  frame [N][3][H][W] fp32 (batch stride frame_bs elements); state [N][H][W][24] in the
 * hot dtype = the previous call's a_nb (NULL: zero state, the first frame of a direction); flow [N][2][H][W] fp32, x then
 * y displacement, batch stride flow_bs (NULL: no warp).  Backward only: flow_bound = device scalar >= max |flow| (read by
 * the kernel, no host sync; it sizes the gather window of the atomics-free d state); dstate [N][H][W][24] hot dtype (NULL:
 * not wanted); dflow [N][2][H][W] fp32 with batch stride dflow_bs (NULL: not wanted). */
typedef struct sr_c3_warp {
  const float* frame; long frame_bs;
  const void* state;
  const float* flow; long flow_bs;
  const float* flow_bound;
  void* dstate;
  float* dflow; long dflow_bs;
  void* x0_save;   /* optional [N][H][W][32] hot dtype: forward writes the gathered input there, backward reads it for the
                      first conv's weight gradient instead of gathering again */
} sr_c3_warp_t;

/* Slabs -> gradient of the trunk's flat parameter inside sr_c3_trunk_bwd (NULL: the caller reduces `parts` itself):
 * gflat[off_k + i] = sum_w parts[k][w][sidx[i]], conv 0 through sidx0 (n0 elements), convs 1..2nb through sidx1 (n1
 * elements each, consecutive in gflat from element n0 on). */
typedef struct sr_c3_unpack {
  const int* sidx0; const int* dst0; int n0;
  const int* sidx1; const int* dst1; int n1;
  float* gflat;
} sr_c3_unpack_t;

/* The whole propagation trunk, ConvResidualBlocks.forward (models/basicvsr_arch.py:108-147), from one call.
 * Input: EITHER x0 [N,H,W,ci0] (ci0 = 32: the 27-channel concat zero-padded, or 24) OR warp (frame, state, flow; ci0 = 32)
 * -- exactly one of the two is non-NULL.  acts [(nb+1)][N,H,W,24] receives a_0..a_nb (a_nb = output), mids [nb][N,H,W,24]
 * the post-ReLU conv1 outputs; blob = every conv's packed weights in one buffer, conv k (0 = first conv, 1+2i / 2+2i =
 * conv1 / conv2 of block i) at ELEMENT offset blob_off[k] (host array).
 * TWO TRUNKS IN ONE CALL (n_dir > 0; the two time directions of a BasicVSR frame step, basicvsr_arch.py:67-88, are independent):
 * images [0, n_dir) of the batch run through the trunk whose packed weights sit at `blob`, images [n_dir, N) through the one
 * `blob_dir_stride` ELEMENTS behind it (same layout, same blob_off); n_dir = 0: one trunk. */
int sr_c3_trunk_fwd(const void* x0, const sr_c3_warp_t* warp, void* acts, void* mids, const void* blob,
                    const long* blob_off, int nb, int N, int H, int W, int ci0, int dtype, int n_dir, long blob_dir_stride,
                    sr_stream_t stream);
/* Its backward.  ga [(nb+1)][N,H,W,24]: the caller writes d(loss)/d(a_nb) into slot nb, the call fills the other
 * slots; gt [nb][...] scratch (gradients at the post-ReLU points); parts [(1+2nb)][wgs][9*1024] fp32 weight-gradient
 * slabs per conv (layout packing.c3_tables "grad"); dx0 (may be NULL unless warp->dstate / dflow is asked for) =
 * gradient w.r.t. the first conv's 32-channel input.
 * n_dir > 0 (two trunks, as in sr_c3_trunk_fwd): wgs must be even -- the first wgs / 2 slabs of every conv sum over the first trunk's
 * images, the others over the second's -- and unpack->gflat receives the two trunks' gradients one behind the other. */
int sr_c3_trunk_bwd(const void* x0, const sr_c3_warp_t* warp, const void* acts, const void* mids, void* ga, void* gt,
                    const void* blob, const long* blob_off, float* parts, void* dx0, const sr_c3_unpack_t* unpack, int nb,
                    int wgs, int N, int H, int W, int ci0, int dtype, int n_dir, long blob_dir_stride, sr_stream_t stream);

/* Training patches cut on device from a resident uint8 cache.  Replaces, per patch,
 * ImageSuperResolutionDataset._sample_patch + _augment + to_tensor, datasets/_isr.py:68-121.  cache: every LR and HR image
 * as PIL lays them out (H x W x 3 uint8), back to back; recs[B]: one record per patch, drawn on the host in the
 * reference's RNG order; lr_out [B][3][P][P], hr_out [B][3][P*scale][P*scale] fp32 in [0, 1] (either may be NULL). */
typedef struct sr_patch_rec {
  long lr_off, hr_off;      /* byte offsets of the two images in the cache */
  int lr_w, hr_w;           /* their widths in pixels */
  int x, y;                 /* LR crop origin: row x, column y (the reference's names) */
  int flags;                /* 1: flip rows, 2: flip columns, 4: swap axes -- applied in this order */
  int pad_;
} sr_patch_rec_t;
int sr_patch_gather(const unsigned char* cache, const void* recs, float* lr_out, float* hr_out, int B, int P, int scale,
                    sr_stream_t stream);

/* Evaluation metrics on device: psnr (luma = 0; common/metrics.py:10-19: 8-bit quantised sr) and psnr_y (luma = 1 for
 * 3-channel images; :22-38: clamped but NOT quantised -- the reference drops its quantised copy -- with the luma filter
 * on the difference; luma = -1: the same without the filter, what the reference does when dim 1 is not 3).  sr, hr
 * [N][C][H][W] fp32; partial [N * wgs] scratch; out[0] = sum over the batch of -10 log10(mse) (the reference sums). */
int sr_psnr(const float* sr, const float* hr, float* partial, float* out, int N, int C, int H, int W, int shave, int luma,
            int wgs, sr_stream_t stream);

/* flow_warp, models/spynet_arch.py:98-129 (bilinear, zeros padding, align_corners=True): x, out NCHW fp32;
 * flow (N,H,W,2).  Backward: dx (zero-filled by the caller, may be NULL) and dflow (may be NULL). */
int sr_flow_warp_fwd(const float* x, const float* flow, float* out, int N, int C, int H, int W, sr_stream_t stream);
int sr_flow_warp_bwd(const float* x, const float* flow, const float* gout, float* dx, float* dflow, int N, int C,
                     int H, int W, sr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * NAS supernet block: Split_Block.forward_body (models/wdsr_b.py:482-496) with Conv_sep branches
 * (:375-402), the BinaryConv2d masks (models/ops.py:7-43) and the hard skip/keep gate of
 * MyAggregationLayer.forward (:517-546):  y = mg*yin + beta2 * ms * sum_k p_k relu(pw_k(relu(dw_k(mg*ms*yin)))).
 * Tables/fragments: mobilesuperresolution_amd/packing.py nas_tables().  V / GZ: [3][N][H][W][F].
 * ------------------------------------------------------------------------------------------------ */
int sr_nas_dw_fwd(const void* yin, void* V, const float* dwp, int N, int H, int W, int F, int dtype, sr_stream_t stream);
int sr_nas_pw_fwd(const void* yin, const void* V, void* y, const void* frags, const float* tabs, const float* scal,
                  int N, int H, int W, int F, int dtype, sr_stream_t stream);
/* partial[wgs][3*(1024+64)+4]: per branch dWpw tile | dbp[32] | r[32]; then sum(gy*mg*yin). */
int sr_nas_pw_bwd(const void* yin, const void* V, const void* gy, void* GZ, const void* frags, const float* tabs,
                  const float* scal, float* partial, int wgs, int N, int H, int W, int F, int dtype,
                  sr_stream_t stream);
/* partial[wgs][88*32]: dWdw[83 taps][32] | dbd[3][32] | sum(g_br*mg*yin)[32] | sum(g_x*yin)[32]. */
int sr_nas_dw_bwd(const void* yin, const void* GZ, const void* gy, void* gyin, const float* dwp, float* partial,
                  int wgs, int N, int H, int W, int F, int dtype, sr_stream_t stream);
/* Depthwise weight gradients dWdw[83 taps][32] of the same slab (same `wgs`), on the matrix cores; call after
 * sr_nas_dw_bwd on the same stream (it fills the part of each slab that sr_nas_dw_bwd leaves untouched). */
int sr_nas_dw_wgrad(const void* yin, const void* GZ, const float* dwp, float* partial, int wgs, int N, int H, int W,
                    int F, int dtype, sr_stream_t stream);

/* Every block of the supernet body from one call each way (the per-op entry points above, looped in C: 2 launches per
 * block forward, 3 backward).  ys [(nb+1)][N][H][W][F] (slot 0 = input, slot nb = output); V [nb][3][N][H][W][F];
 * per-block tables at BYTE strides: dwp (dwp_bs), frags (frags_bs), tabs (tabs_bs), scal (scal_bs).  Backward: g_out =
 * gradient at ys[nb]; g_tmp[2] two activation-sized scratch buffers; GZ [3][N][H][W][F] scratch; part_pw / part_dw
 * [nb][wgs][slab] at byte strides pw_bs / dw_bs.  *g_in receives the pointer (g_tmp[0] or g_tmp[1]) that holds the gradient at ys[0]. */
/* 0/1 masks, gates and latency terms of a supernet step, values only (models/ops.py:33-43, wdsr_b.py:517-534,
 * speed_estimator.py:57-76).  split_w (nb, F), alpha (nb, 3), alpha1 / alpha2 (nb).
 * out: mask_hard[F] | c_mask | ms_hard[nb][F] | c_split[nb] | speed_curr[nb] | gates[nb][2]  (F + 1 + nb (F + 4) floats).
 * One workgroup; -1 if (nb + 1) F > 4096. */
int sr_nas_scalars(const float* mask_w, const float* split_w, const float* alpha, const float* alpha1, const float* alpha2,
                   int nb, int F, float* out, float* src, long src_stride, int off_mg, float* scal, sr_stream_t stream);
/* optional outputs of sr_nas_scalars (NULL: skipped): `src` -- the mask columns mg | ms | mg ms | 0 | 1 (3 F + 2 floats at column
 * off_mg) of nb operand-source rows of src_stride floats (sr_param_pack gathers them); `scal` -- nb x {softmax(alpha)[3], gate2}.
 * sr_nas_mask_grads: the gradients of the masks / gates / branch weights from the sums the block kernels left in d(source)
 * (columns off_r: r[3][F], off_sxy, off_sA: [F], off_sB: [F] of rows of `ds` floats), ms (nb, F), p (nb, 3), beta (nb, 2):
 * out = g_p[nb][3] | g_beta[nb][2] | g_ms[nb][F] | g_mg[F]   (wdsr_b.py:517-546 differentiated) */
int sr_nas_mask_grads(const float* dsrc, long ds, int off_r, int off_sxy, int off_sA, int off_sB, const float* ms, const float* p,
                      const float* beta, int nb, int F, float* out, sr_stream_t stream);
int sr_nas_body_fwd(void* ys, void* V, const float* dwp, long dwp_bs, const void* frags, long frags_bs, const float* tabs,
                    long tabs_bs, const float* scal, long scal_bs, int nb, int N, int H, int W, int F, int dtype,
                    sr_stream_t stream);
int sr_nas_body_bwd(const void* ys, const void* V, const void* g_out, void* g_tmp0, void* g_tmp1, void* GZ, const float* dwp,
                    long dwp_bs, const void* frags, long frags_bs, const float* tabs, long tabs_bs, const float* scal,
                    long scal_bs, float* part_pw, long pw_bs, float* part_dw, long dw_bs, int wgs, int nb, int N, int H, int W,
                    int F, int dtype, void** g_in, sr_stream_t stream);

/* Parameter plumbing of a table-described model (the supernet body; the same kernels sr_wdsr_net_forward / _backward run on
 * BASIC_MODEL's tables): weight-norm of the flat fp32 parameters into `src` (reference: torch.nn.utils.weight_norm around
 * every conv, models/wdsr_b.py:375-402), then up to four gathers of `src` into the packed operand tables the compute
 * kernels read; and the way back: workgroup slabs summed and scattered into `dsrc` (laid out like `src`, plus whatever extra
 * columns the tables name), then the weight-norm backward into `gflat` (laid out like `flat`; entries no table row names are
 * left untouched).  chan_tab: int[n_chan][4] = {v_off, g_off, K, dst_off}; bias_tab: int[n_bias][3] = {src_a, src_b | -1, dst}. */
typedef struct { const int* idx; void* out; long src_off, src_stride; int n, reps, as_float; } sr_pack_seg_t;
typedef struct { const float* partial; const int* sidx; const int* dst; long dst_off, dst_stride, slab; int wgs, n, reps; } sr_unpack_seg_t;
int sr_param_pack(const float* flat, float* src, const int* chan_tab, int n_chan, const int* bias_tab, const float* bias_const,
                  int n_bias, const sr_pack_seg_t* segs, int nseg, int dtype, sr_stream_t stream);
int sr_param_grads(const float* flat, float* dsrc, float* gflat, const int* chan_tab, int n_chan, const int* bias_tab,
                   int n_bias, const sr_unpack_seg_t* segs, int nseg, sr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-network entry points: everything BASIC_MODEL.forward (models/basic_wdsr_b.py:85-93) and its
 * autograd backward do, as ONE call each.  The caller (mobilesuperresolution_amd/models) owns every
 * buffer; `sr_wdsr_net_t` only carries device pointers, table sizes and the geometry.
 * Parameters live in one flat fp32 buffer (layout: mobilesuperresolution_amd/layout.py).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int F, NB, R, dtype, N, H, W;
  float mean;
  /* parameters, their gradient, weight-norm tables */
  const float* flat; float* gflat;
  const int* chan_tab; int n_chan;              /* int4 {v_off, g_off, K, dst_off} per output channel */
  const int* bias_tab; const float* bias_const; int n_bias;   /* int3 {src_a, src_b, dst} per bias element */
  /* canonical effective-weight vectors (src) and their gradient (dsrc): head | body[NB] | tail */
  float* src; float* dsrc;
  long src_head_off, src_body_off, src_body_stride, src_tail_off;
  /* fragment packing tables and packed blobs */
  const int* idx_head; int n_idx_head;
  const int* idx_body; int n_idx_body;
  const int* idx_cinit; int n_idx_cinit;
  const int* idx_tail; int n_idx_tail;
  void* blob_head; void* blob_body; float* cinit_body; void* blob_tail;
  /* gradient slabs written by the weight-gradient kernels and their gather tables {slab idx, dst idx} */
  float* part_a; float* part_b; float* part_tail; float* part_head;
  int wgs_body, wgs_tail, wgs_head;
  int slab_a, slab_b, slab_tail, slab_head;
  const int* ga_sidx; const int* ga_dst; int n_ga;
  const int* gb_sidx; const int* gb_dst; int n_gb;
  const int* gt_sidx; const int* gt_dst; int n_gt;
  const int* gh_sidx; const int* gh_dst; int n_gh;
  /* activations: x NCHW fp32; acts/grads [(NB+1)][N][H][W][F] (acts may be 2 ping-pong slots when
   * save_acts == 0); out / dout NCHW fp32 [N][3][R*H][R*W] */
  const float* x; void* acts; void* grads; float* out; const float* dout;
  /* optional (bf16; NULL = recompute in backward): t and dt of every block, [NB][N][tiles][288][LP] */
  void* tsave; void* dtsave;
  /* optional loss fold (hr != NULL): backward reads `out` and `hr` (NCHW fp32 HR target) instead of `dout` and forms
   * the L1 (loss_kind 1, pretrain.py:73) or Charbonnier (2, train_video_superresolution.py:43-53) gradient on the
   * fly, loss_gscale = upstream gradient / numel; loss_part[wgs_tail] receives the per-workgroup loss sums */
  const float* hr; int loss_kind; float loss_gscale; float* loss_part;
  /* optional two-part backward (sr_wdsr_net_backward_part): first block of the late half, and the rows of chan_tab /
   * bias_tab where the late parameters (body[nb_split..], tail, skip) begin */
  int nb_split, chan_split, bias_split;
  /* sr_wdsr_net_train_step only: 1 = every parameter belongs to exactly one row of chan_tab / bias_tab, so the Adam update is
   * applied by the weight-norm backward's launch (one launch less); 0 = separate sr_adam_step pass */
  int adam_in_wn_bwd;
} sr_wdsr_net_t;

/* weight-norm + packing + head + NB fused blocks + fused tail.  flags: SR_NET_SAVE_ACTS keeps every block input
 * (needed by sr_wdsr_net_backward), otherwise two slots ping-pong (inference); SR_NET_WEIGHTS_PACKED skips the
 * weight-norm + packing launches (the caller guarantees the packed blobs in `net` are those of `flat`, e.g. repeated
 * inference with unchanged parameters). */
#define SR_NET_SAVE_ACTS 1
#define SR_NET_WEIGHTS_PACKED 2
#define SR_NET_PACK_ONLY 4   /* weight-norm + packing of `flat` into the blobs, nothing else (backward after the blobs were re-packed) */
int sr_wdsr_net_forward(const sr_wdsr_net_t* net, int flags, sr_stream_t stream);
/* full backward: d(loss)/d(out) -> gflat (gradient of every parameter in the flat buffer). */
int sr_wdsr_net_backward(const sr_wdsr_net_t* net, sr_stream_t stream);
/* The same in two parts, so that a data-parallel trainer can all-reduce the late parameters' gradient while the early
 * half of the backward still runs (pretrain.py:239 DDP overlap): part 1 = tail + blocks [nb_split, NB) (their entries
 * of gflat are final on return), part 2 = blocks [0, nb_split) + head; part 0 = everything. */
int sr_wdsr_net_backward_part(const sr_wdsr_net_t* net, int part, sr_stream_t stream);

/* Standalone nn.PixelShuffle(r) (models/basic_wdsr_b.py:80-83; basicvsr_arch_origin.py:37,87-88), NCHW fp32, bit-exact:
 * out[n, c, h r + i, w r + j] = in[n, c r^2 + i r + j, h, w].  C = channels of the shuffled tensor, H x W = size before
 * the shuffle.  inverse != 0 runs the inverse permutation (pixel_unshuffle = the op's backward): in is then the
 * N x C x rH x rW tensor and out the N x C r^2 x H x W one. */
int sr_pixel_shuffle(const float* in, float* out, int N, int C, int H, int W, int r, int inverse, sr_stream_t stream);

/* Tail backward with the loss folded in (see sr_wdsr_net_t.hr): sr = network output, hr = target. */
int sr_tail_bwd_loss(const float* sr, const float* hr, int loss_kind, float gscale, float* loss_part, const void* feat,
                     const float* x_nchw, float mean, const void* wblob, void* dfeat, float* partial, int wgs, int N, int H,
                     int W, int F, int R, int dtype, sr_stream_t stream);

/* One Adam step over a flat fp32 parameter buffer with the arithmetic of torch.optim.Adam's default (foreach)
 * implementation (pretrain.py:137: lr 1e-3 x world, betas (0.9, 0.999), eps 1e-8, no weight decay).  The caller
 * computes the step-dependent scalars in double and rounds them to float, as torch does:
 *   w_lerp = 1 - beta1, one_minus_beta2 = 1 - beta2, bc2_sqrt = sqrt(1 - beta2^t), neg_step_size = -lr / (1 - beta1^t).
 * loss_out (may be NULL) = loss_scale * sum(loss_part[0 .. n_loss)). */
typedef struct { float w_lerp, beta2, one_minus_beta2, bc2_sqrt, eps, neg_step_size; } sr_adam_t;
int sr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, const sr_adam_t* a,
                 const float* loss_part, int n_loss, float loss_scale, float* loss_out, sr_stream_t stream);
int sr_loss_value(const float* loss_part, int n_loss, float loss_scale, float* loss_out, sr_stream_t stream);
/* y[i] = x[i] * (*scale), the product in fp32; x, y: n elements of `dtype`, 16-byte aligned; scale: a device scalar.  What
   `loss_weight * criterion(sr, hr)` (search.py:74, pretrain.py:73) makes of a folded loss's data gradient: autograd hands the
   weight over as a device tensor, torch's own multiply would round it to the gradient's dtype first. */
int sr_scale_by(void* y, const void* x, long n, const float* scale, int dtype, sr_stream_t stream);
/* forward + loss-folded backward + Adam in one call (net->hr etc. set; net->flat is updated in place). */
int sr_wdsr_net_train_step(const sr_wdsr_net_t* net, float* exp_avg, float* exp_avg_sq, long n_params, const sr_adam_t* a,
                           float loss_scale, float* loss_out, sr_stream_t stream);

/* ---- SPyNet's 7x7 convolutions (models/spynet_arch.py:17-22), bf16 activations NHWC ----
 * y[n, h, w, co] = act(bias[co] + sum_{ky, kx, ci} w[co, ci, ky, kx] x[n, h + ky - 3, w + kx - 3, ci]), zero padding.
 * x: [N][H][W][CIN] bf16; wpacked: the layer's weights as MFMA fragments [7 ky][k-step][32-row tile][64 lanes][8] bf16 (k-step =
 * (kx, 16 input channels), for CIN = 8: two adjacent taps; packing.conv7_pack builds it); bias: [ceil(COUT / 32) * 32] fp32;
 * y: [N][H][W][COUT] bf16, or fp32 when out_f32.  Supported (CIN, COUT, relu, out_f32): (8,32,1,0) (32,64,1,0) (64,32,1,0)
 * (32,16,1,0) (16,2,0,1) -- the five layers of a BasicModule; anything else returns -1. */
int sr_conv7_fwd(const void* x, const void* wpacked, const float* bias, void* y, int N, int H, int W, int CIN, int COUT, int relu,
                 int out_f32, sr_stream_t stream);

/* ---- hardware probes used by tests/test_gpu_probe.py (lane maps the kernels rely on) ---- */
int sr_probe_mfma_bf16(const void* a_frag, const void* b_frag, float* acc_out, sr_stream_t stream);
int sr_probe_mfma_f32(const float* a_frag, const float* b_frag, float* acc_out, sr_stream_t stream);
int sr_probe_tr_read(const void* img_bf16, int n_elems, const int* lane_elem_off, void* out_bf16x4,
                     sr_stream_t stream);
/* streaming copy of n_bytes (multiple of 16): the achievable-HBM-bandwidth yardstick for bench.py */
int sr_probe_copy(const void* src, void* dst, size_t n_bytes, sr_stream_t stream);
/* Launch-floor probe: `reps` dependent launches of an empty kernel with the given grid, workgroup size and
 * dynamic LDS; writes gx*gy words to `out`.  Timing it gives the fixed cost under every chained kernel. */
int sr_probe_launch_floor(void* out, int gx, int gy, int threads, int lds_bytes, int reps, sr_stream_t stream);
/* The same chain captured in a hipGraph and replayed `iters` times (synchronous; result in *us_per_launch). */
int sr_probe_launch_floor_graph(void* out, int gx, int gy, int threads, int lds_bytes, int reps, int iters,
                                float* us_per_launch, float* host_us_per_graph /* may be NULL */);
/* Debug: in the diagnostic build (libsr_hotpath_dbg.so, `python -m mobilesuperresolution_amd.build --debug`) the
 * instrumented kernels write s_memrealtime stamps to buf[n_workgroups][16 waves][16 stamps][2] u64 (NULL = off); workgroups
 * beyond n_workgroups do not stamp.  The product library contains no stamp code and returns -1. */
int sr_debug_set_stamps(void* buf, long n_workgroups);

#ifdef __cplusplus
}
#endif
#endif /* SR_HOTPATH_H */
