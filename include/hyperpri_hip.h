/* hyperpri_hip.h -- C ABI of libhyperpri_hip.so (MI355X / gfx950 only).
 *
 * The reference (GatorSense/HyperPRI) has no FFI: its hot path is plain torch.nn modules
 * (src/Experiments/model_parts.py, src/Experiments/models.py) whose arithmetic lives in ATen/cuDNN.
 * This library is what a maintainer would bind in their place: one launcher per kernel family and
 * direction, plain pointers and sizes, no torch types.  hyperpri_amd/_lib.py holds the ctypes binding;
 * INTEGRATION.md shows the reference-side shim.
 *
 * Conventions
 *   - Activations are fp32 NHWC.  A tensor view is (pointer, cs, coff): element (pixel p, channel c)
 *     lives at ptr[p*cs + coff + c]; cs and coff are multiples of 4 (16-byte channel vectors) and the
 *     pointer is 16-byte aligned.  Views into a wider buffer are how skip-concat (model_parts.py:87)
 *     costs no copy on the consumer side.
 *   - Every buffer, including workspaces, is owned by the caller (PyTorch caching allocator); the
 *     library never allocates, frees or synchronises.  All launches go to the given hipStream_t.
 *   - Return value: 0 = ok, <0 = error (HPRI_ERR_*); hpri_last_error() returns the thread's message.
 *     Nothing throws across the ABI.
 *   - Launchers are re-entrant (autograd calls them from worker threads).  The only process-wide state is immutable after
 *     its first use or atomic: the launch-plan options (read once from the environment under std::call_once, then
 *     atomics; hpri_set_option) and a per-device cache of the compute-unit count; the error message is thread-local.  Stamp
 *     buffers exist in the diagnostic builds (-DHPRI_STAMPS) only.
 */
#ifndef HYPERPRI_HIP_H
#define HYPERPRI_HIP_H

#include <stddef.h>
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPRI_OK 0
#define HPRI_ERR_ARG (-1)
#define HPRI_ERR_UNSUPPORTED (-2)
#define HPRI_ERR_WORKSPACE (-3)
#define HPRI_ERR_LAUNCH (-4)

#define HPRI_A_DIRECT 0 /* A operand read in place                                   */
#define HPRI_A_S2D 1    /* A operand gathered as 2x2 stride-2 patches (convT grads)  */
#define HPRI_E_DIRECT 0 /* NHWC store                                                */
#define HPRI_E_D2S 1    /* 2x2 stride-2 pixel-shuffle store (convT forward)          */

int hpri_version(void);
/* Launch-plan options (process-wide): "conv_nbx_min", "wgrad_xcd_min_tiles", "wgrad_xcd_min_strips", "bf16v3_tile_width", "bn_wide_cq" -- the problem sizes
 * from which the conv / weight-gradient kernels switch to their XCD-aware 1-D grids -- and "wgrad_cu_reserve": compute
 * units the fp32 Winograd weight gradient (one workgroup per CU, grids planned as exact multiples of the CU count) leaves to other
 * kernels, e.g. the channels of a collective that runs beside the backward (0 = none).  Results do not depend
 * on them, only block order and (for weight gradients) the number of partial slabs, i.e. the summation order. */
int hpri_set_option(const char* name, int value);
/* A non-blocking stream of the lowest priority the current device offers (*priority receives it); the caller owns it. */


int hpri_get_option(const char* name);
const char* hpri_last_error(void);

/* ---- item queues of the persistent MFMA kernels (hpri_conv_bf16v3*, hpri_gemm_bf16v3 / hpri_convt_*_bf16*, hpri_gemm_f32v2 /
 * hpri_convt_*_f32v2) ----------------------------------------------------------------------------------------------------------
 * These launches start 2 x CUs workgroups, two per CU.  Without a queue every workgroup walks a fixed list of work items; a
 * workgroup that cannot become resident at once (another kernel -- an RCCL collective beside the backward of a DDP step,
 * PLTrainer.py:434-442 -- holds part of its CU) then runs its whole list late: one such workgroup costs a bf16 step 6-15 %.
 * hpri_set_item_queue(queue, bytes, stream) gives every later persistent launch ON THAT STREAM `queue` -- hpri_item_queue_bytes()
 * bytes of device memory, 256-byte aligned, zeroed once by the caller, owned by the caller and alive while it may launch -- as its
 * item counters: workgroups draw their items (same items, same per-XCD order, results bit-identical) and a late workgroup finds
 * nothing left.  The buffer holds two halves; launches on the stream alternate between them and each launch zeroes the half the
 * next one will use (the library keeps the parity per registered stream).  One queue per stream (launches of one stream run one
 * after the other; the null stream of one device per process); queue == NULL unregisters the stream. */
int hpri_item_queue_bytes(void);
/* ---- the half-precision build (libhyperpri_hip_f16.so: the same entry points with IEEE half as the 16-bit type of every "bf16" plane,
 * row and packed-weight argument; precision mode "f16" of hyperpri_amd) needs a loss scale: hpri_set_loss_scale(s) makes the fused
 * heads of the CALLING THREAD (hpri_outconv_bwd_bce, hpri_outconv_bwd_x16 with a target) form s times the loss gradient -- s a power
 * of two chosen by the caller from the number of logits, so that activation gradients sit in half's range -- and
 * hpri_scale_tensors(tensors, numel, n, 1 / s) takes it out of the parameter gradients afterwards.  In libhyperpri_hip.so (bf16:
 * eight exponent bits) both exist and the scale stays 1. */
int hpri_set_loss_scale(float scale);
int hpri_scale_tensors(float* const* tensors, const long long* numel, int ntensors, float scale, hipStream_t stream);
int hpri_set_item_queue(void* queue, size_t bytes, hipStream_t stream);

/* ---- weight packing: nn.Parameter layouts -> [chunk][tap][32][Ncols_pad] LDS panels ---------------
 * mode 0 conv fwd   (replaces cuDNN's filter transform for nn.Conv2d, model_parts.py:22,25,96; nn.Conv3d
 *                    models.py:169; nn.Linear models.py:108,103)
 * mode 1 conv dgrad (autograd of the same), mode 2 convT fwd (model_parts.py:63; models.py:198),
 * mode 3 convT dgrad.  K = reduction length per tap, Ncols = GEMM columns, src_d1 = dim 1 of the
 * source tensor. */
size_t hpri_packed_weight_floats(int K, int Ncols_pad, int T);
int hpri_pack_weight(const float* w, float* wp, int mode, int K, int Ncols, int Ncols_pad, int T, int Cup,
                     int src_d0, int src_d1, hipStream_t stream);

/* eval-mode conv->BN->ReLU folded into one conv (predict / validate / test paths, PLTrainer.py:142-162, 530-532):
 * hpri_bn_fold gives scale = gamma/sqrt(var+eps) and the folded bias; hpri_pack_weight_scaled packs w*scale; bit 1 of
 * hpri_conv_fwd's `accumulate` argument (value 2) turns on the ReLU epilogue. */
int hpri_bn_fold(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                 const float* conv_bias, float eps, int C, float* scale, float* fbias, hipStream_t stream);
int hpri_pack_weight_scaled(const float* w, float* wp, const float* colscale, int K, int Ncols, int Ncols_pad, int T,
                            int src_d1, hipStream_t stream);
/* the same fold for the bf16 / bf16x3 / bf16x6 modes (scale applied in fp32, then rounded / split into planes) */
int hpri_pack_weight_bf16_scaled(const float* w, void* wp, const float* colscale, int K, int Ncols, int Ncols_pad, int T,
                                 int src_d1, int split, hipStream_t stream);

/* 64-bit content fingerprint of n_words 32-bit words (order-independent integer sum; `out` is zeroed by the call, on the
 * stream).  Used by the verify mode of the packed-weight cache (hyperpri_amd/engine.py: HPRI_PACK_VERIFY=1), which catches
 * parameters rewritten through `p.data` -- writes the nn.Parameter version counter (the cache key that stands in for
 * model_parts.py:22's weight tensor) does not see. */
int hpri_fingerprint(const void* w, long long n_words, unsigned long long* out, hipStream_t stream);

/* ---- implicit-GEMM convolution, fp32 MFMA (conv_fwd.hip) ----------------------------------------
 * Replaces F.conv2d / F.conv3d / F.linear / F.conv_transpose2d forward and their data gradients
 * (model_parts.py:22,25,63,96; models.py:108,169,177,198).  hpri_conv_fwd_plan (host only) returns the split-K
 * factor chosen for the shape, the workspace it needs, and the number of BatchNorm partial records: stats
 * (optional) receives stat_tiles * Cout_pad float4 (mean, M2, count, 0), image-major. */
int hpri_conv_fwd_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int KS, int amode, int epi, int* ksplit,
                       int* stat_tiles, size_t* ws_floats);
int hpri_conv_fwd(const float* x, int x_cs, int x_coff, const float* wp, const float* bias, float* y, int y_cs,
                  int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw,
                  int KS, int amode, int epi, int accumulate, int H2, int W2, int py0, int px0, int Cup, float* ws,
                  size_t ws_floats, hipStream_t stream);

/* ---- Winograd F(2x2,3x3), exact fp32 MFMA (conv_wino4.hip) --------------------------------------------------------
 * 3x3 / pad 1 / stride 1 convolutions (model_parts.py:22,25; models.py:169,177), forward and data gradient, with 16 instead
 * of 36 multiplies per 2x2 outputs: 4-wave workgroups of 16 x 8 pixels, two per CU, wave = frequency row.  hpri_wino4_pack
 * transforms the filters (U = G g G^T; mode 0 forward, mode 1 data gradient, optional per-column scale for the eval-mode BN
 * fold) into [K/8][16][Ncols_pad][8] (hpri_wino_packed_floats floats); hpri_conv_wino4_plan gives the number of BatchNorm partial
 * records (one per 16 x 8-pixel tile); `accumulate` bit 0: y += result, bit 1: ReLU.  x: fp32 NHWC view with channels
 * [Cin, Cin_pad) zero (Cin_pad a multiple of 8).  (The first form of this kernel, conv_wino.hip, lives in the diagnostics build:
 * hyperpri_hip_diag.h.) */
size_t hpri_wino_packed_floats(int K, int Ncols_pad);
int hpri_wino4_pack(const float* w, float* up, const float* colscale, int mode, int K, int Ncols, int Ncols_pad, int src_d1,
                    hipStream_t stream);
int hpri_conv_wino4_plan(int N, int H, int W, int* stat_tiles);
int hpri_conv_wino4(const float* x, int x_cs, int x_coff, const float* up, const float* bias, float* y, int y_cs, int y_coff,
                    float* stats, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw, int accumulate,
                    hipStream_t stream);

/* hpri_conv_wino4 as a DATA GRADIENT (mode-1 pack, no bias, no accumulate) that also leaves the BatchNorm-backward partial sums of the
 * conv -> BN -> ReLU stage whose output gradient y it writes (model_parts.py:22-27's autograd): bn_x = that stage's pre-BN tensor
 * (fp32 NHWC view, same pixels / channels as y, Cout_pad channels wide), its per-channel mean / invstd / scale / shift, bn_relu;
 * bn_part[N * tiles][2][bn_cpart] (tiles: hpri_conv_wino4_plan) receives sum g*[y>0] and sum g*[y>0]*xhat per tile.  Finish with
 * hpri_bn_relu_bwd_fused, which then skips its two reduction sweeps over g and the pre-BN tensor. */
int hpri_conv_wino4_bnred(const float* x, int x_cs, int x_coff, const float* up, float* y, int y_cs, int y_coff, int N, int H,
                          int W, int Cin_pad, int Cout, int Cout_pad, int y_cw, const float* bn_x, int bn_x_cs, int bn_x_coff,
                          const float* bn_mean, const float* bn_invstd, const float* bn_scale, const float* bn_shift,
                          int bn_relu, float* bn_part, int bn_cpart, hipStream_t stream);
int hpri_bn_relu_bwd_fused(const float* partials, int part_blocks, int part_cpart, const float* dy, int dy_cs, int dy_coff,
                           const float* x, int x_cs, int x_coff, float* dx, int dx_cs, int dx_coff, const float* mean,
                           const float* invstd, const float* scale, const float* shift, float* dgamma, float* dbeta,
                           int accumulate_param_grads, float* dbias, int accumulate_dbias, float* workspace, size_t ws_floats,
                           long long P, long long pix_per_group, int C, int Cw, int relu, int use_batch_stats, void* planes,
                           long long plane_stride, int pl_cs, int pl_coff, int pl_cw, int npl, hipStream_t stream);


/* Winograd weight gradient (dU = sum over tiles of V * (A dY A^T), dg = G^T dU G): slabs ws[split][16][Cr][Nr] from
 * hpri_conv_wino_wgrad (sizes: hpri_wino_wgrad_plan), fixed-order sum + inverse filter transform into OIHW by
 * hpri_wino_wgrad_reduce. */
int hpri_wino_wgrad_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int* splits, int* Cr, int* Nr);
int hpri_conv_wino_wgrad(const float* x, int x_cs, int x_coff, int x_cvalid, const float* dy, int dy_cs, int dy_coff,
                         int dy_cvalid, float* ws, size_t ws_floats, int N, int H, int W, int Cin_pad, int Cout_pad,
                         hipStream_t stream);
int hpri_wino_wgrad_reduce(const float* ws, float* dw, int N, int H, int W, int Cin, int Cin_pad, int Cout, int Cout_pad,
                           int accumulate, hipStream_t stream);

/* bf16-operand variants (precision mode "bf16", BASELINE.json config C5): operands rounded to bf16 while staged into
 * LDS, v_mfma_f32_32x32x16_bf16 with fp32 accumulate, fp32 activations in HBM.  Same modes, plan, workspace and
 * statistics contract as hpri_conv_fwd / hpri_pack_weight (plan: hpri_conv_fwd_bf16_plan).  split = 1 (precision mode
 * "bf16x3"): operands carried as bf16 hi + bf16 lo (16 mantissa bits), three MFMAs per product (hi*hi + hi*lo + lo*hi);
 * split = 2 ("bf16x6"): hi + mid + lo = the fp32 operand exactly, six MFMAs per product (fp32-class accuracy); the
 * packed weights hold split + 1 planes. */
int hpri_conv_fwd_bf16_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int KS, int amode, int epi, int split,
                            int* ksplit, int* stat_tiles, size_t* ws_floats);
int hpri_pack_weight_bf16(const float* w, void* wp, int mode, int K, int Ncols, int Ncols_pad, int T, int src_d1,
                          int Cup, int split, hipStream_t stream);
/* The same for a layer whose INPUT-channel axis carries gap_len structural-zero channels from gap_at on (the padded concat of the
 * bf16 plane mode: [a | zeros up to a multiple of 32 | b]); modes 0 / 1; K (mode 0) resp. Ncols (mode 1) is the padded width, src_d1
 * the weight's own. */
int hpri_pack_weight_bf16_gap(const float* w, void* wp, int mode, int K, int Ncols, int Ncols_pad, int T, int src_d1, int gap_at,
                              int gap_len, hipStream_t stream);
int hpri_conv_fwd_bf16(const float* x, int x_cs, int x_coff, const void* wp, const float* bias, float* y, int y_cs,
                       int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw,
                       int KS, int amode, int epi, int accumulate, int H2, int W2, int py0, int px0, int Cup, int split,
                       float* ws, size_t ws_floats, hipStream_t stream);
/* ConvTranspose2d(k2,s2) forward (model_parts.py:63-64) in the plain bf16 mode, result as ONE bf16 plane: channels [pl_coff,
 * pl_coff + Cup) of a plane buffer with pl_cs elements per hi-res pixel (the decoder's concat planes); also fp32 when y != NULL. */
int hpri_convt_fwd_bf16_pl(const float* x, int x_cs, int x_coff, const void* wp, const float* bias, float* y, int y_cs,
                           int y_coff, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int H2, int W2, int py0, int px0,
                           int Cup, void* planes, int pl_cs, int pl_coff, hipStream_t stream);

/* bf16 activation PLANES: in the bf16 modes the producer of an activation writes it as bf16 NHWC
 * planes (plane 0 = bf16(x), plane 1 = bf16(x - hi), ...; `plane_stride` elements apart, `cs16` elements per pixel,
 * channels [C, cw16) zero), so the 3x3 convolution (model_parts.py:22,25; models.py:169,177; forward, or data gradient
 * with the mode-1 pack) brings BOTH operands into LDS by LDS-DMA.  hpri_to_planes is the generic producer (fp32 NHWC
 * view -> planes); plan / workspace / statistics contract as hpri_conv_fwd (split-K finish: hpri_splitk_finish). */
int hpri_to_planes(const float* x, int cs, int coff, void* planes, long long plane_stride, int cs16, int coff16,
                   long long P, int C, int cw16, int npl, hipStream_t stream);


int hpri_splitk_finish(const float* ws, int ksplit, int Cout_pad, const float* bias, float* y, int y_cs, int y_coff,
                       float* stats, int N, int HW, int Cout, int y_cw, int accumulate, int relu, hipStream_t stream);
/* First-layer fused ingest (conv_ingest.hip; predict path of the 16-bit modes).  Replaces, for nn.Conv3d(1, F, (D,3,3), padding=(0,1,1)) on
 * the caller's (N,1,D,H,W) cube -- reference models.py:169 (first_conv) as called at models.py:215-216 -- the pair layout pass
 * (hpri_nchw_to_nhwc_pl) + hpri_conv_bf16v3: the 3x3 pad-1 convolution reads the contiguous fp32 NC(D)HW tensor `x` itself (C = D
 * channels) and rounds it to the library's 16-bit type while staging.  `wp`: weights packed by hpri_pack_weight_bf16(_scaled)
 * ([chunk][tap][Cout_pad][32], channels beyond C zero; eval-mode BatchNorm folded in by the caller: hpri_bn_fold), `bias` (may be
 * null), `relu`; result as 16-bit rows y + pixel * y_cs + y_coff + channel (Cout a multiple of 4, y 8-byte aligned, y_cs / y_coff
 * multiples of 4) = the planes the next convolution stages.  One image of x must stay below 2 GiB.  Same accumulation order as
 * hpri_conv_bf16v3: bit-identical to the pair it replaces. */
int hpri_conv3x3_ingest_h16(const float* x, const void* wp, const float* bias, void* y, int y_cs, int y_coff, int N, int C, int H,
                            int W, int Cout, int Cout_pad, int relu, hipStream_t stream);
/* Third form of the plane convolution (conv_bf16v3.hip): 4-wave workgroups of 256 pixels x 64 channels, TWO per CU (one's
 * prologue / store + statistics epilogue runs under the other's MFMAs), v_mfma_f32_16x16x32_bf16 with the weights as the A
 * operand (a lane's accumulator registers are consecutive channels of one pixel: 16-byte stores without an LDS transpose).
 * x_plane is unused (one plane); `split` must be 0; `accumulate` bit 0: y += result, bit 1: ReLU; plan / workspace / statistics
 * contract as hpri_conv_fwd (records per 256-pixel tile: hpri_conv_bf16v3_plan); the output view must be float4-aligned.  Bit 2 of `accumulate` (value 4): the output view is bf16 (y points at bf16 elements,
 * y_cs / y_coff in elements; not with bit 0, not for split-K problems) -- the pre-BN tensor at 2 bytes per element, read by
 * hpri_bn_apply_relu_x16 / hpri_bn_relu_bwd_x16.  (Its predecessor conv_bf16v2.hip, the _dbg entry with a caller-given stagger and stamp
 * buffer, and the variant with BatchNorm-backward sums in the epilogue live in the diagnostics build: hyperpri_hip_diag.h.) */
int hpri_conv_bf16v3_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int* ksplit, int* stat_tiles,
                          size_t* ws_floats);
int hpri_conv_bf16v3(const void* xp, long long x_plane, int x_cs, int x_coff, const void* wp, const float* bias, float* y,
                     int y_cs, int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw,
                     int accumulate, int split, float* ws, size_t ws_floats, hipStream_t stream);


/* hpri_conv_bf16v3 (no accumulate; hpri_conv_bf16v3_plan must report ksplit 1) whose result channels [y2_c0, y2_c0 + y2_cw) -- whole
 * 64-channel blocks -- are also (y2_only bit 0: only) written as bf16 rows, y2 + pixel * y2_cs + y2_coff + (channel - y2_c0): the
 * gradient of the upsampled half of a decoder concat for the plane-fed transposed-convolution kernels.  y2_only bit 1 (round 4): the
 * main output y holds bf16 rows as well (y_cs / y_coff in elements) -- the gradient of a planes-only skip tensor. */
int hpri_conv_bf16v3_y2(const void* xp, int x_cs, int x_coff, const void* wp, const float* bias, float* y, int y_cs, int y_coff,
                        float* stats, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw, void* y2, int y2_cs,
                        int y2_coff, int y2_c0, int y2_cw, int y2_only, hipStream_t stream);


/* ---- fp32 GEMM for the 1x1 forms of the exact-fp32 mode, second form (gemm_f32v2.hip, round 4): ConvTranspose2d(k=2,s=2) forward and
 * data gradient (model_parts.py:63-64; models.py:198) and the plain row GEMM, on v_mfma_f32_32x32x2_f32 with both operands by LDS-DMA,
 * two persistent 4-wave workgroups of 256 pixels x 128 columns per CU.  hpri_pack_weight_f32k16 packs [chunk of 16 k][Ncols_pad][16]
 * (modes as hpri_pack_weight: 0 Linear forward W[n][k], 1 its data gradient, 2 ConvTranspose2d forward (column = tap*Cup + co), 3 its
 * data gradient (k = tap*Cup + co); src_d1 = dim 1 of the source tensor for modes 0 / 1).  x rows: x_cs floats apart (multiple of 4),
 * K_pad (multiple of 16) of them read from x_coff on; pad channels must hold zeros or meet zero weights.  `accumulate` bit 0: y +=
 * result (not for the depth-to-space form).  No statistics epilogue: layers in front of a BatchNorm keep hpri_conv_fwd. */
size_t hpri_packed_weight_f32k16_floats(int K, int Ncols_pad);
int hpri_pack_weight_f32k16(const float* w, float* wp, int mode, int K, int Ncols, int Ncols_pad, int Cup, int src_d1,
                            hipStream_t stream);
int hpri_gemm_f32v2(const float* x, int x_cs, int x_coff, const float* wp, const float* bias, float* y, int y_cs, int y_coff,
                    int N, long long HW, int K_pad, int Ncols, int Ncols_pad, int y_cw, int accumulate, hipStream_t stream);
int hpri_convt_fwd_f32v2(const float* x, int x_cs, int x_coff, const float* wp, const float* bias, float* y, int y_cs,
                         int y_coff, int N, int H, int W, int K_pad, int Cup, int Ncols_pad, int H2, int W2, int py0, int px0,
                         hipStream_t stream);
int hpri_convt_dgrad_f32v2(const float* dy, int dy_cs, int dy_coff, const float* wp, float* dx, int dx_cs, int dx_coff, int N,
                           int H, int W, int Cup, int Cin, int Cin_pad, int dx_cw, int H2, int W2, int py0, int px0,
                           int accumulate, hipStream_t stream);

/* ---- plane-fed GEMM for the 1x1 forms of the bf16 mode (gemm_bf16v3.hip): nn.Linear / Conv2d(k=1) forward and data gradient
 * (models.py:105-115,143; model_parts.py:96) and ConvTranspose2d(k=2,s=2) forward / data gradient (model_parts.py:63-64).
 * hpri_gemm_bf16v3: y[p, n] (+)= sum_k x[p, k] w[n, k] + bias[n]; x = bf16 planes of N*HW rows (x_cs elements per row, K_pad read
 * from x_coff on), w = hpri_pack_weight_bf16 with T = 1 (modes 0 / 1); outputs fp32 view y and / or bf16 view y16 (either may be
 * NULL), y_cw columns written; accumulate bit 0: add to y, bit 1: ReLU; stats: one record row of stat_cp columns per 256-row tile
 * (hpri_gemm_bf16v3_plan; tiles never straddle images) for hpri_bn_finalize.
 * hpri_convt_fwd_bf16v3: column tap*Cup + co (pack mode 2) -> pixel (py0 + 2y + tap/2, px0 + 2x + tap%2), channel co of the
 * [N, H2, W2] views; Cup % 16 == 0.  hpri_convt_dgrad_bf16v3: dx (+)= gather over the four parities of the dy planes (pack mode 3,
 * K = 4*Cup); Cup % 32 == 0. */
int hpri_gemm_bf16v3_plan(int N, long long HW, int* stat_tiles);
int hpri_gemm_bf16v3(const void* xp, int x_cs, int x_coff, const void* wp, const float* bias, float* y, int y_cs, int y_coff,
                     void* y16, int y16_cs, int y16_coff, float* stats, int stat_cp, int N, long long HW, int K_pad, int Ncols,
                     int Ncols_pad, int y_cw, int accumulate, hipStream_t stream);
int hpri_convt_fwd_bf16v3(const void* xp, int x_cs, int x_coff, const void* wp, const float* bias, float* y, int y_cs, int y_coff,
                          void* y16, int y16_cs, int y16_coff, int N, int H, int W, int K_pad, int Cup, int Ncols_pad, int H2,
                          int W2, int py0, int px0, hipStream_t stream);
int hpri_convt_dgrad_bf16v3(const void* dyp, int dy_cs, int dy_coff, const void* wp, float* dx, int dx_cs, int dx_coff, int N,
                            int H, int W, int Cup, int Cin, int Cin_pad, int dx_cw, int H2, int W2, int py0, int px0,
                            int accumulate, hipStream_t stream);
/* ... the same written as bf16 rows (round 4; no accumulate): a decoder stage's input has one reader of its gradient. */
int hpri_convt_dgrad_bf16v3_y16(const void* dyp, int dy_cs, int dy_coff, const void* wp, void* dx16, int dx_cs, int dx_coff, int N,
                                int H, int W, int Cup, int Cin, int Cin_pad, int dx_cw, int H2, int W2, int py0, int px0,
                                hipStream_t stream);

/* ---- weight gradients, fp32 MFMA, deterministic split-K (conv_wgrad.hip) --------------------------
 * Replaces the wgrad half of autograd for the same layers.  dst_mode 0 writes OIHW / (out,in),
 * dst_mode 1 writes ConvTranspose2d's (Cin,Cout,2,2).  Workspace: splits*KS*KS*Cr*Nr floats. */
int hpri_wgrad_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int KS, int* splits, int* Cr, int* Nr);
int hpri_conv_wgrad(const float* x, int x_cs, int x_coff, int x_cvalid, const float* dy, int dy_cs, int dy_coff,
                    int dy_cvalid, float* ws, size_t ws_floats, int N, int H, int W, int Cin_pad, int Cout_pad, int KS,
                    int bmode, int H2, int W2, int py0, int px0, int Cup, hipStream_t stream);
int hpri_conv_wgrad_bf16(const float* x, int x_cs, int x_coff, int x_cvalid, const float* dy, int dy_cs, int dy_coff,
                         int dy_cvalid, float* ws, size_t ws_floats, int N, int H, int W, int Cin_pad, int Cout_pad,
                         int KS, int bmode, int H2, int W2, int py0, int px0, int Cup, int split, hipStream_t stream);
int hpri_wgrad_reduce(const float* ws, float* dw, int N, int H, int W, int Cin, int Cin_pad, int Cout, int Cout_pad,
                      int KS, int dst_mode, int Cup, int accumulate, hipStream_t stream);
/* The same fixed-order reduction for slabs whose pixel splits the caller planned itself (hpri_wgrad_bf16v2_plan). */
int hpri_wgrad_reduce_ex(const float* ws, float* dw, int splits, int Cr, int Nr, int Cin, int Cout, int KS, int dst_mode,
                         int Cup, int accumulate, hipStream_t stream);
/* Weight gradient of the 3x3 / pad 1 convolutions from bf16 PLANES (conv_wgrad_bf16v2.hip; precision mode "bf16"; the
 * autograd of model_parts.py:22,25 and models.py:169,177): plane 0 of the conv input and of the output gradient (NHWC,
 * strides / offsets / valid widths multiples of 8 elements), both by LDS-DMA; slabs ws[splits][9][Nr][Cr] (sizes from
 * the plan), finished by hpri_wgrad_reduce_ex(ws, dw, splits, Cr, Nr, Cin, Cout, 3, 0, 0, accumulate). */
int hpri_wgrad_bf16v2_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int* splits, int* Cr, int* Nr);
int hpri_conv_wgrad_bf16v2(const void* x_planes, int x_cs, int x_coff, int x_cvalid, const void* dy_planes, int dy_cs,
                           int dy_coff, int dy_cvalid, float* ws, size_t ws_floats, int N, int H, int W, int Cin_pad,
                           int Cout_pad, hipStream_t stream);

/* Weight gradient of the 1x1 layers (nn.Linear / Conv2d(k=1): models.py:105-115,143) from bf16 PLANES (wgrad_bf16v3.hip): plane 0
 * of the layer input and of the output gradient, P pixel rows each (strides / offsets / valid widths multiples of 8 elements), both
 * by LDS-DMA; slabs ws[splits][Nr][Cr] (sizes from the plan), finished by hpri_wgrad_reduce_ex(ws, dw, splits, Cr, Nr, Cin, Cout, 1,
 * 0, 0, accumulate). */
int hpri_wgrad1x1_bf16v3_plan(long long P, int Cin_pad, int Cout_pad, int* splits, int* Cr, int* Nr);
int hpri_wgrad1x1_bf16v3(const void* x_planes, int x_cs, int x_coff, int x_cvalid, const void* dy_planes, int dy_cs, int dy_coff,
                         int dy_cvalid, float* ws, size_t ws_floats, long long P, int Cin_pad, int Cout_pad, hipStream_t stream);

/* ConvTranspose2d(k=2,s=2) weight gradient (model_parts.py:63-64) from bf16 planes: x [N,H,W] (Cin channels) and the gradient of the
 * upsampled tensor dy [N,H2,W2] (Cup channels from dy_coff on, Cup % 64 == 0), gathered by parity; slab rows n = tap*Cup + co; plan:
 * hpri_wgrad1x1_bf16v3_plan(N*H*W, Cin_pad, 4*Cup); finish: hpri_wgrad_reduce_ex(ws, dw, splits, Cr, Nr, Cin, 4*Cup, 1, 1, Cup, acc). */
int hpri_wgrad_convt_bf16v3(const void* x_planes, int x_cs, int x_coff, int x_cvalid, const void* dy_planes, int dy_cs, int dy_coff,
                            float* ws, size_t ws_floats, int N, int H, int W, int Cin_pad, int Cup, int H2, int W2, int py0, int px0,
                            hipStream_t stream);

/* ---- BatchNorm (+ReLU) (bn.hip): nn.BatchNorm2d/3d/1d + nn.ReLU, model_parts.py:23-27; models.py:113-114,
 * 172-173,178-179.  G groups = independent statistic sets (G = N for SpectralUNET's per-image loop,
 * models.py:132). */
int hpri_bn_finalize(const float* partials, int tiles_per_group, int G, int Cp, int C, const float* gamma,
                     const float* beta, float eps, float momentum, float* mean, float* invstd, float* var_unbiased,
                     float* scale, float* shift, float* running_mean, float* running_var,
                     long long* num_batches_tracked, hipStream_t stream);
int hpri_bn_eval_prepare(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                         float eps, int C, float* mean, float* invstd, float* scale, float* shift, hipStream_t stream);
int hpri_bn_apply_relu(const float* x, int x_cs, int x_coff, float* y, int y_cs, int y_coff, const float* scale,
                       const float* shift, long long P, long long pix_per_group, int C, int Cw, int relu,
                       hipStream_t stream);
int hpri_col_reduce_plan(long long pix_per_group, int G, int C, int* nblk, int* Cpart);
int hpri_bn_relu_bwd(const float* dy, int dy_cs, int dy_coff, const float* x, int x_cs, int x_coff, float* dx,
                     int dx_cs, int dx_coff, const float* mean, const float* invstd, const float* scale,
                     const float* shift, float* dgamma, float* dbeta, int accumulate_param_grads, float* dbias,
                     int accumulate_dbias, float* workspace, size_t ws_floats, long long P, long long pix_per_group,
                     int C, int Cw, int relu, int use_batch_stats, hipStream_t stream);
int hpri_col_sum(const float* src, int cs, int coff, float* out, int accumulate, float* workspace, size_t ws_floats,
                 long long P, int C, hipStream_t stream);
/* the same sum taken from the per-tile (mean, M2, count, 0) records a convolution epilogue leaves (`stats` of hpri_conv_wino4 /
 * hpri_conv_bf16v3 / hpri_conv_fwd): out[c] (+)= sum over tiles of mean * count of channel c0 + c.  The ConvTranspose2d bias gradient
 * (model_parts.py:63-64) without a pass over the gradient tensor: the data-gradient kernel that wrote it recorded its columns. */
int hpri_colsum_from_stats(const float* stats, int tiles, int Cpad, int c0, int C, float* out, int accumulate,
                           hipStream_t stream);
/* the same two passes, also writing their output as bf16 planes (hpri_to_planes layout; channels [C, pl_cw) zero) for
 * the bf16-mode convolutions: the producer writes the planes, so no conversion pass reads the fp32 tensor again */
int hpri_nchw_to_nhwc_pl(const float* src, float* dst, int N, int C, long long P, int cs, int coff, int Cw, void* planes,
                         long long plane_stride, int pl_cs, int pl_coff, int pl_cw, int npl, hipStream_t stream);
int hpri_maxpool2_fwd_pl(const float* x, int x_cs, int x_coff, float* y, int y_cs, int y_coff, int N, int H, int W, int C,
                         void* planes, long long plane_stride, int pl_cs, int pl_coff, int pl_cw, int npl,
                         hipStream_t stream);
int hpri_bn_apply_relu_pl(const float* x, int x_cs, int x_coff, float* y, int y_cs, int y_coff, const float* scale,
                          const float* shift, long long P, long long pix_per_group, int C, int Cw, int relu, void* planes,
                          long long plane_stride, int pl_cs, int pl_coff, int pl_cw, int npl, hipStream_t stream);
/* dx may be NULL when planes are given: the gradient is then written as bf16 planes only (plane-mode consumers read nothing else). */
int hpri_bn_relu_bwd_pl(const float* dy, int dy_cs, int dy_coff, const float* x, int x_cs, int x_coff, float* dx,
                        int dx_cs, int dx_coff, const float* mean, const float* invstd, const float* scale,
                        const float* shift, float* dgamma, float* dbeta, int accumulate_param_grads, float* dbias,
                        int accumulate_dbias, float* workspace, size_t ws_floats, long long P, long long pix_per_group,
                        int C, int Cw, int relu, int use_batch_stats, void* planes, long long plane_stride, int pl_cs,
                        int pl_coff, int pl_cw, int npl, hipStream_t stream);
/* The pre-BatchNorm tensor stored as bf16 (bf16 precision mode, HPRI_YR_BF16: hpri_conv_bf16v3 with bit 2 of `accumulate` writes
 * it; x16[p * x_cs + x_coff + c], 8-byte aligned channel quads): the normalise pass and the backward read 2 instead of 4 bytes
 * per element of model_parts.py:23,26's input; arithmetic and every other argument as the fp32 forms. */
int hpri_bn_apply_relu_x16(const void* x16, int x_cs, int x_coff, float* y, int y_cs, int y_coff, const float* scale,
                           const float* shift, long long P, long long pix_per_group, int C, int Cw, int relu, void* planes,
                           long long plane_stride, int pl_cs, int pl_coff, int pl_cw, int npl, hipStream_t stream);

int hpri_bn_relu_bwd_x16(const float* dy, int dy_cs, int dy_coff, const void* x16, int x_cs, int x_coff, float* dx,
                         int dx_cs, int dx_coff, const float* mean, const float* invstd, const float* scale,
                         const float* shift, float* dgamma, float* dbeta, int accumulate_param_grads, float* dbias,
                         int accumulate_dbias, float* workspace, size_t ws_floats, long long P, long long pix_per_group,
                         int C, int Cw, int relu, int use_batch_stats, void* planes, long long plane_stride, int pl_cs,
                         int pl_coff, int pl_cw, int npl, hipStream_t stream);
/* ... and with the incoming gradient stored as bf16 as well (the inner tensor of a DoubleConv in the bf16 mode: written by its only
 * producer, hpri_conv_bf16v3 with the bf16-output bit, read only here; dy_cs / dy_coff in elements) */
int hpri_bn_relu_bwd_x16_dy16(const void* dy16, int dy_cs, int dy_coff, const void* x16, int x_cs, int x_coff, float* dx,
                              int dx_cs, int dx_coff, const float* mean, const float* invstd, const float* scale,
                              const float* shift, float* dgamma, float* dbeta, int accumulate_param_grads, float* dbias,
                              int accumulate_dbias, float* workspace, size_t ws_floats, long long P, long long pix_per_group,
                              int C, int Cw, int relu, int use_batch_stats, void* planes, long long plane_stride, int pl_cs,
                              int pl_coff, int pl_cw, int npl, hipStream_t stream);

/* ---- bandwidth-bound ops (elementwise.hip) ---------------------------------------------------------
 * layout change at the module boundary (dataset.py:267-271 hands NC(D)HW), nn.MaxPool2d(2)
 * (model_parts.py:40), F.pad + torch.cat (model_parts.py:77-87; models.py:235-239), OutConv / final Linear
 * (model_parts.py:96; models.py:103,143), synthetic generator. */
int hpri_nchw_to_nhwc(const float* src, float* dst, int N, int C, long long P, int cs, int coff, int Cw,
                      hipStream_t stream);
int hpri_nhwc_to_nchw(const float* src, float* dst, int N, int C, long long P, int cs, int coff, int accumulate,
                      hipStream_t stream);
int hpri_maxpool2_fwd(const float* x, int x_cs, int x_coff, float* y, int y_cs, int y_coff, int N, int H, int W, int C,
                      hipStream_t stream);
int hpri_maxpool2_bwd(const float* x, int x_cs, int x_coff, const float* dy, int dy_cs, int dy_coff, float* dx,
                      int dx_cs, int dx_coff, int N, int H, int W, int C, int accumulate, hipStream_t stream);
/* bf16 mode (round 4): max-pooling over / into bf16 rows -- the skip tensors of the U-Nets (model_parts.py:40 after a bf16-mode
 * DoubleConv) exist as planes only.  _fwd_x16: input = plane 0 of the skip's planes, y (fp32) optional (NULL: planes only).
 * _bwd_x16: x_bf16 / dx_bf16 select the storage of the pool's input and of its gradient; dy is fp32. */
int hpri_maxpool2_fwd_x16(const void* x16, int x_cs, int x_coff, float* y, int y_cs, int y_coff, int N, int H, int W, int C,
                          void* planes, long long plane_stride, int pl_cs, int pl_coff, int pl_cw, int npl, hipStream_t stream);
int hpri_maxpool2_bwd_x16(const void* x, int x_bf16, int x_cs, int x_coff, const float* dy, int dy_cs, int dy_coff, void* dx,
                          int dx_bf16, int dx_cs, int dx_coff, int N, int H, int W, int C, int accumulate, hipStream_t stream);
int hpri_copy_slice(const float* src, int s_cs, int s_coff, float* dst, int d_cs, int d_coff, long long P, int C,
                    int accumulate, hipStream_t stream);
int hpri_copy_slice_any(const float* src, int s_cs, int s_coff, float* dst, int d_cs, int d_coff, long long P, int C,
                        int Cz, int accumulate, hipStream_t stream);
int hpri_fill_pad(float* dst, int cs, int coff, int N, int H, int W, int C, int y0, int y1, int x0, int x1,
                  hipStream_t stream);
/* F.pad with positive and/or negative widths in one pass (model_parts.py:77-80, the crop case): dst[n][y][x] = src[n][y-oy][x-ox]
 * inside the Hs x Ws source, 0 outside. */
int hpri_shift_copy(const float* src, int s_cs, int s_coff, int Hs, int Ws, float* dst, int d_cs, int d_coff, int N, int Hd,
                    int Wd, int oy, int ox, int C, int accumulate, hipStream_t stream);
int hpri_fill(float* dst, long long n, float value, hipStream_t stream);
int hpri_outconv_fwd(const float* x, int x_cs, int x_coff, const float* w, const float* b, float* y, int N,
                     long long P, int C, int K, hipStream_t stream);
int hpri_outconv_bwd_plan(int N, long long P, int C, int K, int* nblk, int* Cpart);
int hpri_outconv_bwd(const float* dy, const float* x, int x_cs, int x_coff, const float* w, float* dx, int dx_cs,
                     int dx_coff, int dx_cw, int dx_accumulate, float* dw, float* db, int accumulate_param_grads,
                     float* workspace, size_t ws_floats, int N, long long P, int C, int K, hipStream_t stream);
/* The head fused with the loss of PLTrainer.py:86: hpri_outconv_fwd_bce also leaves per-block fp64 partial sums of
 * BCEWithLogits(logits, target) (hpri_outconv_fwd_bce_blocks doubles; finish: hpri_bce_finish -> mean loss); hpri_outconv_bwd_bce
 * takes the LOGITS, the target and the device scalar arriving at the loss and forms (sigmoid(logits) - target) * g / n inside
 * the data- and weight-gradient kernels.  Three launches and two passes over the logits less than loss and head apart. */
size_t hpri_outconv_fwd_bce_blocks(int N, long long P);
int hpri_outconv_fwd_bce(const float* x, int x_cs, int x_coff, const float* w, const float* b, float* y, const float* target,
                         double* partial, size_t partial_doubles, int N, long long P, int C, int K, hipStream_t stream);
int hpri_bce_finish(const double* partial, int nblk, long long n, float* loss, hipStream_t stream);
int hpri_outconv_bwd_bce(const float* logits, const float* target, const float* gscale, const float* x, int x_cs, int x_coff,
                         const float* w, float* dx, int dx_cs, int dx_coff, int dx_cw, int dx_accumulate, float* dw, float* db,
                         int accumulate_param_grads, float* workspace, size_t ws_floats, int N, long long P, int C, int K,
                         hipStream_t stream);
/* The head of the bf16 mode (model_parts.py:96 / models.py:103,143 after a bf16-mode layer): the input is plane 0 of the last
 * activation's plane buffer -- bf16 NHWC rows, x_cs / x_coff in elements (multiples of 8), pad channels zero -- so no fp32 copy of
 * that tensor is written or read.  target != NULL: with the loss partials (forward) / `dy` holds the logits and the loss gradient is
 * formed inside the kernels (backward), as hpri_outconv_fwd_bce / hpri_outconv_bwd_bce.  dx: fp32, or (dx_bf16, one class) bf16 rows. */
int hpri_outconv_fwd_x16(const void* x16, int x_cs, int x_coff, const float* w, const float* b, float* y, const float* target,
                         double* partial, size_t partial_doubles, int N, long long P, int C, int K, hipStream_t stream);
int hpri_outconv_bwd_x16(const float* dy, const float* target, const float* gscale, const void* x16, int x_cs, int x_coff,
                         const float* w, void* dx, int dx_bf16, int dx_cs, int dx_coff, int dx_cw, int dx_accumulate, float* dw,
                         float* db, int accumulate_param_grads, float* workspace, size_t ws_floats, int N, long long P, int C, int K,
                         hipStream_t stream);
/* nn.Upsample(scale_factor=2, 'bilinear', align_corners=True) (model_parts.py:57; models.py:195) writing at a pixel
 * offset of a padded destination, its gather-form gradient, and the element-wise "attention" product x2*x1
 * (model_parts.py:84-85). */
int hpri_upsample2x_fwd(const float* x, int x_cs, int x_coff, float* y, int y_cs, int y_coff, int N, int H, int W, int H2,
                        int W2, int py0, int px0, int C, hipStream_t stream);
int hpri_upsample2x_bwd(const float* dy, int dy_cs, int dy_coff, float* dx, int dx_cs, int dx_coff, int N, int H, int W,
                        int H2, int W2, int py0, int px0, int C, int accumulate, hipStream_t stream);
int hpri_mul(const float* a, int a_cs, int a_coff, const float* b, int b_cs, int b_coff, float* out, int o_cs, int o_coff,
             long long P, int C, int accumulate, hipStream_t stream);
int hpri_synth_fill(float* dst, long long n, unsigned long long seed, int mode, float thr, float scale,
                    hipStream_t stream);

/* ---- the caller-side tail of a step (step.hip) ---------------------------------
 * What RootLightningModel does with the logits once the network returns them:
 *   hpri_bce_logits_fwd / _bwd   nn.BCEWithLogitsLoss() mean, `loss = self.f_criterion(pred, batch['mask'])`
 *                                (PLTrainer.py:86,109,131; params_HyperPRI.py:60) and its gradient
 *                                dlogits = (sigmoid(x) - y) * grad_out / n; `loss` and `grad_out` are device scalars.
 *   hpri_seg_counts              `seg = torch.sigmoid(pred.detach()) > threshold` and the TP/FP/FN/TN counts that
 *                                Accuracy / JaccardIndex / Dice (PLTrainer.py:62-68,88-91) reduce to;
 *                                counts[4] (int64: TP, FP, FN, TN) are accumulated, so one buffer serves an epoch.
 *   hpri_pr_curve_hist           PrecisionRecallCurve('binary', thresholds=500) (PLTrainer.py:542-543; torchmetrics
 *                                1.2.0 binned update): hist[2][T+1] int64, bin = #{k : thresholds[k] <= p}, accumulated.
 *   hpri_adam_step / hpri_sgd_step  optim.Adam / optim.SGD over every parameter tensor (PLTrainer.py:171-181):
 *                                host arrays of device pointers + element counts; `grad_scale` (nullable device
 *                                scalar) multiplies the gradients first (1/world_size after a sum all-reduce). */
size_t hpri_bce_workspace_doubles(long long n);
int hpri_bce_logits_fwd(const float* logits, const float* target, long long n, float* loss, double* workspace,
                        size_t ws_doubles, hipStream_t stream);
int hpri_bce_logits_bwd(const float* logits, const float* target, long long n, const float* grad_out, float* dlogits,
                        hipStream_t stream);
int hpri_seg_counts(const float* pred, const float* target, long long n, float threshold, int is_logits,
                    long long* counts, hipStream_t stream);
int hpri_pr_curve_hist(const float* pred, const float* target, long long n, const float* thresholds, int T,
                       int is_logits, long long* hist, hipStream_t stream);
int hpri_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                   const long long* numel, int ntensors, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, const float* grad_scale, hipStream_t stream);
int hpri_sgd_step(float* const* params, const float* const* grads, float* const* momentum_buf, const long long* numel,
                  int ntensors, float lr, float momentum, float weight_decay, int first_step, const float* grad_scale,
                  hipStream_t stream);

/* ---- ingest fast path (ingest.hip) ---------------------------------------------------
 * `HyperpriDataset.__getitem__` (dataset.py:261-271) loads an ENVI cube as (H, W, B), moves the band axis to the
 * front on the host and slices [hsi_lo:hsi_hi].  These two take the (H, W, B) array as it is:
 *   hpri_hwb_ingest  device (H,W,B) f32/f16 -> zero-padded channels-last fp32 [P][dst_cs], bands [lo, lo+C)
 *   hpri_hwb_h2d     host (pinned) fp32 (H,W,B) -> the same layout directly, as one async 2-D H2D copy
 *                    (pad channels must have been zeroed once by the caller). */
int hpri_hwb_ingest(const void* src, int src_dtype, float* dst, long long P, int B, int lo, int C, int dst_cs,
                    int dst_cw, hipStream_t stream);
int hpri_hwb_h2d(const float* host_src, float* dst, long long P, int B, int lo, int C, int dst_cs, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HYPERPRI_HIP_H */
