/* hyperpri_hip_diag.h -- entry points of the DIAGNOSTICS build of libhyperpri_hip.so only.
 *
 * Superseded kernel generations and switches that measured neutral or negative are kept for A/B measurements
 * (tools/v3_bench.py, tools/wino_check.py, tools/v2_stamps.py, ...) but are not part of the product library:
 *
 *     HPRI_DIAG=1 python -m hyperpri_amd.build      ->  hyperpri_amd/lib/libhyperpri_hip_diag.so   (-DHPRI_DIAG_KERNELS)
 *     HPRI_DIAG=1 python tools/v3_bench.py          (hyperpri_amd._lib then loads that library and binds this header too)
 *
 * The default library (include/hyperpri_hip.h) carries ONE kernel family per precision mode and form:
 *   fp32 3x3 fwd / dgrad  conv_wino4.hip         (first form conv_wino.hip: here)
 *   bf16 3x3 fwd / dgrad  conv_bf16v3.hip        (conv_bf16v2.hip: here; BatchNorm-backward sums in its epilogue: here)
 * Same conventions as hyperpri_hip.h (caller-owned buffers, error codes, stream-explicit). */
#ifndef HYPERPRI_HIP_DIAG_H
#define HYPERPRI_HIP_DIAG_H
#include "hyperpri_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* A non-blocking stream of the lowest priority the current device offers (*priority receives it); the caller owns it. */
int hpri_stream_create_low_priority(void** stream, int* priority);

/* A non-blocking stream of the lowest priority the current device offers (*priority receives it); the caller owns it. */
int hpri_stream_destroy(void* stream);

int hpri_wino_pack(const float* w, float* up, const float* colscale, int mode, int K, int Ncols, int Ncols_pad, int src_d1,
                   hipStream_t stream);

int hpri_conv_wino_plan(int N, int H, int W, int* stat_tiles);

int hpri_conv_wino(const float* x, int x_cs, int x_coff, const float* up, const float* bias, float* y, int y_cs, int y_coff,
                   float* stats, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw, int accumulate,
                   hipStream_t stream);

/* diagnostic builds (-DHPRI_STAMPS, tools/build_wino4_diag.sh) only; the product library returns HPRI_ERR_UNSUPPORTED */
int hpri_wino4_set_stamps(unsigned long long* stamps);

int hpri_conv_bf16v2_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int* ksplit, int* stat_tiles,
                          size_t* ws_floats);

int hpri_conv_bf16v2(const void* xp, long long x_plane, int x_cs, int x_coff, const void* wp, const float* bias, float* y,
                     int y_cs, int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw,
                     int accumulate, int split, float* ws, size_t ws_floats, hipStream_t stream);

int hpri_conv_bf16v3_dbg(const void* xp, long long x_plane, int x_cs, int x_coff, const void* wp, const float* bias, float* y,
                         int y_cs, int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw,
                         int accumulate, int split, float* ws, size_t ws_floats, unsigned long long* stamps,
                         int stagger_cycles, hipStream_t stream);

/* The data gradient of a 3x3 layer in the bf16 plane mode whose input x = ReLU(BN(bn_x16)) has no other consumer, with that
 * BatchNorm's backward reduction taken in the epilogue (bf16 counterpart of hpri_conv_wino4_bnred): bn_x16 = the pre-BN tensor as
 * bf16 (what hpri_conv_bf16v3 wrote with accumulate bit 2; same pixels as y, stride / offset in elements, multiples of 4);
 * bn_part[stat_tiles][2][bn_cpart] (stat_tiles from hpri_conv_bf16v3_plan, which must report ksplit 1; bn_cpart >= Cout) receives
 * sum g*[y>0] and sum g*[y>0]*xhat per tile.  Finish with hpri_bn_relu_bwd_fused. */
int hpri_conv_bf16v3_bnred(const void* xp, int x_cs, int x_coff, const void* wp, float* y, int y_cs, int y_coff, int N, int H,
                           int W, int Cin_pad, int Cout, int Cout_pad, int y_cw, const void* bn_x16, int bn_x_cs, int bn_x_coff,
                           const float* bn_mean, const float* bn_invstd, const float* bn_scale, const float* bn_shift,
                           int bn_relu, float* bn_part, int bn_cpart, hipStream_t stream);

int hpri_bn_relu_bwd_fused_x16(const float* partials, int part_blocks, int part_cpart, const float* dy, int dy_cs, int dy_coff,
                               const void* x16, int x_cs, int x_coff, float* dx, int dx_cs, int dx_coff, const float* mean,
                               const float* invstd, const float* scale, const float* shift, float* dgamma, float* dbeta,
                               int accumulate_param_grads, float* dbias, int accumulate_dbias, float* workspace, size_t ws_floats,
                               long long P, long long pix_per_group, int C, int Cw, int relu, int use_batch_stats, void* planes,
                               long long plane_stride, int pl_cs, int pl_coff, int pl_cw, int npl, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
