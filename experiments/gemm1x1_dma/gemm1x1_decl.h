/* declarations to paste into include/hyperpri_hip.h when this kernel is built into the library */
/* ---- fp32 pixel GEMM with both operands by LDS-DMA (gemm1x1.hip): the dense forms of nn.ConvTranspose2d(k=2, s=2)
 * (model_parts.py:63-64): forward (epi = HPRI_E_D2S: columns n = tap*Cup + co scattered as 2x2 patches into the hi-res view,
 * bias[co]) and data gradient (amode = HPRI_A_S2D: A gathered from the hi-res gradient).  B from hpri_gemm1x1_pack (mode 0:
 * convT forward, 1: convT data gradient, 2: plain [Ncols][K]).  x_floats = floats in the A source view (DMA descriptor range). */
size_t hpri_gemm1x1_packed_floats(int K, int Ncols);
int hpri_gemm1x1_pack(const float* w, float* wp, int mode, int K, int Ncols, int Cin, int Cup, hipStream_t stream);
int hpri_gemm1x1(const float* x, int x_cs, int x_coff, long long x_floats, const float* wp, const float* bias, float* y, int y_cs,
                 int y_coff, int N, int H, int W, int K_pad, int Ncols, int y_cw, int amode, int epi, int H2, int W2, int py0,
                 int px0, int Cup, int accumulate, hipStream_t stream);

