// Implicit-GEMM convolution (3x3 pad 1, or 1x1) over NHWC fp32 activations on the CDNA4 matrix
// cores, exact fp32 (v_mfma_f32_32x32x2_f32).  One kernel family serves
//   * Conv2d 3x3 forward               (reference model_parts.py:22,25; models.py:177)
//   * the CubeNET Conv3d(1->F,(D,3,3)) (models.py:169) == 3x3 conv over D input channels
//   * 3x3 data-gradient                (same kernel, flipped/transposed weight pack)
//   * Linear forward / data-gradient   (models.py:108,143: 1x1 conv over pixels)
//   * ConvTranspose2d k2 s2 forward    (model_parts.py:63: 1x1 GEMM + 2x2 pixel-shuffle store, E_D2S)
//   * ConvTranspose2d data-gradient    (2x2 patch gather, A_S2D)
//
// Tiling (per 256-thread workgroup = 4 waves, 2 workgroups per CU):
//   output tile  = TH x 32 pixels (TH = 2*WM rows) x BN = 64*WN channels; each wave owns 2 rows x 64 ch
//                  = 2x2 MFMA tiles of 32 pixels x 32 channels (64 accumulator VGPRs)
//   A (input)    : a (TH+KS-1) x (32+KS-1) pixel halo x 32 channels is staged ONCE per 32-channel chunk in
//                  LDS ([pixel][36] dwords, conflict-free ds_read_b128) and reused by all KS*KS taps
//   B (weights)  : one [32 k][BN] panel per (chunk, tap), double-buffered in LDS, prefetched to registers
//   k ordering   : inside an 8-channel group MFMA step j uses channel 4*half + j, so one ds_read_b128
//                  feeds four MFMAs (A and B agree on the permutation; fp32 result differs from a
//                  k-ascending chain only by summation order)
// Epilogue: + bias, NHWC store (or 2x2 scatter), optional accumulate, optional per-tile BatchNorm
// partial statistics (mean, M2, count) for the training-mode BN that follows every conv in the model.
#include "common.h"

struct ConvFwdArgs {
  const float* x; int x_cs; int x_coff;
  const float* wp;       // packed weights [chunks][T][32][Cout_pad]
  const float* bias;     // [Cout] or nullptr
  float* y; int y_cs; int y_coff;
  float4* stats;         // [N*tiles_y*tiles_x][Cout_pad] (mean, M2, count, 0) or nullptr
  int N, H, W;           // GEMM-M image: output pixels (DIRECT) / low-res pixels (S2D, D2S)
  int Cin_pad;           // K per tap, multiple of 8 (for S2D: 4*Cup)
  int Cout;              // valid output channels (for D2S: 4*Cup)
  int Cout_pad;          // multiple of BN
  int y_cw;              // channels written (>= Cout; extra ones get zeros)
  int tiles_x, tiles_y;
  int accumulate;        // y += result instead of y = result
  int H2, W2, py0, px0, Cup;  // S2D / D2S geometry: hi-res image dims, pad offsets, channels per tap
};

template <int KS, int WM, int WN, int AMODE, int EPI>
__global__ __launch_bounds__(256, 2) void conv_fwd_kernel(ConvFwdArgs a) {
  constexpr int T = KS * KS, PAD = KS / 2;
  constexpr int TH = 2 * WM, TW = 32, HH = TH + KS - 1, HW = TW + KS - 1, HP = HH * HW;
  constexpr int BN = 64 * WN;
  constexpr int CS = 36;                         // dwords per staged pixel (32 channels + 4 pad)
  constexpr int NLD_A = (HP * 8 + 255) / 256;    // float4 loads per thread per A chunk
  constexpr int NLD_B = (32 * BN / 4) / 256;     // float4 loads per thread per B panel
  __shared__ __attribute__((aligned(16))) float smem[HP * CS + 2 * 32 * BN];
  float* a_lds = smem;
  float* b_lds = smem + HP * CS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  int tile = blockIdx.x;
  const int tx = tile % a.tiles_x; tile /= a.tiles_x;
  const int ty = tile % a.tiles_y;
  const int img = tile / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int nb = blockIdx.y;

#ifdef HPRI_STAGGER_F
  {
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
    if ((lin >> 8) & 1) { for (int i = 0; i < HPRI_STAGGER_F; ++i) __builtin_amdgcn_s_sleep(100); }
  }
#endif
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = (a.Cin_pad + 31) >> 5;
  const int S = nchunks * T;
  const float* wpanel = a.wp + (size_t)nb * BN;

  // ---- B panel prefetch registers (macros, not lambdas: keeps breg[] in VGPRs) ----
  f32x4 breg[NLD_B];
#define LOAD_PANEL(s_)                                                                              \
  _Pragma("unroll") for (int p = 0; p < NLD_B; ++p) {                                               \
    const int f = tid + p * 256;                                                                    \
    const int row = f / (BN / 4), c4 = f % (BN / 4);                                                \
    breg[p] = *reinterpret_cast<const f32x4*>(wpanel + ((size_t)(s_) * 32 + row) * a.Cout_pad + c4 * 4); \
  }
#define STORE_PANEL(buf_)                                                                           \
  _Pragma("unroll") for (int p = 0; p < NLD_B; ++p) {                                               \
    const int f = tid + p * 256;                                                                    \
    const int row = f / (BN / 4), c4 = f % (BN / 4);                                                \
    *reinterpret_cast<f32x4*>(b_lds + (buf_) * 32 * BN + row * BN + c4 * 4) = breg[p];             \
  }

  // ---- A halo chunk: global -> registers (prefetched one chunk ahead) -> LDS ----
  f32x4 areg[NLD_A];
#define LOAD_A(c0_)                                                                                   \
  {                                                                                                   \
    const int kq = min(8, (a.Cin_pad - (c0_)) >> 2); /* valid float4 per pixel in this chunk */       \
    _Pragma("unroll") for (int p = 0; p < NLD_A; ++p) {                                               \
      const int f = tid + p * 256;                                                                    \
      const int pix = f >> 3, q = f & 7;                                                              \
      f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                                 \
      if (pix < HP && q < kq) {                                                                       \
        const int hy = pix / HW, hx = pix - hy * HW;                                                  \
        const int iy = y0 + hy - PAD, ix = x0 + hx - PAD;                                             \
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {                                             \
          if (AMODE == HPRI_A_DIRECT) {                                                               \
            v = *reinterpret_cast<const f32x4*>(                                                      \
                a.x + ((size_t)(img * a.H + iy) * a.W + ix) * a.x_cs + a.x_coff + (c0_) + q * 4);     \
          } else { /* S2D: k = tap*Cup + co ; source pixel (2*iy + t_y + py0, 2*ix + t_x + px0) */    \
            const int k4 = (c0_) + q * 4;                                                             \
            const int tp = k4 / a.Cup, co = k4 - tp * a.Cup;                                          \
            const int sy = 2 * iy + (tp >> 1) + a.py0, sx = 2 * ix + (tp & 1) + a.px0;                \
            v = *reinterpret_cast<const f32x4*>(                                                      \
                a.x + ((size_t)(img * a.H2 + sy) * a.W2 + sx) * a.x_cs + a.x_coff + co);              \
          }                                                                                           \
        }                                                                                             \
      }                                                                                               \
      areg[p] = v;                                                                                    \
    }                                                                                                 \
  }
#define STORE_A()                                                                                     \
  _Pragma("unroll") for (int p = 0; p < NLD_A; ++p) {                                                 \
    const int f = tid + p * 256;                                                                      \
    const int pix = f >> 3, q = f & 7;                                                                \
    if (pix < HP) *reinterpret_cast<f32x4*>(a_lds + pix * CS + q * 4) = areg[p];                      \
  }

  const int a_base = ((wm * 2) * HW + li) * CS + lh * 4;
  const int b_base = lh * 4 * BN + wn * 64 + li;

#define MFMA_GROUP(g_)                                                                                \
  {                                                                                                   \
    f32x4 af[2];                                                                                      \
    float bf[2][4];                                                                                   \
    _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                  \
        af[mt] = *reinterpret_cast<const f32x4*>(ap + mt * HW * CS + (g_) * 8);                       \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) bf[nt][j] = bp[((g_) * 8 + j) * BN + nt * 32];  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                     \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                              \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                          \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][j], bf[nt][j], acc[mt][nt], 0, 0, 0); \
  }

  LOAD_PANEL(0)
  LOAD_A(0)
  for (int s = 0; s < S; ++s) {
    const int chunk = s / T, tap = s - chunk * T;
    if (tap == 0) {
      __syncthreads();                 // everyone is done reading the previous A chunk
      STORE_A()
    }
    STORE_PANEL(s & 1)
    __syncthreads();                   // panel s (and the A chunk) visible
    if (s + 1 < S) { LOAD_PANEL(s + 1) }
    if (tap == T - 1 && chunk + 1 < nchunks) { LOAD_A((chunk + 1) * 32) }   // lands during this panel's MFMAs

    const int kg = min(4, (a.Cin_pad - chunk * 32) >> 3);
    const int dy = tap / KS, dx = tap - dy * KS;
    const float* ap = a_lds + a_base + (dy * HW + dx) * CS;
    const float* bp = b_lds + (s & 1) * 32 * BN + b_base;
    for (int g = 0; g < kg; ++g) MFMA_GROUP(g)
  }
#undef MFMA_GROUP
#undef LOAD_A
#undef STORE_A
#undef LOAD_PANEL
#undef STORE_PANEL

  // ------------------------------- epilogue -------------------------------
  // acc[mt][nt][r]: pixel row = wm*2+mt, pixel col = (r&3) + 8*(r>>2) + 4*lh, channel = nb*BN + wn*64 + nt*32 + li
  float bv[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = nb * BN + wn * 64 + nt * 32 + li;
    float b = 0.f;
    if (a.bias != nullptr && n < a.Cout) b = (EPI == HPRI_E_D2S) ? a.bias[n % a.Cup] : a.bias[n];
    bv[nt] = b;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] += b;
  }

#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int iy = y0 + wm * 2 + mt;
    if (iy >= a.H) continue;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = nb * BN + wn * 64 + nt * 32 + li;
      if (EPI == HPRI_E_DIRECT) {
        if (n >= a.y_cw) continue;
        float* yrow = a.y + ((size_t)(img * a.H + iy) * a.W) * a.y_cs + a.y_coff + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ix = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (ix < a.W) {
            float* p = yrow + (size_t)ix * a.y_cs;
            float v = (n < a.Cout) ? acc[mt][nt][r] : 0.f;
            if (a.accumulate) v += *p;
            *p = v;
          }
        }
      } else {  // D2S: n = tap*Cup + co -> hi-res pixel (2*iy + t_y + py0, 2*ix + t_x + px0), channel co
        if (n >= a.Cout) continue;
        const int tap = n / a.Cup, co = n - tap * a.Cup;
        const int sy = 2 * iy + (tap >> 1) + a.py0;
        float* yrow = a.y + ((size_t)(img * a.H2 + sy) * a.W2) * a.y_cs + a.y_coff + co;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ix = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (ix < a.W) {
            float* p = yrow + (size_t)(2 * ix + (tap & 1) + a.px0) * a.y_cs;
            float v = acc[mt][nt][r];
            if (a.accumulate) v += *p;
            *p = v;
          }
        }
      }
    }
  }

  if (a.stats != nullptr) {
    // per-tile, per-channel (mean, M2, count) over the tile's valid pixels; two passes over the
    // accumulators (sum, then squared deviations from the tile mean) -- no E[x^2]-E[x]^2 cancellation.
    float* red = smem;                      // [4 waves][64 channels], reuses the A staging area
    const int vrows = min(TH, a.H - y0), vcols = min(TW, a.W - x0);
    const float cnt = (float)(vrows * vcols);
    float mean[2];
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float sacc = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const bool rowok = (y0 + wm * 2 + mt) < a.H;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ix = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (rowok && ix < a.W) {
              const float v = acc[mt][nt][r];
              if (pass == 0) sacc += v;
              else { const float d = v - mean[nt]; sacc += d * d; }
            }
          }
        }
        sacc += __shfl_xor(sacc, 32);
        if (lh == 0) red[wave * 64 + nt * 32 + li] = sacc;
      }
      __syncthreads();
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float t = 0.f;
#pragma unroll
        for (int m = 0; m < WM; ++m) t += red[(m * WN + wn) * 64 + nt * 32 + li];
        if (pass == 0) mean[nt] = t / cnt;
        else if (wm == 0 && lh == 0) {
          const int n = nb * BN + wn * 64 + nt * 32 + li;
          a.stats[(size_t)blockIdx.x * a.Cout_pad + n] = make_float4(mean[nt], t, cnt, 0.f);
        }
      }
      __syncthreads();
    }
  }
}

template <int KS, int WM, int WN, int AMODE, int EPI>
static int launch_conv(const ConvFwdArgs& a0, hipStream_t stream) {
  ConvFwdArgs a = a0;
  constexpr int TH = 2 * WM, BN = 64 * WN;
  a.tiles_x = hpri_cdiv(a.W, 32);
  a.tiles_y = hpri_cdiv(a.H, TH);
  dim3 grid((unsigned)(a.N * a.tiles_y * a.tiles_x), (unsigned)(a.Cout_pad / BN));
  hipLaunchKernelGGL((conv_fwd_kernel<KS, WM, WN, AMODE, EPI>), grid, dim3(256), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// Tile-shape choice shared by the launcher and the workspace/partials sizing query.
static inline void conv_cfg(int Cout_pad, int* wm, int* wn) {
  if (Cout_pad % 128 == 0) { *wm = 2; *wn = 2; } else { *wm = 4; *wn = 1; }
}

extern "C" int hpri_conv_fwd_tiles(int N, int H, int W, int Cout_pad) {
  int wm, wn; conv_cfg(Cout_pad, &wm, &wn);
  return N * hpri_cdiv(H, 2 * wm) * hpri_cdiv(W, 32);
}

extern "C" int hpri_conv_fwd(const float* x, int x_cs, int x_coff, const float* wp, const float* bias,
                             float* y, int y_cs, int y_coff, float* stats,
                             int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw,
                             int KS, int amode, int epi, int accumulate,
                             int H2, int W2, int py0, int px0, int Cup, hipStream_t stream) {
  HPRI_REQUIRE(x && wp && y, "conv_fwd: null pointer");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0, "conv_fwd: empty image");
  HPRI_REQUIRE(Cin_pad > 0 && Cin_pad % 8 == 0, "conv_fwd: Cin_pad must be a positive multiple of 8");
  HPRI_REQUIRE(Cout_pad % 64 == 0 && Cout <= Cout_pad && Cout > 0, "conv_fwd: Cout_pad must be a multiple of 64 >= Cout");
  HPRI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0, "conv_fwd: input channel stride/offset must be multiples of 4");
  HPRI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)wp & 15) == 0, "conv_fwd: pointers must be 16-byte aligned");
  HPRI_REQUIRE(KS == 1 || KS == 3, "conv_fwd: kernel size must be 1 or 3");
  ConvFwdArgs a;
  a.x = x; a.x_cs = x_cs; a.x_coff = x_coff; a.wp = wp; a.bias = bias;
  a.y = y; a.y_cs = y_cs; a.y_coff = y_coff; a.stats = reinterpret_cast<float4*>(stats);
  a.N = N; a.H = H; a.W = W; a.Cin_pad = Cin_pad; a.Cout = Cout; a.Cout_pad = Cout_pad;
  a.y_cw = y_cw < Cout ? Cout : y_cw; a.tiles_x = a.tiles_y = 0; a.accumulate = accumulate;
  a.H2 = H2; a.W2 = W2; a.py0 = py0; a.px0 = px0; a.Cup = Cup;
  if (epi == HPRI_E_DIRECT) {
    HPRI_REQUIRE(a.y_cw + y_coff <= y_cs, "conv_fwd: output channels exceed the channel stride");
  }
  if (amode == HPRI_A_S2D || epi == HPRI_E_D2S) {
    HPRI_REQUIRE(KS == 1, "conv_fwd: S2D/D2S need KS == 1");
    HPRI_REQUIRE(Cup > 0 && Cup % 4 == 0, "conv_fwd: Cup must be a positive multiple of 4");
    HPRI_REQUIRE(py0 >= 0 && px0 >= 0 && 2 * H + py0 <= H2 && 2 * W + px0 <= W2, "conv_fwd: 2x2 patch grid exceeds the hi-res image");
    HPRI_REQUIRE(stats == nullptr, "conv_fwd: statistics epilogue is only available for direct stores");
    if (amode == HPRI_A_S2D) HPRI_REQUIRE(Cin_pad == 4 * Cup, "conv_fwd: S2D needs Cin_pad == 4*Cup");
    if (epi == HPRI_E_D2S) HPRI_REQUIRE(Cout == 4 * Cup, "conv_fwd: D2S needs Cout == 4*Cup");
    HPRI_REQUIRE(!(amode == HPRI_A_S2D && epi == HPRI_E_D2S), "conv_fwd: S2D and D2S are exclusive");
  }
  int wm, wn; conv_cfg(Cout_pad, &wm, &wn);
#define HPRI_DISPATCH(KS_, AM_, EP_)                                                        \
  return (wm == 2) ? launch_conv<KS_, 2, 2, AM_, EP_>(a, stream) : launch_conv<KS_, 4, 1, AM_, EP_>(a, stream)
  if (KS == 3) { HPRI_DISPATCH(3, HPRI_A_DIRECT, HPRI_E_DIRECT); }
  if (amode == HPRI_A_S2D) { HPRI_DISPATCH(1, HPRI_A_S2D, HPRI_E_DIRECT); }
  if (epi == HPRI_E_D2S) { HPRI_DISPATCH(1, HPRI_A_DIRECT, HPRI_E_D2S); }
  HPRI_DISPATCH(1, HPRI_A_DIRECT, HPRI_E_DIRECT);
#undef HPRI_DISPATCH
}
