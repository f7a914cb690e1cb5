// fp32 "pixel GEMM" on v_mfma_f32_32x32x2_f32 with both operands by LDS-DMA:  C[pixel][n] = sum_k A[pixel][k] B[k][n].
// Serves the two dense forms of nn.ConvTranspose2d(k=2, s=2) (reference model_parts.py:63-64, `self.up`):
//   forward        A = x1 (low-res NHWC), K = Cin, n = tap*Cup + co, epilogue scatters the 2x2 patches into the (padded)
//                  hi-res slice of the concat buffer and adds bias[co]                                  (EPI = D2S)
//   data gradient  A[pixel][k = tap*Cup + co] = dY[hi-res pixel (2y + tap>>1, 2x + tap&1)][co] gathered through the DMA
//                  source address, n = cin, plain (optionally accumulating) NHWC store                  (AMODE = S2D)
// The round-1 implicit-GEMM kernel runs these at 0.55-0.63 of the fp32 MFMA peak: with one tap instead of nine it stages an A
// chunk through VGPRs and passes two barriers for every 8 MFMAs per wave.  This kernel is built like conv_wino4.hip without
// the transforms:
//   workgroup  256 threads = 4 waves, 128 pixels x 256 columns, 40 KB of LDS: two per CU (VGPR-limited)
//   wave w     columns [64 w, 64 w + 64) of all 128 pixels: 4 x 2 accumulator tiles (128 VGPRs); per 8-channel stage
//              4 + 2 ds_read_b128 and 32 MFMAs, no vector arithmetic at all
//   A          [pixel][32 ch] rows of 128 B, double-buffered per 32-channel chunk, quads XOR-swizzled with (pixel>>1)&7
//              through the DMA source (conflict-free ds_read_b128); one barrier per chunk (128 MFMAs per wave)
//   B          packed k-innermost [K/8][N][8] (hpri_gemm1x1_pack); the wave's panel (64 n x 8 k = 2 KB) is private and
//              single-buffered: read into 8 registers at the start of a stage, then overwritten by the next stage's DMA
//   DMA        buffer_load ... lds with 32-bit per-lane offsets; pixels beyond the tile's end and channels beyond K are
//              zero-filled by the descriptor's range check
#include "common.h"

struct Gemm1x1Args {
  const float* x; int x_cs, x_coff;          // A source (NHWC view): low-res input (direct) or hi-res gradient (S2D)
  const float* wp;                           // packed B, k innermost
  const float* bias;                         // [Cup] (D2S) or [Ncols] (direct) or nullptr
  float* y; int y_cs, y_coff;
  int N, H, W;                               // low-res grid: pixels = N*H*W
  int K_pad, Ncols, Ncols_pad, y_cw, accumulate;
  int amode, epi;                            // HPRI_A_S2D gather / HPRI_E_D2S scatter
  int H2, W2, py0, px0, Cup;                 // hi-res grid and patch origin (S2D / D2S)
  long long x_bytes;                         // size of the A source view in bytes (descriptor range)
};

#define G1_M 128
#define G1_A_BYTES (G1_M * 128)
#define G1_B_WAVE 2048

__global__ __launch_bounds__(256, 2) void gemm1x1_kernel(Gemm1x1Args a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * G1_A_BYTES + 4 * G1_B_WAVE];
  unsigned char* b_lds = smem + 2 * G1_A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int nbc = a.Ncols_pad >> 8;                 // 256-column blocks of one pixel tile are adjacent in launch order
  const int bx = blockIdx.x / nbc, nb = blockIdx.x - bx * nbc;
  const long long P = (long long)a.N * a.H * a.W;
  const long long m0 = (long long)bx * G1_M;
  const int n_w = nb * 256 + wave * 64;             // first column of this wave

  // ---- A DMA: piece i (16 per chunk) covers pixel rows [8i, 8i+8); lane -> row 8i + (lane>>3), physical quad lane&7 ----
  const hpri_rsrc_t rs_a = HPRI_MAKE_RSRC(a.x + a.x_coff, (unsigned)a.x_bytes);   // < 4 GiB - 64 KiB (host check)
  unsigned aoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (q * 4 + wave) * 8 + (lane >> 3);
    const long long p = m0 + row;
    const int lquad = (lane & 7) ^ ((row >> 1) & 7);
    unsigned off = HPRI_DMA_OOB;
    if (p < P) {
      if (a.amode == HPRI_A_S2D) {
        const int x_ = (int)(p % a.W); const long long t_ = p / a.W; const int y_ = (int)(t_ % a.H); const int img = (int)(t_ / a.H);
        off = (unsigned)((((long long)img * a.H2 + 2 * y_ + a.py0) * a.W2 + 2 * x_ + a.px0) * a.x_cs * 4 + lquad * 16);
      } else {
        off = (unsigned)(p * a.x_cs * 4 + lquad * 16);
      }
    }
    aoff[q] = off;
  }
  const int lq = lane & 7;
#define LOAD_A(chunk_)                                                                                                \
  {                                                                                                                   \
    unsigned char* la_ = smem + ((chunk_) & 1) * G1_A_BYTES;                                                          \
    int so_ = (chunk_) * 128;                     /* direct: channel offset of the chunk */                           \
    if (a.amode == HPRI_A_S2D) {                  /* gather: k = tap*Cup + co; a chunk lies inside one tap (Cup % 32 == 0) */ \
      const int k0_ = (chunk_) * 32, tap_ = k0_ / a.Cup, co_ = k0_ - tap_ * a.Cup;                                    \
      so_ = (((tap_ >> 1) * a.W2 + (tap_ & 1)) * a.x_cs + co_) * 4;                                                   \
    }                                                                                                                 \
    const bool tail_ = ((chunk_) * 32 + 32) > a.K_pad;                                                                \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                   \
      unsigned vo_ = aoff[q];                                                                                         \
      if (tail_) {                                                                                                    \
        const int row_ = (q * 4 + wave) * 8 + (lane >> 3);                                                            \
        if (((chunk_) * 32 + (lq ^ ((row_ >> 1) & 7)) * 4) >= a.K_pad) vo_ = HPRI_DMA_OOB;                            \
      }                                                                                                               \
      HPRI_LDS_DMA16(rs_a, la_ + (q * 4 + wave) * 1024, vo_, so_);                                                    \
    }                                                                                                                 \
  }
  // ---- B DMA: stage s, piece p (2 per stage): lane -> row n = 32 p + (lane>>1), physical 16-byte half lane&1 holding
  //      logical half (lane&1) ^ ((n>>3)&1) ----
  const hpri_rsrc_t rs_b = HPRI_MAKE_RSRC(a.wp, (unsigned)((long long)(a.K_pad >> 3) * a.Ncols_pad * 8 * 4));
  const unsigned goff0 = (unsigned)((n_w + (lane >> 1)) * 8 + 4 * ((lane & 1) ^ ((lane >> 4) & 1))) * 4u;
  const int gstage = a.Ncols_pad * 8 * 4;           // bytes per 8-channel stage
  unsigned char* bw = b_lds + wave * G1_B_WAVE;
#define LOAD_B(s_)                                                                                                    \
  {                                                                                                                   \
    HPRI_LDS_DMA16(rs_b, bw, goff0, (s_) * gstage);                                                                   \
    HPRI_LDS_DMA16(rs_b, bw + 1024, goff0, (s_) * gstage + 1024);                                                     \
  }
  const int boff = li * 32 + ((lh ^ ((li >> 3) & 1)) << 4);
  int pre[4];                                       // XOR-form addresses of this lane's four A rows (pixel mt*32 + li)
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int row = mt * 32 + li;
    pre[mt] = row * 128 + (((row >> 1) & 7) << 4);
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int nstages = a.K_pad >> 3;
  const int nchunks = (a.K_pad + 31) >> 5;
  LOAD_A(0)
  LOAD_B(0)
#define WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
  f32x4 bfr[2];
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int aboff = (chunk & 1) * G1_A_BYTES;
    const int sg = min(4, nstages - chunk * 4);
    for (int g = 0; g < sg; ++g) {
      const int s = chunk * 4 + g;
      // this stage's weights were issued one stage ago, BEFORE that stage's four A pieces (if any): in-order completion
      if (g == 1 && chunk + 1 < nchunks) WAIT_VM(4); else WAIT_VM(0);
      if (g == 0) __builtin_amdgcn_s_barrier();     // the chunk's A rows are visible; everyone has left the previous chunk
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bfr[nt] = *reinterpret_cast<const f32x4*>(bw + nt * 1024 + boff);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < nstages) LOAD_B(s + 1)
      if (g == 0 && chunk + 1 < nchunks) LOAD_A(chunk + 1)
      __builtin_amdgcn_sched_barrier(0);
      const int q16 = (2 * g + lh) << 4;
      f32x4 av[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(smem + ((pre[mt] ^ q16) + aboff));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt][j], bfr[nt][j], acc[mt][nt], 0, 0, 0);
    }
  }
#undef WAIT_VM
#undef LOAD_A
#undef LOAD_B

  // ---- epilogue: acc[mt][nt][r] = C[pixel m0 + mt*32 + (r&3) + 8*(r>>2) + 4*lh][column n_w + nt*32 + li]; 32 lanes store 128
  //      contiguous bytes.  The pixel -> address arithmetic (a division by W and H for the 2x2 scatter) is done once per group
  //      of four consecutive pixels, not once per store ----
  int cof[2], tapy[2], tapx[2];
  bool nok[2];
  float bv[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = n_w + nt * 32 + li;
    int co = n, tap = 0;
    if (a.epi == HPRI_E_D2S) { tap = n / a.Cup; co = n - tap * a.Cup; }
    cof[nt] = co; tapy[nt] = tap >> 1; tapx[nt] = tap & 1;
    nok[nt] = n < a.Ncols && co < a.y_cw;
    bv[nt] = (a.bias != nullptr && n < a.Ncols) ? a.bias[co] : 0.f;
  }
  const int Pi = (int)P;                            // P < 2^31 (host check)
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int p0 = (int)m0 + mt * 32 + 8 * rg + 4 * lh;
      int x_ = 0, y_ = 0, img = 0;
      if (a.epi == HPRI_E_D2S) { x_ = p0 % a.W; const int t_ = p0 / a.W; y_ = t_ % a.H; img = t_ / a.H; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = p0 + i;
        size_t pix;                                 // output pixel index (tap offset added per column group)
        if (a.epi == HPRI_E_D2S) {
          pix = ((size_t)img * a.H2 + 2 * y_ + a.py0) * a.W2 + 2 * x_ + a.px0;
          if (++x_ == a.W) { x_ = 0; if (++y_ == a.H) { y_ = 0; ++img; } }
        } else {
          pix = (size_t)p;
        }
        if (p < Pi) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            if (nok[nt]) {
              const size_t o = (pix + (size_t)tapy[nt] * a.W2 + tapx[nt]) * a.y_cs + a.y_coff + cof[nt];
              const float v = acc[mt][nt][rg * 4 + i] + bv[nt];
              a.y[o] = a.accumulate ? a.y[o] + v : v;
            }
        }
      }
    }
}

// B packed with k innermost: wp[(k/8 * Ncols_pad + n) * 8 + k%8].
//   mode 0 (convT forward):       B[k = cin][n = tap*Cup + co] = W[cin][co][tap]      (W: ConvTranspose2d weight [Cin][Cup][2][2])
//   mode 1 (convT data gradient): B[k = tap*Cup + co][n = cin] = W[cin][co][tap]
//   mode 2 (plain):               B[k][n] = W[n][k]                                   (W: [Ncols][K], Conv2d 1x1 / Linear)
__global__ void gemm1x1_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int mode, int K, int K_pad, int Ncols,
                                    int Ncols_pad, int Cin, int Cup) {
  const size_t total = (size_t)K_pad * Ncols_pad;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int kk = (int)(idx & 7);
    const int n = (int)((idx >> 3) % Ncols_pad);
    const int k = (int)(idx / ((size_t)Ncols_pad * 8)) * 8 + kk;
    float v = 0.f;
    if (k < K && n < Ncols) {
      if (mode == 0) { const int tap = n / Cup, co = n - tap * Cup; v = w[((size_t)k * Cup + co) * 4 + tap]; }
      else if (mode == 1) { const int tap = k / Cup, co = k - tap * Cup; v = w[((size_t)n * Cup + co) * 4 + tap]; }
      else v = w[(size_t)n * K + k];
    }
    wp[idx] = v;
  }
}

extern "C" size_t hpri_gemm1x1_packed_floats(int K, int Ncols) {
  return (size_t)(hpri_cdiv(K, 8) * 8) * (hpri_cdiv(Ncols, 256) * 256);
}

extern "C" int hpri_gemm1x1_pack(const float* w, float* wp, int mode, int K, int Ncols, int Cin, int Cup, hipStream_t stream) {
  HPRI_REQUIRE(w && wp && K > 0 && Ncols > 0 && mode >= 0 && mode <= 2, "gemm1x1_pack: bad arguments");
  if (mode == 0) HPRI_REQUIRE(Cup > 0 && Ncols == 4 * Cup && K == Cin, "gemm1x1_pack: mode 0 needs K = Cin, Ncols = 4*Cup");
  if (mode == 1) HPRI_REQUIRE(Cup > 0 && K == 4 * Cup && Ncols == Cin, "gemm1x1_pack: mode 1 needs K = 4*Cup, Ncols = Cin");
  const int K_pad = hpri_cdiv(K, 8) * 8, Ncols_pad = hpri_cdiv(Ncols, 256) * 256;
  const size_t total = (size_t)K_pad * Ncols_pad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gemm1x1_pack_kernel, dim3(blocks), dim3(256), 0, stream, w, wp, mode, K, K_pad, Ncols, Ncols_pad, Cin, Cup);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// C = A B (+ bias) over N*H*W pixels.  amode HPRI_A_DIRECT: A = x (NHWC view, channels [K, K_pad) zero, K_pad % 8 == 0);
// HPRI_A_S2D: A gathered from the hi-res view x = dY (H2 x W2, patch origin py0/px0, K_pad = 4*Cup, Cup % 32 == 0).
// epi HPRI_E_DIRECT: y[pixel][n]; HPRI_E_D2S: n = tap*Cup + co scattered to the hi-res view y (Ncols = 4*Cup, bias[co]).
// x_floats: number of floats in the A source view from its first element (range of the DMA descriptor).
extern "C" int hpri_gemm1x1(const float* x, int x_cs, int x_coff, long long x_floats, const float* wp, const float* bias, float* y,
                            int y_cs, int y_coff, int N, int H, int W, int K_pad, int Ncols, int y_cw, int amode, int epi, int H2,
                            int W2, int py0, int px0, int Cup, int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(x && wp && y, "gemm1x1: null pointer");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0 && K_pad > 0 && K_pad % 8 == 0 && Ncols > 0, "gemm1x1: bad sizes");
  HPRI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)wp & 15) == 0, "gemm1x1: A / B alignment");
  HPRI_REQUIRE(x_floats > 0 && x_floats * 4 < (1ll << 32) - 65536, "gemm1x1: the A source view must be smaller than 4 GiB");
  if (amode == HPRI_A_S2D) {
    HPRI_REQUIRE(Cup > 0 && Cup % 32 == 0 && K_pad == 4 * Cup, "gemm1x1: S2D needs K_pad = 4*Cup, Cup % 32 == 0");
    HPRI_REQUIRE(py0 >= 0 && px0 >= 0 && 2 * H + py0 <= H2 && 2 * W + px0 <= W2, "gemm1x1: patch grid exceeds the hi-res image");
    HPRI_REQUIRE(x_coff + Cup <= x_cs, "gemm1x1: gathered channels exceed the channel stride");
  } else {
    HPRI_REQUIRE(x_coff + K_pad <= x_cs, "gemm1x1: input channels exceed the channel stride");
  }
  if (epi == HPRI_E_D2S) {
    HPRI_REQUIRE(Cup > 0 && Ncols == 4 * Cup, "gemm1x1: D2S needs Ncols = 4*Cup");
    HPRI_REQUIRE(py0 >= 0 && px0 >= 0 && 2 * H + py0 <= H2 && 2 * W + px0 <= W2, "gemm1x1: patch grid exceeds the hi-res image");
  }
  Gemm1x1Args a;
  a.x = x; a.x_cs = x_cs; a.x_coff = x_coff; a.wp = wp; a.bias = bias; a.y = y; a.y_cs = y_cs; a.y_coff = y_coff;
  a.N = N; a.H = H; a.W = W; a.K_pad = K_pad; a.Ncols = Ncols; a.Ncols_pad = hpri_cdiv(Ncols, 256) * 256;
  a.y_cw = y_cw < 1 ? ((epi == HPRI_E_D2S) ? Cup : Ncols) : y_cw; a.accumulate = accumulate;
  a.amode = amode; a.epi = epi; a.H2 = H2; a.W2 = W2; a.py0 = py0; a.px0 = px0; a.Cup = Cup > 0 ? Cup : 1;
  a.x_bytes = (x_floats - x_coff) * 4;
  const long long P = (long long)N * H * W;
  const long long tiles = (P + G1_M - 1) / G1_M;
  HPRI_REQUIRE(P < (1ll << 31) - 256 && tiles * (a.Ncols_pad >> 8) < (1ll << 31), "gemm1x1: too many pixels");
  dim3 grid((unsigned)(tiles * (a.Ncols_pad >> 8)), 1u, 1u);
  hipLaunchKernelGGL(gemm1x1_kernel, grid, dim3(256), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}
