// Winograd F(2x2,3x3) weight gradient, second form (the first one: conv_wino.hip, conv_wino_wgrad_kernel).  Same arithmetic,
//     dU[xi][cin][cout] = sum over tiles  V[xi][tile][cin] * dM[xi][tile][cout],   V = B^T x B,  dM = A dY A^T,
// same slabs ws[split][xi][Cr][Nr] and the same fixed-order reduce (G^T dU G into OIHW), but organised like conv_wino4.hip:
//   workgroup  256 threads = 4 waves, 32 input channels x 64 output channels x 16 frequencies, 70 KB of LDS: TWO per CU, with
//              independent barriers (the 8-wave form runs its 2 waves per SIMD in lock-step behind one barrier per unit)
//   wave a     owns frequency ROW a: all four column frequencies, 4 x 2 accumulator tiles (128 VGPRs); the row transform of
//              x (P[c] = d[r1][c] + sigma d[r2][c]) and of dY (0 or 1 add per column, compiled per row) is done ONCE for the four
//              column frequencies: 2.0 vector instructions per MFMA instead of 2.0-2.75, 12-16 LDS reads per 8 MFMAs instead of 16-20
//   unit       one strip of 16 tiles (2 output rows x 32 columns): x halo 4 x 36 pixels x 32 cin and dY 2 x 32 pixels x 64 cout
//              as [pixel][channel] by buffer_load ... lds, double-buffered; one barrier per unit of 64 MFMAs per wave
#include "common.h"

struct WinoWgrad4Args {
  const float* x; int x_cs, x_coff, x_cvalid;
  const float* dy; int dy_cs, dy_coff, dy_cvalid;
  float* ws;                    // [splits][16][Cr][Nr]
  int N, H, W, Cr, Nr, cblk;    // cblk = Cr / 32
  int strips_x, strips_y, total, per_split;
};

#define W4G_XROW 36                                  // staged halo pixels per row (34 used)
#define W4G_XPIX (4 * W4G_XROW)                      // 144 pixels x 128 B
#define W4G_XI 18                                    // x DMA pieces (8 pixels x 128 B each)
#define W4G_YI 16                                    // dY DMA pieces (4 pixels x 256 B each)
#define W4G_X_BYTES (W4G_XI * 1024)
#define W4G_Y_BYTES (W4G_YI * 1024)
#define W4G_STAGE (W4G_X_BYTES + W4G_Y_BYTES)

__global__ __launch_bounds__(256, 2) void conv_wino_wgrad4_kernel(WinoWgrad4Args a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * W4G_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int fa = wave;
  const int split = blockIdx.x, cb = blockIdx.y % a.cblk, nbk = blockIdx.y / a.cblk;
  const int c_blk = cb * 32, n_blk = nbk * 64;
  const int u0 = split * a.per_split, u1 = min(a.total, u0 + a.per_split);

  // row roles as the forward kernel: B^T row fa = s1 (d[r1] + sigma d[r2])
  const int r1 = (fa == 0) ? 0 : 1, r2 = (fa == 3) ? 3 : 2;
  const float s1 = (fa == 2) ? -1.f : 1.f, s2 = (fa == 1 || fa == 2) ? 1.f : -1.f;
  const float sigma = s1 * s2;

  f32x16 acc[4][2];                                 // [column frequency b][cout tile nt]
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][nt][r] = 0.f;

  // ---- DMA roles: x piece i (18): pixels [8i, 8i+8) of the 4 x 36 halo, lane -> pixel 8i + (lane>>3), channel quad lane&7;
  //      dY piece i (16): pixels [4i, 4i+4) of the 2 x 32 strip, lane -> pixel 4i + (lane>>4), channel quad lane&15.
  //      Offsets are relative to the unit's first halo pixel (y0-1, x0-1) resp. first strip pixel (y0, x0). ----
  constexpr int NX = (W4G_XI + 3) / 4, NY = W4G_YI / 4;
  unsigned xvo[NX]; int xrow[NX], xpx[NX];
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const int p = (j * 4 + wave) * 8 + (lane >> 3);
    const int row = p / W4G_XROW, px = p - row * W4G_XROW;
    const bool ok = (j * 4 + wave) < W4G_XI && px < 34 && c_blk + (lane & 7) * 4 < a.x_cvalid;
    xvo[j] = (unsigned)((row * a.W + px) * a.x_cs * 4 + (lane & 7) * 16);
    xrow[j] = ok ? row : (1 << 24);                 // never inside the image: zero-filled by the range check
    xpx[j] = px;
  }
  unsigned yvo[NY]; int yrow[NY], ypx[NY];
#pragma unroll
  for (int j = 0; j < NY; ++j) {
    const int p = (j * 4 + wave) * 4 + (lane >> 4);
    const int row = p >> 5, px = p & 31;
    const bool ok = n_blk + (lane & 15) * 4 < a.dy_cvalid;
    yvo[j] = (unsigned)((row * a.W + px) * a.dy_cs * 4 + (lane & 15) * 16);
    yrow[j] = ok ? row : (1 << 24);
    ypx[j] = px;
  }
#define LOAD_UNIT(u_, buf_)                                                                                            \
  {                                                                                                                    \
    int q_ = (u_);                                                                                                     \
    const int sx_ = q_ % a.strips_x; q_ /= a.strips_x;                                                                 \
    const int sy_ = q_ % a.strips_y; const int img_ = q_ / a.strips_y;                                                 \
    const int y0_ = sy_ * 2, x0_ = sx_ * 32;                                                                           \
    unsigned char* lb_ = smem + (buf_) * W4G_STAGE;                                                                    \
    const hpri_rsrc_t rx_ = HPRI_MAKE_RSRC(a.x + ((long long)(img_ * a.H + y0_ - 1) * a.W + x0_ - 1) * a.x_cs + a.x_coff + c_blk, 0x7FFFFF00); \
    const hpri_rsrc_t ry_ = HPRI_MAKE_RSRC(a.dy + ((long long)(img_ * a.H + y0_) * a.W + x0_) * a.dy_cs + a.dy_coff + n_blk, 0x7FFFFF00);     \
    _Pragma("unroll") for (int j = 0; j < NX; ++j) {                                                                   \
      const int i_ = j * 4 + wave;                                                                                     \
      if (i_ < W4G_XI) {                                                                                               \
        const bool in_ = (unsigned)(y0_ - 1 + xrow[j]) < (unsigned)a.H && (unsigned)(x0_ - 1 + xpx[j]) < (unsigned)a.W; \
        HPRI_LDS_DMA16(rx_, lb_ + i_ * 1024, in_ ? xvo[j] : HPRI_DMA_OOB, 0);                                          \
      }                                                                                                                \
    }                                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < NY; ++j) {                                                                   \
      const int i_ = j * 4 + wave;                                                                                     \
      const bool in_ = (y0_ + yrow[j]) < a.H && (x0_ + ypx[j]) < a.W;                                                  \
      HPRI_LDS_DMA16(ry_, lb_ + W4G_X_BYTES + i_ * 1024, in_ ? yvo[j] : HPRI_DMA_OOB, 0);                              \
    }                                                                                                                  \
  }

  // dM row part per frequency row (A = [1 0; 1 1; 1 -1; 0 -1]): a = 0: dY[0] | 1: dY[0] + dY[1] | 2: dY[0] - dY[1] | 3: -dY[1];
  // column part b = 0: t0 | 1: t0 + t1 | 2: t0 - t1 | 3: -t1.  The minus signs of row 3 and column 3 are applied at the slab write.
#define K_LOOP(FA_)                                                                                                    \
    _Pragma("unroll 2") for (int kk = 0; kk < 8; ++kk) {      /* MFMA k-step: tiles 2 kk + lh of the strip */            \
      const int tile = 2 * kk + lh;                                                                                    \
      float P[4];                                                                                                      \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                  \
        const float d1 = xs[(r1 * W4G_XROW + 2 * tile + c) * 32 + li];                                                 \
        const float d2 = xs[(r2 * W4G_XROW + 2 * tile + c) * 32 + li];                                                 \
        P[c] = d1 + sigma * d2;                                                                                        \
      }                                                                                                                \
      const float va[4] = {P[0] - P[2], P[1] + P[2], P[2] - P[1], P[1] - P[3]};                                        \
      float vb[4][2];                                                                                                  \
      _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                               \
        float t0, t1;                                                                                                  \
        if (FA_ == 0 || FA_ == 3) {                                                                                    \
          t0 = ys[(((FA_ == 3) ? 1 : 0) * 32 + 2 * tile + 0) * 64 + nt * 32 + li];                                     \
          t1 = ys[(((FA_ == 3) ? 1 : 0) * 32 + 2 * tile + 1) * 64 + nt * 32 + li];                                     \
        } else {                                                                                                       \
          const float y00 = ys[(0 * 32 + 2 * tile + 0) * 64 + nt * 32 + li], y01 = ys[(0 * 32 + 2 * tile + 1) * 64 + nt * 32 + li]; \
          const float y10 = ys[(1 * 32 + 2 * tile + 0) * 64 + nt * 32 + li], y11 = ys[(1 * 32 + 2 * tile + 1) * 64 + nt * 32 + li]; \
          t0 = (FA_ == 1) ? (y00 + y10) : (y00 - y10);                                                                 \
          t1 = (FA_ == 1) ? (y01 + y11) : (y01 - y11);                                                                 \
        }                                                                                                              \
        vb[0][nt] = t0; vb[1][nt] = t0 + t1; vb[2][nt] = t0 - t1; vb[3][nt] = t1;                                      \
      }                                                                                                                \
      _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                                    \
          _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                             \
              acc[b][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[b], vb[b][nt], acc[b][nt], 0, 0, 0);                \
    }
#define UNIT_LOOP(FA_)                                                                                                 \
  for (int u = u0; u < u1; ++u) {                                                                                      \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                   \
    __builtin_amdgcn_s_barrier();          /* this unit has landed for all four waves; the other buffer is free */     \
    if (u + 1 < u1) LOAD_UNIT(u + 1, (u + 1 - u0) & 1)                                                                 \
    const float* xs = reinterpret_cast<const float*>(smem + ((u - u0) & 1) * W4G_STAGE);                               \
    const float* ys = xs + W4G_X_BYTES / 4;                                                                            \
    K_LOOP(FA_)                                                                                                        \
  }
  if (u0 < u1) LOAD_UNIT(u0, 0)
  switch (wave) {
    case 0: UNIT_LOOP(0) break;
    case 1: UNIT_LOOP(1) break;
    case 2: UNIT_LOOP(2) break;
    default: UNIT_LOOP(3) break;
  }
#undef UNIT_LOOP
#undef K_LOOP
#undef LOAD_UNIT
  // slab: ws[split][xi][c][n]; accumulator rows = cin (register index), columns = cout (lane)
  float* slab = a.ws + (size_t)split * 16 * a.Cr * a.Nr;
  const float sdy = (fa == 3) ? -s1 : s1;                               // input-transform row sign x dY row sign
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int xi = fa * 4 + b;
    const float sg = (b == 3) ? -sdy : sdy;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = n_blk + nt * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c_blk + (r & 3) + 8 * (r >> 2) + 4 * lh;
        slab[((size_t)xi * a.Cr + c) * a.Nr + n] = sg * acc[b][nt][r];
      }
    }
  }
}

static void wgrad4_geometry(int N, int H, int W, int Cin_pad, int Cout_pad, WinoWgrad4Args* a, int* splits) {
  a->cblk = hpri_cdiv(Cin_pad, 32); a->Cr = a->cblk * 32; a->Nr = hpri_cdiv(Cout_pad, 64) * 64;
  a->strips_x = hpri_cdiv(W, 32); a->strips_y = hpri_cdiv(H, 2); a->total = N * a->strips_x * a->strips_y;
  const int tiles = a->cblk * (a->Nr / 64);
  // two workgroups per CU: splits such that tiles * splits is close to a multiple of 512, with >= 8 strips per split
  int best = 1; double best_eff = 0.0;
  for (int k = 1; k <= 1024; ++k) {
    if (k > 1 && a->total / k < 8) break;
    const double per = (double)tiles * k / 512.0;
    double eff = per / (double)((long long)(per + 0.999999));
    if (per < 1.0) eff = per;
    if (eff > best_eff + 1e-9) { best_eff = eff; best = k; }
  }
  *splits = best;
  a->per_split = hpri_cdiv(a->total, best);
}

// Workspace of hpri_conv_wino_wgrad4: splits * 16 * Cr * Nr floats.
extern "C" int hpri_wino_wgrad4_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int* splits, int* Cr, int* Nr) {
  WinoWgrad4Args a;
  wgrad4_geometry(N, H, W, Cin_pad, Cout_pad, &a, splits);
  *Cr = a.Cr; *Nr = a.Nr;
  return HPRI_OK;
}

extern "C" int hpri_conv_wino_wgrad4(const float* x, int x_cs, int x_coff, int x_cvalid, const float* dy, int dy_cs, int dy_coff,
                                     int dy_cvalid, float* ws, size_t ws_floats, int N, int H, int W, int Cin_pad, int Cout_pad,
                                     hipStream_t stream) {
  HPRI_REQUIRE(x && dy && ws, "conv_wino_wgrad4: null pointer");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0 && Cin_pad > 0 && Cout_pad > 0, "conv_wino_wgrad4: bad sizes");
  HPRI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0 && dy_cs % 4 == 0 && dy_coff % 4 == 0 && x_cvalid % 4 == 0 && dy_cvalid % 4 == 0,
               "conv_wino_wgrad4: channel strides / offsets / valid counts must be multiples of 4");
  HPRI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "conv_wino_wgrad4: pointers must be 16-byte aligned");
  HPRI_REQUIRE((long long)5 * W * x_cs * 4 < (1ll << 31) && (long long)3 * W * dy_cs * 4 < (1ll << 31),
               "conv_wino_wgrad4: image rows too long for 32-bit buffer offsets");
  WinoWgrad4Args a;
  int splits;
  wgrad4_geometry(N, H, W, Cin_pad, Cout_pad, &a, &splits);
  if ((size_t)splits * 16 * a.Cr * a.Nr > ws_floats) return hpri_set_error(HPRI_ERR_WORKSPACE, "conv_wino_wgrad4: workspace too small");
  a.x = x; a.x_cs = x_cs; a.x_coff = x_coff; a.x_cvalid = x_cvalid; a.dy = dy; a.dy_cs = dy_cs; a.dy_coff = dy_coff; a.dy_cvalid = dy_cvalid;
  a.ws = ws; a.N = N; a.H = H; a.W = W;
  dim3 grid((unsigned)splits, (unsigned)(a.cblk * (a.Nr / 64)), 1u);
  hipLaunchKernelGGL(conv_wino_wgrad4_kernel, grid, dim3(256), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}
