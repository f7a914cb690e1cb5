// Winograd F(2x2, 3x3) convolution in exact fp32 on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32): forward and data
// gradient of the 3x3 / pad-1 / stride-1 convolutions of DoubleConv (reference model_parts.py:22,25; models.py:169,177).
//
// The direct implicit GEMM spends 36 multiplies per 2x2 outputs and channel pair; Winograd's minimal filtering spends 16:
//     V = B^T d B   (4x4 input tile d, per input channel)         B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
//     U = G g G^T   (3x3 filter g, per channel pair; packed once) G   = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
//     M[xi] = sum_cin V[xi] * U[xi]   for the 16 frequencies xi   (16 GEMMs: [tiles x Cin] x [Cin x Cout])
//     Y = A^T M A   (2x2 outputs)                                 A^T = [1 1 1 0; 0 1 -1 -1]
// i.e. 2.25x fewer MFMAs in the same arithmetic type (cuDNN runs the reference's fp32 convolutions the same way);
// results differ from the direct sum by fp32 rounding of the transforms only (measured: profiles/r02_precision_error.json).
//
//   workgroup  512 threads = 8 waves, one per CU (two waves per SIMD); output tile 16x16 pixels = 64 Winograd tiles x 64
//              output channels.  Wave w owns frequencies (a, 2b) and (a, 2b+1) with a = w>>1, b = w&1 for all 64 tiles:
//              2 frequencies x 2 tile groups x 2 channel groups of 32x32 accumulators = 128 VGPRs.
//   A (input)  18x18 halo x 32 channels per chunk, fp32, 128-byte pixel rows in a 20-wide grid, filled by LDS-DMA
//              (quads XOR-swizzled with (pixel>>1)&7: the stride-2 tile reads are 2-way conflicted at worst), double
//              buffered; pixels outside the image come from a page of zeros.  The input transform happens in registers on
//              the way to the MFMA: 6 ds_read_b128 + 5 packed adds per tile group and 8-channel stage.
//   B (U)      per 8-channel stage each wave DMA-loads the 2 x [8 k][64 n] panels of ITS frequencies into a private LDS
//              area (double buffered): no workgroup barrier is needed for the weights at all.
//   sync       one barrier per 32-channel chunk (128 MFMAs per wave); vmcnt(0) per stage (the DMA has had a whole stage)
//   epilogue   the 16 frequency planes meet in LDS ([xi][32 tiles][32 ch], four passes), each thread applies A^T . A
//              to two (tile, channel) pairs per pass, adds the bias, stores 2x2 pixels (128-byte runs per pixel) and
//              keeps its 32 outputs for the two-pass BatchNorm partial statistics of the tile.
#include "common.h"

// The first form of the fused forward kernel (8 waves, 16 x 16 pixels, one workgroup per CU; superseded by conv_wino4.hip in round 2)
// is compiled into the DIAGNOSTICS build only (HPRI_DIAG=1 python -m hyperpri_amd.build: -DHPRI_DIAG_KERNELS; include/hyperpri_hip_diag.h);
// the weight-gradient kernels below are the product's.
#ifdef HPRI_DIAG_KERNELS
__device__ __attribute__((aligned(64))) float hpri_wino_zero[16];

struct WinoArgs {
  const float* x; int x_cs, x_coff;
  const float* up;              // packed U: [Cin_pad/8][16][8][Cout_pad]
  const float* bias;
  float* y; int y_cs, y_coff;
  float4* stats;                // [N*tiles_img][Cout_pad] (mean, M2, count, 0) or nullptr
  int N, H, W, Cin_pad, Cout, Cout_pad, y_cw, accumulate, relu;
  int tiles_x, tiles_y;
#ifdef HPRI_STAMPS
  unsigned long long* stamps;   // diagnostic builds only (tools/build_wino_diag.sh)
#endif
};

#ifdef HPRI_STAMPS
#define STAMP(i_)                                                                                          \
  {                                                                                                        \
    unsigned long long t_;                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                            \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if (a.stamps != nullptr && (threadIdx.x & 255) == 0)                                                   \
      a.stamps[((size_t)blockIdx.x * 2 + (threadIdx.x >> 8)) * 8 + (i_)] = t_;                              \
  }
#else
#define STAMP(i_)
#endif

#define WINO_HW 20               // halo grid width (18 used)
#define WINO_SLOTS (18 * WINO_HW)
#define WINO_A_BYTES (WINO_SLOTS * 128)
#define WINO_B_WAVE 4096         // bytes per wave per stage: 2 frequencies x [8][64] fp32
#define WINO_B_BYTES (8 * WINO_B_WAVE)

__global__ __launch_bounds__(512, 2) void conv_wino_kernel(WinoArgs a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * WINO_A_BYTES + 2 * WINO_B_BYTES];
  unsigned char* a_lds = smem;
  unsigned char* b_lds = smem + 2 * WINO_A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int fa = wave >> 1, fb = wave & 1;          // frequency row a, column pair b: xi = (fa, 2 fb + e), e = 0, 1
  const int tiles_img = a.tiles_x * a.tiles_y;
  const int nbc = a.Cout_pad >> 6;                  // channel blocks of one pixel tile are adjacent in launch order: the
  const int bx = blockIdx.x / nbc, nb = blockIdx.x - bx * nbc;   // input halo is fetched from HBM once and re-read from L2
  const int img = bx / tiles_img, tin = bx - img * tiles_img;
  const int ty0 = tin / a.tiles_x, tx0 = tin - ty0 * a.tiles_x;
  const int Y0 = ty0 * 16, X0 = tx0 * 16;

  // B^T row fa: two input rows r1, r2 with signs; columns needed for b pair fb: c0..c0+2 (R0,R1,R2 or R1,R2,R3)
  //   a=0: d0 - d2 | a=1: d1 + d2 | a=2: d2 - d1 | a=3: d1 - d3
  const int r1 = (fa == 0) ? 0 : 1, r2 = (fa == 3) ? 3 : 2;
  const float s1 = (fa == 2) ? -1.f : 1.f, s2 = (fa == 1 || fa == 2) ? 1.f : -1.f;
  const int c0 = fb;                                // fb = 0: columns 0,1,2 -> V[.][0] = R0 - R2, V[.][1] = R1 + R2
                                                    // fb = 1: columns 1,2,3 -> V[.][2] = R2 - R1, V[.][3] = R1 - R3
  // ---- A halo DMA: instruction i covers slots [8i, 8i+8); lane -> slot 8i + (lane>>3), physical quad lane&7 ----
  const float* ximg = a.x + (size_t)img * a.H * a.W * a.x_cs + a.x_coff;
  constexpr int NIA = (WINO_SLOTS / 8 + 7) / 8;     // per wave (the last round is partial)
  int aoff[NIA];
#pragma unroll
  for (int q = 0; q < NIA; ++q) {
    const int slot = (q * 8 + wave) * 8 + (lane >> 3);
    int off = -1;
    if (slot < WINO_SLOTS) {
      const int hy = slot / WINO_HW, hx = slot - hy * WINO_HW;
      const int iy = Y0 + hy - 1, ix = X0 + hx - 1;
      if (hx < 18 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
        off = (iy * a.W + ix) * a.x_cs + ((((lane & 7) ^ ((slot >> 1) & 7))) << 2);
    }
    aoff[q] = off;
  }
  const int lq = (lane & 7);                        // physical quad this lane fills; logical quad = lq ^ swz(slot)
  const float* zpage = hpri_wino_zero + (lane & 3) * 4;
#define LOAD_A(chunk_)                                                                                                \
  {                                                                                                                   \
    unsigned char* la_ = a_lds + ((chunk_) & 1) * WINO_A_BYTES;                                                       \
    _Pragma("unroll") for (int q = 0; q < NIA; ++q) {                                                                 \
      const int inst_ = q * 8 + wave;                                                                                 \
      if (inst_ * 8 < WINO_SLOTS) {                                                                                   \
        const int slot_ = inst_ * 8 + (lane >> 3);                                                                    \
        const int lquad_ = lq ^ ((slot_ >> 1) & 7);                                                                   \
        const bool ok_ = aoff[q] >= 0 && ((chunk_) * 32 + lquad_ * 4) < a.Cin_pad;                                    \
        const float* src_ = ok_ ? (ximg + (size_t)(unsigned)aoff[q] + (chunk_) * 32) : zpage;                         \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,                         \
                                         (__attribute__((address_space(3))) void*)(la_ + inst_ * 1024), 16, 0, 0);    \
      }                                                                                                               \
    }                                                                                                                 \
  }
  // ---- B DMA: stage s (8 channels): this wave's frequencies e = 0,1, halves h = 0,1 of the 8 k rows ----
  int goff[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int e = p >> 1, h = p & 1;
    const int xi = fa * 4 + 2 * fb + e;
    goff[p] = ((xi * 8 + 4 * h + (lane >> 4)) * a.Cout_pad) + nb * 64 + (lane & 15) * 4;
  }
  unsigned char* bw = b_lds + wave * WINO_B_WAVE;
#define LOAD_B(s_)                                                                                                    \
  {                                                                                                                   \
    const float* pb_ = a.up + (size_t)(s_) * 16 * 8 * a.Cout_pad;                                                     \
    unsigned char* lb_ = bw + ((s_) & 1) * WINO_B_BYTES;                                                              \
    _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                                     \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb_ + goff[p]),              \
                                         (__attribute__((address_space(3))) void*)(lb_ + p * 1024), 16, 0, 0);        \
  }

  // lane's tile in tile group mt: (ty, tx) = (mt*4 + (li>>3), li&7); halo slot of its input pixel (r, c):
  //   hp = (2 ty + r) * 20 + 2 tx + c
  int hpb[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) hpb[mt] = (2 * (mt * 4 + (li >> 3))) * WINO_HW + 2 * (li & 7);

  f32x16 acc[2][2][2];                              // [frequency e][tile group mt][channel group nt]
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[e][mt][nt][r] = 0.f;

  const int nstages = a.Cin_pad >> 3;
  const int nchunks = (a.Cin_pad + 31) >> 5;
  STAMP(0)
  LOAD_A(0)
  LOAD_B(0)
  // The two waves of a SIMD run the same program; to keep them from issuing their DMA pieces (each costs the issuing wave
  // 60-150 cycles without an MFMA) at the same moment, waves 0-3 issue the next stage's loads at the start of a stage and
  // waves 4-7 in its middle, between the two tile groups: one wave's issue stall is covered by its partner's MFMAs.
  // Issue order inside a stage is weights first, then (first stage of a chunk) the next chunk's halo, so the counted wait
  // at the end of that stage retires the weights and leaves the halo in flight for up to four stages.
  //
  // Vector-ALU diet (the VALU shares the SIMD's issue port with the MFMAs; tools/wino_stamps.py ablations):
  //   * halo addresses: h*128 + ((quad ^ swz(h)) << 4) == (h*128 + (swz(h) << 4)) ^ (quad << 4): twelve per-lane constants,
  //     one XOR + one ADD (buffer select) per read
  //   * transform: with sigma = s1*s2, P[c] = d1[c] + sigma d2[c] (one FMA), V'0 / V'1 = one add/sub each, and the common sign
  //     s1 (-1 only for frequency row 2) is applied once to the accumulators in the epilogue
  //   * the column-pair variant (fb) is a compile-time parameter of the loop body: no per-element selects
  //   * the 16 weight values of the NEXT stage are read right after the wait that retires their DMA, behind the last MFMAs
  const bool late = wave >= 4;
  const float sigma = s1 * s2;
  int pre[2][6];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h1 = hpb[mt] + r1 * WINO_HW + c0 + c, h2 = hpb[mt] + r2 * WINO_HW + c0 + c;
      pre[mt][2 * c + 0] = h1 * 128 + (((h1 >> 1) & 7) << 4);
      pre[mt][2 * c + 1] = h2 * 128 + (((h2 >> 1) & 7) << 4);
    }
#define WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#if defined(WINO_DIAG) && (WINO_DIAG & 4)
#define ISSUE_LOADS() {}
#else
#define ISSUE_LOADS()                                                   \
  {                                                                     \
    if (s + 1 < nstages) { LOAD_B(s + 1) }                              \
    if (g == 0 && chunk + 1 < nchunks) { LOAD_A(chunk + 1) }            \
  }
#endif
#define READ_BFR(s_)                                                                                                   \
  {                                                                                                                    \
    const float* bp_ = reinterpret_cast<const float*>(bw + ((s_) & 1) * WINO_B_BYTES);                                 \
    _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                                      \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                               \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) bfr[e][nt][j] = bp_[(e * 8 + lh * 4 + j) * 64 + nt * 32 + li]; \
  }
#define TILE_GROUP(mt, FB_)                                                                                            \
  {                                                                                                                    \
    f32x4 P[3];                                                                                                        \
    _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                                                    \
      const f32x4 d1 = *reinterpret_cast<const f32x4*>(smem + ((pre[mt][2 * c + 0] ^ q16) + aboff));                   \
      const f32x4 d2 = *reinterpret_cast<const f32x4*>(smem + ((pre[mt][2 * c + 1] ^ q16) + aboff));                   \
      P[c] = d1 + sigma * d2;                                                                                          \
    }                                                                                                                  \
    /* FB_ = 0: V0 = P0 - P2, V1 = P1 + P2   |   FB_ = 1 (P = columns 1,2,3): V2 = P2 - P1, V3 = P1 - P3 */            \
    const f32x4 v0 = FB_ ? (P[1] - P[0]) : (P[0] - P[2]);                                                              \
    const f32x4 v1 = FB_ ? (P[0] - P[2]) : (P[1] + P[2]);                                                              \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                      \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                             \
          acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[j], bfr[0][nt][j], acc[0][mt][nt], 0, 0, 0);        \
          acc[1][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[j], bfr[1][nt][j], acc[1][mt][nt], 0, 0, 0);        \
        }                                                                                                              \
  }
#define MAIN_LOOP(FB_)                                                                                                 \
  for (int chunk = 0; chunk < nchunks; ++chunk) {                                                                      \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                   \
    __builtin_amdgcn_s_barrier();          /* chunk's halo is visible; everyone has left the previous chunk */         \
    const int aboff = (chunk & 1) * WINO_A_BYTES;                                                                      \
    const int sg = min(4, nstages - chunk * 4);                                                                        \
    if (chunk == 0) { READ_BFR(0) }                                                                                    \
    for (int g = 0; g < sg; ++g) {                                                                                     \
      const int s = chunk * 4 + g;                                                                                     \
      if (!late) ISSUE_LOADS()                                                                                         \
      const int q16 = (2 * g + lh) << 4;    /* this lane half's channel quad inside the chunk, as a byte offset */      \
      TILE_GROUP(0, FB_)                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      if (late) ISSUE_LOADS()                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      TILE_GROUP(1, FB_)                                                                                               \
      /* the next stage's weights must have landed; a halo issued in this stage (after them) may stay in flight */     \
      if (g == 0 && chunk + 1 < nchunks && s + 1 < nstages) { if (wave < 5) WAIT_VM(6); else WAIT_VM(5); }             \
      else WAIT_VM(0);                                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      if (s + 1 < nstages) { READ_BFR(s + 1) }                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
    }                                                                                                                  \
  }
  float bfr[2][2][4];                               // weights of the stage: bfr[e][nt][j] = U[xi_e][k = 4 lh + j][n = nt*32 + li]
  if (fb) { MAIN_LOOP(1) } else { MAIN_LOOP(0) }
#undef MAIN_LOOP
#undef READ_BFR
#undef TILE_GROUP
#undef ISSUE_LOADS
#undef WAIT_VM
#undef LOAD_A
#undef LOAD_B
  STAMP(1)
  __syncthreads();

  // ------------------------------- epilogue -------------------------------
  // acc[e][mt][nt][r]: tile t = (r&3) + 8*(r>>2) + 4*lh of group mt, channel nt*32 + li, frequency (fa, 2 fb + e).
  // Two passes (tile groups): the 16 frequency planes of 32 tiles x 64 channels meet in LDS (128 KB), then every thread
  // owns one (tile, channel quad): 16 float4 reads, A^T . A, bias, and 2x2 pixels stored as float4.
  float* ex = reinterpret_cast<float*>(smem);       // [16 xi][32 tiles][64 ch]
  static_assert(16 * 32 * 64 * 4 <= 2 * WINO_A_BYTES + 2 * WINO_B_BYTES, "exchange buffer must fit the staging LDS");
  const int ot = tid >> 4, oq = tid & 15;           // output phase: tile of the group, channel quad
  const int n0 = nb * 64 + oq * 4;
  f32x4 outv[2][4];                                 // [mt][pixel] x 4 channels
  const int vrows = min(16, a.H - Y0), vcols = min(16, a.W - X0);
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias != nullptr) {
#pragma unroll
    for (int c = 0; c < 4; ++c) if (n0 + c < a.Cout) bias4[c] = a.bias[n0 + c];
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int xi = fa * 4 + 2 * fb + e;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = (r & 3) + 8 * (r >> 2) + 4 * lh;
          ex[(xi * 32 + t) * 64 + nt * 32 + li] = s1 * acc[e][mt][nt][r];     // s1: the transform sign left out of the loop
        }
    }
    if (mt == 0) { STAMP(4) }
    __syncthreads();
    if (mt == 0) { STAMP(5) }
    f32x4 m[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) m[xi] = *reinterpret_cast<const f32x4*>(ex + (xi * 32 + ot) * 64 + oq * 4);
    // Y = A^T M A with A^T = [1 1 1 0; 0 1 -1 -1]
    const f32x4 u0 = m[0] + m[4] + m[8], u1 = m[1] + m[5] + m[9], u2 = m[2] + m[6] + m[10], u3 = m[3] + m[7] + m[11];
    const f32x4 w0 = m[4] - m[8] - m[12], w1 = m[5] - m[9] - m[13], w2 = m[6] - m[10] - m[14], w3 = m[7] - m[11] - m[15];
    f32x4 o[4] = {u0 + u1 + u2, u1 - u2 - u3, w0 + w1 + w2, w1 - w2 - w3};
    const int py = 2 * (mt * 4 + (ot >> 3)), px = 2 * (ot & 7);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4 v = o[k] + bias4;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (a.relu) v[c] = fmaxf(v[c], 0.f);
        if (n0 + c >= a.Cout) v[c] = 0.f;
      }
      const int yy = py + (k >> 1), xx = px + (k & 1);
      const bool ok = yy < vrows && xx < vcols;
      if (ok && n0 < a.y_cw) {
        float* dst = a.y + ((size_t)(img * a.H + Y0 + yy) * a.W + X0 + xx) * a.y_cs + a.y_coff + n0;
        if (a.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = v;
      }
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      outv[mt][k] = ok ? v : zero4;
    }
    if (mt == 0) { STAMP(6) }
    __syncthreads();
    if (mt == 0) { STAMP(7) }
  }

  STAMP(2)
  if (a.stats != nullptr) {
    // two-pass per-tile statistics over the valid pixels: thread (ot, oq) holds 8 pixels of channels 4 oq .. 4 oq + 3
    float* red = reinterpret_cast<float*>(smem);    // [32 ot][64 ch]
    const float cnt = (float)(vrows * vcols);
    f32x4 mean4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int py = 2 * (mt * 4 + (ot >> 3)), px = 2 * (ot & 7);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool ok = (py + (k >> 1)) < vrows && (px + (k & 1)) < vcols;
          if (pass == 0) sacc += outv[mt][k];
          else if (ok) { const f32x4 d = outv[mt][k] - mean4; sacc += d * d; }
        }
      }
      *reinterpret_cast<f32x4*>(red + ot * 64 + oq * 4) = sacc;
      __syncthreads();
      // channel c = tid (64 threads): sum over the 32 tiles
      if (tid < 64) {
        float tsum = 0.f;
#pragma unroll 8
        for (int q = 0; q < 32; ++q) tsum += red[q * 64 + tid];
        if (pass == 0) red[32 * 64 + tid] = tsum / cnt;
        else a.stats[(size_t)bx * a.Cout_pad + nb * 64 + tid] = make_float4(red[32 * 64 + tid], tsum, cnt, 0.f);
      }
      __syncthreads();
      if (pass == 0) mean4 = *reinterpret_cast<const f32x4*>(red + 32 * 64 + oq * 4);
      __syncthreads();
    }
  }
  STAMP(3)
}

// ---- filter transform: U = G g G^T, packed [Cin_pad/8][16][8][Ncols_pad] ------------------------------------------------
// mode 0: forward       g(k = c, col = n)[t] = W[n][c][t]            (W: [Cout][Cin][3][3], src_d1 = Cin)
// mode 1: data gradient g(k = n, col = c)[t] = W[n][c][8 - t]        (K = Cout, columns = Cin)
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ up, int mode, int K, int Ncols, int Ncols_pad,
                                 int stages, int src_d1, const float* __restrict__ colscale) {
  const size_t total = (size_t)stages * 8 * Ncols_pad;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(idx % Ncols_pad);
    const int kk = (int)((idx / Ncols_pad) % 8);
    const int st = (int)(idx / ((size_t)Ncols_pad * 8));
    const int k = st * 8 + kk;
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = 0.f;
      if (k < K && col < Ncols) {
        v = (mode == 0) ? w[((size_t)col * src_d1 + k) * 9 + t] : w[((size_t)k * src_d1 + col) * 9 + (8 - t)];
        if (colscale != nullptr) v *= colscale[col];
      }
      g[t / 3][t % 3] = v;
    }
    // Gg: 4x3
    float gg[4][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      gg[0][q] = g[0][q];
      gg[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
      gg[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
      gg[3][q] = g[2][q];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float u0 = gg[i][0], u1 = 0.5f * (gg[i][0] + gg[i][1] + gg[i][2]), u2 = 0.5f * (gg[i][0] - gg[i][1] + gg[i][2]),
                  u3 = gg[i][2];
      const float u[4] = {u0, u1, u2, u3};
#pragma unroll
      for (int j = 0; j < 4; ++j) up[(((size_t)st * 16 + i * 4 + j) * 8 + kk) * Ncols_pad + col] = u[j];
    }
  }
}

#endif   // HPRI_DIAG_KERNELS

extern "C" size_t hpri_wino_packed_floats(int K, int Ncols_pad) { return (size_t)hpri_cdiv(K, 8) * 16 * 8 * Ncols_pad; }

#ifdef HPRI_DIAG_KERNELS
extern "C" int hpri_wino_pack(const float* w, float* up, const float* colscale, int mode, int K, int Ncols, int Ncols_pad,
                              int src_d1, hipStream_t stream) {
  HPRI_REQUIRE(w && up, "wino_pack: null pointer");
  HPRI_REQUIRE((mode == 0 || mode == 1) && K > 0 && Ncols > 0 && Ncols_pad >= Ncols && Ncols_pad % 64 == 0, "wino_pack: bad arguments");
  const int stages = hpri_cdiv(K, 8);
  const size_t total = (size_t)stages * 8 * Ncols_pad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, stream, w, up, mode, K, Ncols, Ncols_pad, stages, src_d1, colscale);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// stamp buffer: diagnostic builds (-DHPRI_STAMPS) only; the product library keeps no state (re-entrant, include/hyperpri_hip.h)
#ifdef HPRI_STAMPS
static unsigned long long* hpri_wino_stamps = nullptr;
extern "C" int hpri_wino_set_stamps(unsigned long long* p) { hpri_wino_stamps = p; return HPRI_OK; }
#endif

extern "C" int hpri_conv_wino_plan(int N, int H, int W, int* stat_tiles) {
  *stat_tiles = N * hpri_cdiv(H, 16) * hpri_cdiv(W, 16);
  return HPRI_OK;
}

// 3x3 / pad 1 / stride 1 convolution, Winograd F(2x2,3x3), fp32.  x: fp32 NHWC view with channels [Cin, Cin_pad) zero
// (Cin_pad a multiple of 8); up from hpri_wino_pack; accumulate bit 0: y += result, bit 1: ReLU epilogue.
extern "C" int hpri_conv_wino(const float* x, int x_cs, int x_coff, const float* up, const float* bias, float* y, int y_cs,
                              int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw,
                              int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(x && up && y, "conv_wino: null pointer");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0, "conv_wino: empty image");
  HPRI_REQUIRE(Cin_pad > 0 && Cin_pad % 8 == 0, "conv_wino: Cin_pad must be a positive multiple of 8");
  HPRI_REQUIRE(Cout_pad % 64 == 0 && Cout <= Cout_pad && Cout > 0, "conv_wino: Cout_pad must be a multiple of 64 >= Cout");
  HPRI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0 && x_coff + Cin_pad <= x_cs, "conv_wino: input channel stride/offset");
  HPRI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)up & 15) == 0, "conv_wino: pointers must be 16-byte aligned");
  HPRI_REQUIRE((long long)H * W * x_cs < (1ll << 31), "conv_wino: one image of the input view exceeds 2^31 elements");
  HPRI_REQUIRE(!((accumulate & 1) && stats != nullptr), "conv_wino: statistics are not available together with accumulate");
  WinoArgs a;
  a.x = x; a.x_cs = x_cs; a.x_coff = x_coff; a.up = up; a.bias = bias; a.y = y; a.y_cs = y_cs; a.y_coff = y_coff;
  a.stats = reinterpret_cast<float4*>(stats);
  a.N = N; a.H = H; a.W = W; a.Cin_pad = Cin_pad; a.Cout = Cout; a.Cout_pad = Cout_pad;
  a.y_cw = y_cw < Cout ? Cout : y_cw; a.accumulate = accumulate & 1; a.relu = (accumulate >> 1) & 1;
  HPRI_REQUIRE(a.y_cw + y_coff <= y_cs, "conv_wino: output channels exceed the channel stride");
  HPRI_REQUIRE(y_cs % 4 == 0 && y_coff % 4 == 0 && a.y_cw % 4 == 0 && ((uintptr_t)y & 15) == 0,
               "conv_wino: the output view must be float4-aligned (stride, offset and written width multiples of 4)");
  a.tiles_x = hpri_cdiv(W, 16); a.tiles_y = hpri_cdiv(H, 16);
#ifdef HPRI_STAMPS
  a.stamps = hpri_wino_stamps;
#endif
  dim3 grid((unsigned)(N * a.tiles_x * a.tiles_y * (Cout_pad / 64)), 1u, 1u);
  hipLaunchKernelGGL(conv_wino_kernel, grid, dim3(512), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

#endif   // HPRI_DIAG_KERNELS

// =====================================================================================================================
// Winograd weight gradient.  With M = U (.) V and Y = A^T M A, the gradient of the transformed filter is
//     dU[xi][cin][cout] = sum over tiles  V[xi][tile][cin] * dM[xi][tile][cout],   dM = A dY A^T   (A = [1 0; 1 1; 1 -1; 0 -1])
// and dg = G^T dU G: again 16 instead of 36 multiplies per 2x2 output pixels and channel pair.  Sixteen GEMMs whose
// reduction runs over the TILES: MFMA rows = cin, columns = cout, k = tile; both operands are transformed in registers.
//
//   workgroup  8 waves, wave (a, b) owns frequencies (a, 2b), (a, 2b+1) of one 64 (cin) x 64 (cout) block:
//              2 x 2 x 2 accumulator tiles = 128 VGPRs; it walks a contiguous run of strips ("split-K" over the tiles) and
//              writes ONE partial slab ws[split][xi][cin][cout]; hpri_wino_wgrad_reduce sums the slabs in a fixed order
//              (deterministic) and applies G^T . G on the way into the OIHW gradient.
//   stage      one strip of 16 tiles (2 output rows x 32 columns): input halo 4 x 34 pixels x 64 cin and dY 2 x 32 pixels
//              x 64 cout, both [pixel][channel] by LDS-DMA (lanes index the channel: every ds_read_b32 is 32 consecutive
//              dwords), double buffered; 64 MFMAs per wave, one barrier.
struct WinoWgradArgs {
  const float* x; int x_cs, x_coff, x_cvalid;
  const float* dy; int dy_cs, dy_coff, dy_cvalid;
  float* ws;                    // [splits][16][Cr][Nr]
  int N, H, W, Cr, Nr, cblk;
  int strips_x, strips_y, total, per_split;
  int ntile, items, per_xcd;    // (c, n) tiles per split; work items; items per XCD band
};

#define WG_XROW 36               // staged halo pixels per row (34 used)
#define WG_X_BYTES (4 * WG_XROW * 256)
#define WG_Y_BYTES (2 * 32 * 256)
#define WG_STAGE_BYTES (WG_X_BYTES + WG_Y_BYTES)

__global__ __launch_bounds__(512, 2) void conv_wino_wgrad_kernel(WinoWgradArgs a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * WG_STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int fa = wave >> 1, fb = wave & 1;
  // XCD-aware order (round 3): workgroup id mod 8 labels the XCD (round-robin dispatch; speed only); XCD x owns the items
  // [x * per_xcd, (x+1) * per_xcd), an item = (pixel split, (c, n) tile) with the TILE fastest: the cblk x nblk tiles that read the
  // same pixel strips run back to back on ONE XCD and share its L2.  (Before: grid (splits, tiles) -- the tiles of a split were
  // `splits` workgroups apart in dispatch order and re-read x and dy from beyond L2: 515 MB read per launch against ~300 MB.)
  const int item = (int)(blockIdx.x & 7) * a.per_xcd + (int)(blockIdx.x >> 3);
  if (item >= a.items) return;
  const int split = item / a.ntile, tile = item - split * a.ntile;
  const int cb = tile % a.cblk, nbk = tile / a.cblk;
  const int c_blk = cb * 64, n_blk = nbk * 64;
  const int u0 = split * a.per_split, u1 = min(a.total, u0 + a.per_split);

  // input-transform roles (as the forward kernel): rows r1, r2 with signs, columns c0 .. c0+2
  const int r1 = (fa == 0) ? 0 : 1, r2 = (fa == 3) ? 3 : 2;
  const float s1 = (fa == 2) ? -1.f : 1.f, s2 = (fa == 1 || fa == 2) ? 1.f : -1.f;
  const int c0 = fb;
  const float sigma = s1 * s2;

  f32x16 acc[2][2][2];                              // [frequency e][cin tile ct][cout tile nt]
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[e][ct][nt][r] = 0.f;

  // DMA pieces of a stage: x rows 4 x 9 pieces (4 pixels x 256 B each), dy 2 x 8 pieces; piece p of this wave = p*8 + wave.
  // buffer_load ... lds: the descriptor base is the unit's first (halo) pixel -- a wave-uniform pointer that may lie before the
  // tensor for border units --, the piece's row / pixel-group offset goes in as the scalar offset, and the lane part is ONE
  // per-lane constant; lanes outside the image, beyond the valid channels or in the unused tail of a staged row get an
  // out-of-range offset, which the hardware range check turns into zeros in LDS (tools/lds_dma_oob.hip).  About 4 vector
  // instructions per piece instead of ~20 (every one of them costs fp32-MFMA issue time).
  constexpr int NPX = 4 * (WG_XROW / 4), NPY = 2 * 8, NPW = (NPX + NPY + 7) / 8;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const int l4 = lane >> 4;                          // pixel of the piece's group of four
  const unsigned xlane = (c_blk + (lane & 15) * 4 < a.x_cvalid) ? (unsigned)(l4 * a.x_cs * 4 + (lane & 15) * 16) : OOB;
  const unsigned ylane = (n_blk + (lane & 15) * 4 < a.dy_cvalid) ? (unsigned)(l4 * a.dy_cs * 4 + (lane & 15) * 16) : OOB;
#define LOAD_UNIT(u_, buf_)                                                                                            \
  {                                                                                                                    \
    int q_ = (u_);                                                                                                     \
    const int sx_ = q_ % a.strips_x; q_ /= a.strips_x;                                                                 \
    const int sy_ = q_ % a.strips_y; const int img_ = q_ / a.strips_y;                                                 \
    const int y0_ = sy_ * 2, x0_ = sx_ * 32;                                                                           \
    unsigned char* lb_ = smem + (buf_) * WG_STAGE_BYTES;                                                               \
    const float* xu_ = a.x + ((long long)(img_ * a.H + y0_ - 1) * a.W + x0_ - 1) * a.x_cs + a.x_coff + c_blk;          \
    const float* yu_ = a.dy + ((long long)(img_ * a.H + y0_) * a.W + x0_) * a.dy_cs + a.dy_coff + n_blk;               \
    const hpri_rsrc_t rx_ = HPRI_MAKE_RSRC(xu_, 0x7FFFFF00);       \
    const hpri_rsrc_t ry_ = HPRI_MAKE_RSRC(yu_, 0x7FFFFF00);       \
    _Pragma("unroll") for (int p = 0; p < NPW; ++p) {                                                                  \
      const int piece_ = p * 8 + wave;                                                                                 \
      if (piece_ < NPX) {                                                                                              \
        const int row_ = piece_ / (WG_XROW / 4), pxb_ = (piece_ % (WG_XROW / 4)) * 4;          /* wave-uniform */       \
        const bool rowok_ = (unsigned)(y0_ + row_ - 1) < (unsigned)a.H;                                                \
        const bool ok_ = rowok_ && (pxb_ + l4) < 34 && (unsigned)(x0_ + pxb_ + l4 - 1) < (unsigned)a.W;                \
        HPRI_LDS_DMA16(rx_, lb_ + piece_ * 1024, ok_ ? xlane : OOB, (row_ * a.W + pxb_) * a.x_cs * 4);           \
      } else if (piece_ < NPX + NPY) {                                                                                 \
        const int pp_ = piece_ - NPX;                                                                                  \
        const int row_ = pp_ >> 3, pxb_ = (pp_ & 7) * 4;                                                               \
        const bool ok_ = (y0_ + row_) < a.H && (x0_ + pxb_ + l4) < a.W;                                                \
        HPRI_LDS_DMA16(ry_, lb_ + piece_ * 1024, ok_ ? ylane : OOB, (row_ * a.W + pxb_) * a.dy_cs * 4);          \
      }                                                                                                                \
    }                                                                                                                  \
  }

// The loop body is compiled per (frequency row, column pair): the row part of dM = A dY A^T multiplies by constants that are
// 0 or +-1 -- a = 0: dY[0], a = 1: dY[0] + dY[1], a = 2: dY[0] - dY[1], a = 3: -dY[1] -- so rows 0 and 3 need no arithmetic at
// all and rows 1, 2 one add per column; the column part for b = 2 fb + e is t0 | t0 + t1 (fb = 0), t0 - t1 | -t1 (fb = 1).  The
// two minus signs (row 3, and b = 3) are left out here and applied once to the accumulators when the slab is written.
#define K_LOOP(FA_, FB_)                                                                                               \
    _Pragma("unroll 2") for (int kk = 0; kk < 8; ++kk) {      /* MFMA k-step: tiles 2 kk + lh of the strip */            \
      const int tile = 2 * kk + lh;                                                                                    \
      float va[2][2], vb[2][2];                     /* [frequency e][cin tile | cout tile] */                          \
      _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) {                                                               \
        float P[3];                                                                                                    \
        _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                                                \
          const float d1 = xs[(r1 * WG_XROW + 2 * tile + c0 + c) * 64 + ct * 32 + li];                                 \
          const float d2 = xs[(r2 * WG_XROW + 2 * tile + c0 + c) * 64 + ct * 32 + li];                                 \
          P[c] = d1 + sigma * d2;                                                                                      \
        }                                                                                                              \
        va[0][ct] = FB_ ? (P[1] - P[0]) : (P[0] - P[2]);                                                               \
        va[1][ct] = FB_ ? (P[0] - P[2]) : (P[1] + P[2]);                                                               \
      }                                                                                                                \
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
        vb[0][nt] = FB_ ? (t0 - t1) : t0;                                                                              \
        vb[1][nt] = FB_ ? t1 : (t0 + t1);                                                                              \
      }                                                                                                                \
      _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                                    \
          _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                             \
              _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                         \
                  acc[e][ct][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[e][ct], vb[e][nt], acc[e][ct][nt], 0, 0, 0); \
    }
  // the variant is chosen once per wave, outside the unit loop (eight copies of the loop, no selects inside)
#define UNIT_LOOP(FA_, FB_)                                                                                            \
  for (int u = u0; u < u1; ++u) {                                                                                      \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                   \
    __builtin_amdgcn_s_barrier();          /* this unit has landed for everyone; the other buffer is free */           \
    if (u + 1 < u1) LOAD_UNIT(u + 1, (u + 1 - u0) & 1)                                                                 \
    const float* xs = reinterpret_cast<const float*>(smem + ((u - u0) & 1) * WG_STAGE_BYTES);                          \
    const float* ys = xs + WG_X_BYTES / 4;                                                                             \
    K_LOOP(FA_, FB_)                                                                                                   \
  }
  if (u0 < u1) LOAD_UNIT(u0, 0)
  switch (wave) {                          // wave = 2 fa + fb
    case 0: UNIT_LOOP(0, 0) break;
    case 1: UNIT_LOOP(0, 1) break;
    case 2: UNIT_LOOP(1, 0) break;
    case 3: UNIT_LOOP(1, 1) break;
    case 4: UNIT_LOOP(2, 0) break;
    case 5: UNIT_LOOP(2, 1) break;
    case 6: UNIT_LOOP(3, 0) break;
    default: UNIT_LOOP(3, 1) break;
  }
#undef UNIT_LOOP
#undef K_LOOP
#undef LOAD_UNIT
  // slab: ws[split][xi][c][n]; accumulator rows = cin (register index), columns = cout (lane)
  float* slab = a.ws + (size_t)split * 16 * a.Cr * a.Nr;
  const float sdy = (fa == 3) ? -s1 : s1;                              // input-transform row sign x dY row sign
  const float sgn[2] = {sdy, (fb == 1) ? -sdy : sdy};                  // ... x the sign of column frequency b = 3
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int xi = fa * 4 + 2 * fb + e;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int n = n_blk + nt * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = c_blk + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          slab[((size_t)xi * a.Cr + c) * a.Nr + n] = sgn[e] * acc[e][ct][nt][r];   // the transform signs left out of the loop
        }
      }
  }
}

// dW[n][c][3][3] (+)= G^T (sum over splits of dU[.][c][n]) G, fixed summation order.  One thread per (c, n).
__global__ void wino_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits, int Cr, int Nr,
                                         int Cin, int Cout, int accumulate) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), c = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (n >= Cout || c >= Cin) return;
  float u[16];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) u[xi] = 0.f;
  const size_t slab = (size_t)16 * Cr * Nr;
  for (int k = 0; k < splits; ++k) {
    const float* p = ws + (size_t)k * slab + (size_t)c * Nr + n;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) u[xi] += p[(size_t)xi * Cr * Nr];
  }
  // t = G^T u (3x4 . 4x4), G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
  float t[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[0][j] = u[0 * 4 + j] + 0.5f * (u[1 * 4 + j] + u[2 * 4 + j]);
    t[1][j] = 0.5f * (u[1 * 4 + j] - u[2 * 4 + j]);
    t[2][j] = 0.5f * (u[1 * 4 + j] + u[2 * 4 + j]) + u[3 * 4 + j];
  }
  float* o = dw + ((size_t)n * Cin + c) * 9;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float g0 = t[i][0] + 0.5f * (t[i][1] + t[i][2]), g1 = 0.5f * (t[i][1] - t[i][2]), g2 = 0.5f * (t[i][1] + t[i][2]) + t[i][3];
    o[i * 3 + 0] = accumulate ? o[i * 3 + 0] + g0 : g0;
    o[i * 3 + 1] = accumulate ? o[i * 3 + 1] + g1 : g1;
    o[i * 3 + 2] = accumulate ? o[i * 3 + 2] + g2 : g2;
  }
}

// The same for MANY slabs (the 608 x 968 layers: 256 slabs of 16 x 64 x 64, 67 MB, which the kernel above reads with 16
// workgroups): S slab slices x 64 columns per workgroup and ONE input channel; slice s sums slabs s, s + S, ... in order, the
// slices meet in LDS in order (deterministic).
template <int S>
__global__ __launch_bounds__(64 * S) void wino_wgrad_reduce_wide_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                        int splits, int Cr, int Nr, int Cin, int Cout,
                                                                        int accumulate) {
  __shared__ float red[S - 1][16][64];
  const int ln = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + ln, c = blockIdx.y;            // n < Nr (a multiple of 64), c < Cin <= Cr
  float u[16];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) u[xi] = 0.f;
  const size_t slab = (size_t)16 * Cr * Nr;
  // (one slab = 16 independent loads in flight per thread, 16 k per workgroup; unrolled by hipcc, the S = 16 form -- 1024
  // threads, 128 registers -- spilled 50 of them)
#pragma unroll 1
  for (int k = sl; k < splits; k += S) {
    const float* p = ws + (size_t)k * slab + (size_t)c * Nr + n;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) u[xi] += p[(size_t)xi * Cr * Nr];
  }
  if (sl > 0) {
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) red[sl - 1][xi][ln] = u[xi];
  }
  __syncthreads();
  if (sl > 0 || n >= Cout) return;
#pragma unroll 2
  for (int q = 0; q < S - 1; ++q)
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) u[xi] += red[q][xi][ln];
  float t[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[0][j] = u[0 * 4 + j] + 0.5f * (u[1 * 4 + j] + u[2 * 4 + j]);
    t[1][j] = 0.5f * (u[1 * 4 + j] - u[2 * 4 + j]);
    t[2][j] = 0.5f * (u[1 * 4 + j] + u[2 * 4 + j]) + u[3 * 4 + j];
  }
  float* o = dw + ((size_t)n * Cin + c) * 9;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float g0 = t[i][0] + 0.5f * (t[i][1] + t[i][2]), g1 = 0.5f * (t[i][1] - t[i][2]), g2 = 0.5f * (t[i][1] + t[i][2]) + t[i][3];
    o[i * 3 + 0] = accumulate ? o[i * 3 + 0] + g0 : g0;
    o[i * 3 + 1] = accumulate ? o[i * 3 + 1] + g1 : g1;
    o[i * 3 + 2] = accumulate ? o[i * 3 + 2] + g2 : g2;
  }
}

extern "C" int hpri_wino_wgrad_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int* splits, int* Cr, int* Nr) {
  const int cblk = hpri_cdiv(Cin_pad, 64), nblk = hpri_cdiv(Cout_pad, 64);
  const int total = N * hpri_cdiv(H, 2) * hpri_cdiv(W, 32);
  *Cr = cblk * 64; *Nr = nblk * 64;
  const int tiles = cblk * nblk;
  // one workgroup per CU: splits such that tiles * splits is close to a multiple of 256, with >= 8 strips per split
  auto plan = [&](int cus, int kmax, double tol) {
    int best = 1; double best_eff = 0.0;
    for (int k = 1; k <= kmax; ++k) {
      if (k > 1 && total / k < 8) break;
      const double per_cu = (double)tiles * k / (double)cus;
      double eff = per_cu / (double)((long long)(per_cu + 0.999999));
      if (per_cu < 1.0) eff = per_cu;
      if (eff > best_eff + tol + 1e-9) { best_eff = eff; best = k; }
    }
    return best;
  };
  int best = plan(256, 512, 0.0);
  // option wgrad_cu_reserve (api.cpp): plan for that many fewer CUs, so that a few CUs held by another kernel do not push the last
  // workgroups into a second wave -- with at most twice the slabs of the plain plan, and a smaller split preferred unless a larger
  // one fills the CUs at least 5 % better
  const int reserve = hpri_option(5);
  if (reserve > 0) best = plan(256 - (reserve < 192 ? reserve : 192), 2 * best, 0.05);
  *splits = best;
  return HPRI_OK;
}

// Weight gradient of a 3x3 / pad 1 convolution by Winograd: slabs in ws (splits*16*Cr*Nr floats, hpri_wino_wgrad_plan),
// then hpri_wino_wgrad_reduce into dW (OIHW).
extern "C" int hpri_conv_wino_wgrad(const float* x, int x_cs, int x_coff, int x_cvalid, const float* dy, int dy_cs, int dy_coff,
                                    int dy_cvalid, float* ws, size_t ws_floats, int N, int H, int W, int Cin_pad, int Cout_pad,
                                    hipStream_t stream) {
  HPRI_REQUIRE(x && dy && ws, "conv_wino_wgrad: null pointer");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0 && Cin_pad > 0 && Cout_pad > 0, "conv_wino_wgrad: bad sizes");
  HPRI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0 && dy_cs % 4 == 0 && dy_coff % 4 == 0 && x_cvalid % 4 == 0 && dy_cvalid % 4 == 0,
               "conv_wino_wgrad: channel strides / offsets / valid counts must be multiples of 4");
  HPRI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "conv_wino_wgrad: pointers must be 16-byte aligned");
  HPRI_REQUIRE((long long)5 * W * x_cs * 4 < (1ll << 31) && (long long)3 * W * dy_cs * 4 < (1ll << 31),
               "conv_wino_wgrad: image rows too long for 32-bit buffer offsets");
  WinoWgradArgs a;
  a.x = x; a.x_cs = x_cs; a.x_coff = x_coff; a.x_cvalid = x_cvalid; a.dy = dy; a.dy_cs = dy_cs; a.dy_coff = dy_coff; a.dy_cvalid = dy_cvalid;
  a.ws = ws; a.N = N; a.H = H; a.W = W;
  int splits;
  hpri_wino_wgrad_plan(N, H, W, Cin_pad, Cout_pad, &splits, &a.Cr, &a.Nr);
  if ((size_t)splits * 16 * a.Cr * a.Nr > ws_floats) return hpri_set_error(HPRI_ERR_WORKSPACE, "conv_wino_wgrad: workspace too small");
  a.cblk = a.Cr / 64;
  a.strips_x = hpri_cdiv(W, 32); a.strips_y = hpri_cdiv(H, 2); a.total = N * a.strips_x * a.strips_y;
  a.per_split = hpri_cdiv(a.total, splits);
  a.ntile = a.cblk * (a.Nr / 64);
  a.items = splits * a.ntile; a.per_xcd = hpri_cdiv(a.items, 8);
  dim3 grid((unsigned)(a.per_xcd * 8), 1u, 1u);
  hipLaunchKernelGGL(conv_wino_wgrad_kernel, grid, dim3(512), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_wino_wgrad_reduce(const float* ws, float* dw, int N, int H, int W, int Cin, int Cin_pad, int Cout, int Cout_pad,
                                      int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(ws && dw && Cin > 0 && Cout > 0, "wino_wgrad_reduce: bad arguments");
  int splits, Cr, Nr;
  hpri_wino_wgrad_plan(N, H, W, Cin_pad, Cout_pad, &splits, &Cr, &Nr);
  if (splits >= 64) {                               // many slabs, few (c, n) pairs: spread the slabs over the workgroup too
    dim3 grid((unsigned)hpri_cdiv(Cout, 64), (unsigned)Cin);
    hipLaunchKernelGGL(wino_wgrad_reduce_wide_kernel<16>, grid, dim3(1024), 0, stream, ws, dw, splits, Cr, Nr, Cin, Cout, accumulate);
  } else if (splits >= 8) {
    dim3 grid((unsigned)hpri_cdiv(Cout, 64), (unsigned)Cin);
    hipLaunchKernelGGL(wino_wgrad_reduce_wide_kernel<8>, grid, dim3(512), 0, stream, ws, dw, splits, Cr, Nr, Cin, Cout, accumulate);
  } else {
    dim3 grid((unsigned)hpri_cdiv(Cout, 64), (unsigned)hpri_cdiv(Cin, 4));
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, grid, dim3(256), 0, stream, ws, dw, splits, Cr, Nr, Cin, Cout, accumulate);
  }
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}
