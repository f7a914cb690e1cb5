// 3x3 / pad 1 / stride 1 convolution, Winograd F(2x2,3x3) in fp32 on v_mfma_f32_32x32x2_f32 -- second form of the fused
// kernel of conv_wino.hip (reference model_parts.py:22,25; models.py:169,177; forward, and data gradient with the mode-1 pack).
//
// What changed against conv_wino.hip, and why (on this chip every vector / LDS instruction of an fp32-MFMA loop
// is paid in matrix-pipe time, and a 157 KB workgroup leaves its CU idle through its whole output transform):
//   workgroup  256 threads = 4 waves, 16 x 8 output pixels (32 tiles) x 64 channels x 16 frequencies, 78 KB of LDS:
//              TWO workgroups per CU, so one workgroup's output transform runs under the other one's MFMAs and a barrier
//              stalls four waves, not eight
//   wave a     owns frequency ROW a: all four column frequencies b, both channel groups: 4 x 2 accumulator tiles (128 VGPRs).
//              input transform: P[c] = d[r1][c] + sigma d[r2][c] (4 FMAs), V = (P0-P2, P1+P2, P2-P1, P1-P3): 32 vector
//              instructions per 32 MFMAs (was 40), 8 halo reads (was 12)
//   weights    U packed with k innermost ([Cin/8][16 freq][Cout_pad][8]): a lane's four k values are one ds_read_b128
//              (8 per stage instead of 16 dword reads); the panel of a wave is private and SINGLE-buffered: its 32 values are
//              read into registers at the start of a stage, then the next stage's DMA overwrites the panel under the MFMAs
//   epilogue   M A (column transform) is applied in registers before the LDS exchange: 8 instead of 16 planes go through
//              LDS (the ds_write_b32 rate, 64 B/clk/CU, is what the exchange costs)
//   DMA        both operands by buffer_load ... lds: 32-bit per-lane offsets against an SGPR descriptor (no 64-bit address
//              arithmetic per piece), out-of-image halo pixels zero-filled by the descriptor's range check
// Statistics and argument contract as conv_wino.hip; the halo layout is de-interleaved by column parity (below).
#include "common.h"

struct Wino4Args {
  const float* x; int x_cs, x_coff;
  const float* up;              // packed U, k innermost: [Cin_pad/8][16][Cout_pad][8]
  const float* bias;
  float* y; int y_cs, y_coff;
  float4* stats;                // [N*tiles_img][Cout_pad] (mean, M2, count, 0) or nullptr
  // BatchNorm-backward reduction fused into a DATA-GRADIENT launch (round 3): this launch writes g = dL/dy of a conv -> BN -> ReLU
  // stage whose only consumer it is; with bn_part != nullptr the epilogue also reads that stage's pre-BN tensor at its output
  // positions and leaves the per-tile partial sums of g * [y > 0] and g * [y > 0] * xhat, which hpri_bn_relu_bwd_fused
  // finalizes instead of sweeping g and the pre-BN tensor once more (bn.hip: col_reduce).
  const float* bn_x; int bn_x_cs, bn_x_coff;        // the pre-BN tensor (fp32 NHWC view, same pixels and channels as y)
  const float* bn_mean; const float* bn_invstd; const float* bn_scale; const float* bn_shift;
  float* bn_part; int bn_cpart, bn_relu;            // [N*tiles_img][2][bn_cpart]
  int N, H, W, Cin_pad, Cout, Cout_pad, y_cw, accumulate, relu;
  int tiles_x, tiles_y;
  int items, per_xcd, banded;   // work items (pixel tiles x channel blocks), items per XCD band, item order (below)
  int ncu, stagger_cycles;      // compute units of the device (host: hipDeviceProp_t.multiProcessorCount) and the one-off delay of
                                // each CU's second occupant (below); 0 cycles = no stagger
#ifdef HPRI_STAMPS
  unsigned long long* stamps;   // diagnostic builds only (tools/build_wino4_diag.sh)
#endif
};

#ifdef HPRI_STAMPS
#define STAMP(i_)                                                                                          \
  {                                                                                                        \
    unsigned long long t_;                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                            \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if (a.stamps != nullptr && (threadIdx.x & 127) == 0)                                                   \
      a.stamps[((size_t)blockIdx.x * 2 + (threadIdx.x >> 7)) * 8 + (i_)] = t_;                              \
  }
#else
#define STAMP(i_)
#endif

#define W4_HW 18                           // halo grid: 10 rows x 18 columns of 128-byte pixel slots (32 channels)
#define W4_SLOTS (10 * W4_HW)
#define W4_AI 23                           // halo DMA instructions (8 slots each; the last one half used)
#define W4_A_BYTES (W4_AI * 1024)
#define W4_B_WAVE 8192                     // bytes per wave per stage: 4 frequencies x [64 n][8 k] fp32
// Halo slots: a row holds its even columns first, then the odd ones (a tile's columns 2 tx + c are then CONSECUTIVE slots over
// tx), and the eight 16-byte quads of a slot are XOR-swizzled with ((slot >> 1) + row) & 7: brute-forced conflict-free for every
// ds_read_b128 lane group, input row / column and quad (row-major slots with (slot >> 1) & 7 measured 0.57 conflict cycles per
// LDS cycle: up to 4-way)
#define W4_SWZ(slot_, hy_) ((((slot_) >> 1) + (hy_)) & 7)

__global__ __launch_bounds__(256, 2) void conv_wino4_kernel(Wino4Args a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * W4_A_BYTES + 4 * W4_B_WAVE];
  unsigned char* b_lds = smem + 2 * W4_A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int fa = wave;                              // frequency row
  const int tiles_img = a.tiles_x * a.tiles_y;
  const int nbc = a.Cout_pad >> 6;
  // Item order (round 3; workgroup id mod 8 labels the XCD under round-robin dispatch -- speed only, never correctness).  Two
  // things are re-read through an XCD's 4 MB L2: the input halos (by the nbc channel blocks of a tile, and by neighbouring
  // tiles) and the packed U slice of a channel block (Cin_pad x 4 KB, by every tile).  Measured per layer with the plain
  // order item = workgroup id (profiles/r03_wino4_layer_traffic_before_xcd.json) and with XCD bands (..._all_banded.json):
  //   nbc <= 4  BANDED: XCD x owns the items [x * per_xcd, (x+1) * per_xcd), channel blocks of a tile back to back, tiles in
  //             raster order: the halos meet in ONE L2 (read traffic 1.41 x nbc x input -> 1.1-1.3 x at 608x968 / 304x484,
  //             5.0 -> 3.8 x at 152x242) and all of U (<= 8 MB) still fits next to them;
  //   nbc >= 8  PLAIN: block nb of every tile lands on XCD nb mod 8, so an XCD keeps ITS one or two U slices (2-4 MB each) in
  //             L2 and re-reads the (small) input instead: 6.6 x against 12.9 x banded at 76x121, 3.4 x against 19 x at 38x60.
  const int item = a.banded ? (int)(blockIdx.x & 7) * a.per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (item >= a.items) return;
  const int bx = item / nbc, nb = item - bx * nbc;
  const int img = bx / tiles_img, tin = bx - img * tiles_img;
  const int ty0 = tin / a.tiles_x, tx0 = tin - ty0 * a.tiles_x;
  const int Y0 = ty0 * 8, X0 = tx0 * 16;

  // B^T row fa: two input rows r1, r2:  a=0: d0 - d2 | a=1: d1 + d2 | a=2: d2 - d1 | a=3: d1 - d3   (= s1 (d[r1] + sigma d[r2]))
  const int r1 = (fa == 0) ? 0 : 1, r2 = (fa == 3) ? 3 : 2;
  const float s1 = (fa == 2) ? -1.f : 1.f, s2 = (fa == 1 || fa == 2) ? 1.f : -1.f;
  const float sigma = s1 * s2;

  // ---- halo DMA (buffer_load ... lds): instruction i covers slots [8i, 8i+8); lane -> slot 8i + (lane>>3), physical quad
  //      lane&7.  The per-lane byte offset goes in as the 32-bit voffset of a buffer descriptor over this image; pixels outside
  //      the image get an offset beyond the descriptor's range: the hardware range check writes ZEROS into their LDS slots
  //      (tools/lds_dma_oob.hip), so there is no zero page, no select and no 64-bit address arithmetic per piece ----
  const float* ximg = a.x + (size_t)img * a.H * a.W * a.x_cs + a.x_coff;
  const hpri_rsrc_t rs_a =
      HPRI_MAKE_RSRC(ximg, (int)(((long long)a.H * a.W * a.x_cs - a.x_coff) * 4));
  constexpr int NIA = (W4_AI + 3) / 4;              // per wave (the last round is partial)
  constexpr unsigned OOB = 0xFFFFFFF0u;
  unsigned aoff[NIA];
#pragma unroll
  for (int q = 0; q < NIA; ++q) {
    const int slot = (q * 4 + wave) * 8 + (lane >> 3);
    unsigned off = OOB;
    if (slot < W4_SLOTS) {
      const int hy = slot / W4_HW, rem = slot - hy * W4_HW;
      const int hx = rem < W4_HW / 2 ? 2 * rem : 2 * (rem - W4_HW / 2) + 1;      // even columns first, then the odd ones
      const int iy = Y0 + hy - 1, ix = X0 + hx - 1;
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
        off = (unsigned)((iy * a.W + ix) * a.x_cs + ((((lane & 7) ^ W4_SWZ(slot, hy))) << 2)) * 4u;
    }
    aoff[q] = off;
  }
  const int lq = (lane & 7);
#define LOAD_A(chunk_)                                                                                                \
  {                                                                                                                   \
    unsigned char* la_ = smem + ((chunk_) & 1) * W4_A_BYTES;                                                          \
    const bool tail_ = ((chunk_) * 32 + 32) > a.Cin_pad;      /* last chunk of a channel count that is not a multiple of 32 */ \
    _Pragma("unroll") for (int q = 0; q < NIA; ++q) {                                                                 \
      const int inst_ = q * 4 + wave;                                                                                 \
      if (inst_ < W4_AI) {                                                                                            \
        unsigned vo_ = aoff[q];                                                                                       \
        if (tail_) {                                                                                                  \
          const int slot_ = inst_ * 8 + (lane >> 3);                                                                  \
          const int lquad_ = lq ^ W4_SWZ(slot_, slot_ / W4_HW);                                                       \
          if (((chunk_) * 32 + lquad_ * 4) >= a.Cin_pad) vo_ = OOB;                                                   \
        }                                                                                                             \
        HPRI_LDS_DMA16(rs_a, la_ + inst_ * 1024, vo_, (chunk_) * 128);                                               \
      }                                                                                                               \
    }                                                                                                                 \
  }
  // ---- weight DMA: stage s, piece p = 2 b + (n half): lane -> row n = 32 (p&1) + (lane>>1), physical 16-byte half lane&1,
  //      which holds logical half (lane&1) ^ ((n>>3)&1) (conflict-free ds_read_b128 of a lane's four k values) ----
  const unsigned goff0 = (unsigned)(((fa * 4) * a.Cout_pad + nb * 64 + (lane >> 1)) * 8 + 4 * ((lane & 1) ^ ((lane >> 4) & 1))) * 4u;
  const int gstep_b = a.Cout_pad * 8 * 4;           // bytes between column frequencies
  const hpri_rsrc_t rs_b =
      HPRI_MAKE_RSRC(a.up, (int)((long long)(a.Cin_pad >> 3) * 16 * 8 * a.Cout_pad * 4));
  unsigned char* bw = b_lds + wave * W4_B_WAVE;
#define LOAD_B(s_)                                                                                                    \
  {                                                                                                                   \
    const int sb_ = (s_) * 16 * gstep_b;                                                                              \
    B_PIECE(0) B_PIECE(1) B_PIECE(2) B_PIECE(3) B_PIECE(4) B_PIECE(5) B_PIECE(6) B_PIECE(7)                           \
  }
// NB the instruction's immediate offset is added to the LDS address as well as to the memory address: everything goes through soffset
#define B_PIECE(p_)                                                                                                   \
  HPRI_LDS_DMA16(rs_b, bw + (p_) * 1024, goff0, sb_ + ((p_) >> 1) * gstep_b + ((p_) & 1) * 1024);

  // lane's tile: (ty, tx) = (li>>3, li&7); its input pixel (r, c) is halo pixel (hy, hx) = (2 ty + r, 2 tx + c), stored in slot
  // hy * 18 + (hx & 1) * 9 + (hx >> 1)
  int pre[8];                                       // XOR-form halo addresses: [2 c + (row r1 | r2)]
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int hx = 2 * (li & 7) + c, hy1 = 2 * (li >> 3) + r1, hy2 = 2 * (li >> 3) + r2;
    const int h1 = hy1 * W4_HW + (hx & 1) * (W4_HW / 2) + (hx >> 1), h2 = hy2 * W4_HW + (hx & 1) * (W4_HW / 2) + (hx >> 1);
    pre[2 * c + 0] = h1 * 128 + (W4_SWZ(h1, hy1) << 4);
    pre[2 * c + 1] = h2 * 128 + (W4_SWZ(h2, hy2) << 4);
  }
  const int boff = li * 32 + ((lh ^ ((li >> 3) & 1)) << 4);

  f32x16 acc[4][2];                                 // [column frequency b][channel group nt]
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][nt][r] = 0.f;

  const int nstages = a.Cin_pad >> 3;
  const int nchunks = (a.Cin_pad + 31) >> 5;
  // the epilogue's bias quad, loaded here: four conditional loads in the epilogue were four serialized round trips
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias != nullptr) {
    const int n0b = nb * 64 + (tid & 15) * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) bias4[c] = a.bias[min(n0b + c, a.Cout - 1)];
#pragma unroll
    for (int c = 0; c < 4; ++c) if (n0b + c >= a.Cout) bias4[c] = 0.f;
  }
#ifndef WINO4_NO_STAGGER
  // Two workgroups share a CU, and all workgroups of a launch take the same time: without help both run their output
  // transform (no MFMAs) at the same moment, for the whole launch.  The SECOND occupants of the first round (the dispatcher
  // hands workgroups [0, ncu) to ncu different CUs, [ncu, 2 ncu) to the same CUs again) sleep once for a little more than
  // an epilogue (exchange + output transform + statistics: ~16.4 k cycles when shared); from then on each CU's two slots stay
  // that far apart and one's epilogue runs under the other's main loop (tools/wino4_stamps.py: 97-99 % of the epilogue time
  // covered).  Placement is not promised by HIP: a different dispatch order costs the overlap, never correctness.
  if ((unsigned)(blockIdx.x - a.ncu) < (unsigned)a.ncu && a.stagger_cycles > 0) {
    const long long wait = a.stagger_cycles;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
#endif
  STAMP(0)
#ifdef HPRI_STAMPS
  if (a.stamps != nullptr && (threadIdx.x & 127) == 0) {       // where this workgroup runs: HW_ID (CU, SE, ...) and XCC_ID
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.stamps[((size_t)blockIdx.x * 2 + (threadIdx.x >> 7)) * 8 + 6] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
  LOAD_A(0)
  LOAD_B(0)
#define WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
  f32x4 bfr[4][2];                                  // weights of the stage: bfr[b][nt][j] = U[(fa, b)][k = 4 lh + j][n = nt*32 + li]
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int aboff = (chunk & 1) * W4_A_BYTES;
    const int sg = min(4, nstages - chunk * 4);
    for (int g = 0; g < sg; ++g) {
      const int s = chunk * 4 + g;
      // the weights of this stage were issued one stage ago, BEFORE that stage's halo pieces (if any): in-order completion
      if (g == 1 && chunk + 1 < nchunks) { if (wave < 3) WAIT_VM(6); else WAIT_VM(5); }
      else WAIT_VM(0);
      if (g == 0) __builtin_amdgcn_s_barrier();     // the chunk's halo is visible to all four waves; all have left the previous chunk
#ifdef HPRI_STAMPS
      if (s == 0) STAMP(7)                          // end of the prologue: the first halo chunk and weight stage have landed
#endif
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bfr[b][nt] = *reinterpret_cast<const f32x4*>(bw + b * 2048 + nt * 1024 + boff);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the panel is in registers: the next stage may overwrite it
      __builtin_amdgcn_sched_barrier(0);
#if !(defined(WINO_DIAG) && (WINO_DIAG & 4))
      if (s + 1 < nstages) LOAD_B(s + 1)
#endif
#if !(defined(WINO_DIAG) && (WINO_DIAG & 8))
      if (g == 0 && chunk + 1 < nchunks) LOAD_A(chunk + 1)
#endif
      __builtin_amdgcn_sched_barrier(0);
      const int q16 = (2 * g + lh) << 4;            // this lane half's channel quad inside the chunk, as a byte offset
      f32x4 P[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(smem + ((pre[2 * c + 0] ^ q16) + aboff));
        const f32x4 d2 = *reinterpret_cast<const f32x4*>(smem + ((pre[2 * c + 1] ^ q16) + aboff));
        P[c] = d1 + sigma * d2;
      }
      const f32x4 v[4] = {P[0] - P[2], P[1] + P[2], P[2] - P[1], P[1] - P[3]};
#ifndef WINO4_NO_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[b][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[b][j], bfr[b][nt][j], acc[b][nt], 0, 0, 0);
#ifndef WINO4_NO_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
  }
#undef WAIT_VM
#undef LOAD_A
#undef LOAD_B
#undef B_PIECE
  STAMP(1)
  __syncthreads();

  // ------------------------------- epilogue -------------------------------
  // acc[b][nt][r]: tile t = (r&3) + 8*(r>>2) + 4*lh, channel nt*32 + li, frequency (fa, b), without the row sign s1.
  // Column transform in registers: Z[.][0] = M0 + M1 + M2, Z[.][1] = M1 - M2 - M3; the 4 x 2 planes meet in LDS (64 KB), then
  // every thread owns two (tile, channel quad) pairs: 8 float4 reads each, the row transform, bias, 2x2 pixels as float4.
  float* ex = reinterpret_cast<float*>(smem);       // [8 = 2 a + j][32 tiles][64 ch]
  static_assert(8 * 32 * 64 * 4 <= 2 * W4_A_BYTES + 4 * W4_B_WAVE, "exchange buffer must fit the staging LDS");
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r];
      ex[((fa * 2 + 0) * 32 + t) * 64 + nt * 32 + li] = s1 * (m0 + m1 + m2);
      ex[((fa * 2 + 1) * 32 + t) * 64 + nt * 32 + li] = s1 * (m1 - m2 - m3);
    }
  STAMP(4)
  __syncthreads();
  STAMP(5)
  const int oq = tid & 15;
  const int n0 = nb * 64 + oq * 4;
  f32x4 outv[2][4];                                 // [item][pixel] x 4 channels
  const int vrows = min(8, a.H - Y0), vcols = min(16, a.W - X0);
  // accumulate (a skip gradient already holds the other producer's part): ALL old values are loaded before the first store
  // -- a load behind a store waits for the store as well (vmcnt counts in order)
  f32x4 oldv[2][4];
  if (a.accumulate) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int ot = (tid >> 4) + 16 * it;
      const int py = 2 * (ot >> 3), px = 2 * (ot & 7);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int yy = py + (k >> 1), xx = px + (k & 1);
        const bool ok = yy < vrows && xx < vcols && n0 < a.y_cw;
        const float* src = a.y + ((size_t)(img * a.H + Y0 + min(yy, vrows - 1)) * a.W + X0 + min(xx, vcols - 1)) * a.y_cs + a.y_coff + min(n0, a.y_cw - 4);
        const f32x4 t = *reinterpret_cast<const f32x4*>(src);
        oldv[it][k] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ot = (tid >> 4) + 16 * it;            // tile of the workgroup: (ot>>3, ot&7)
    f32x4 m[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) m[i][j] = *reinterpret_cast<const f32x4*>(ex + ((i * 2 + j) * 32 + ot) * 64 + oq * 4);
    // Y = A^T Z with A^T = [1 1 1 0; 0 1 -1 -1]
    f32x4 o[4] = {m[0][0] + m[1][0] + m[2][0], m[0][1] + m[1][1] + m[2][1], m[1][0] - m[2][0] - m[3][0], m[1][1] - m[2][1] - m[3][1]};
    const int py = 2 * (ot >> 3), px = 2 * (ot & 7);
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
        if (a.accumulate) v += oldv[it][k];
        *reinterpret_cast<f32x4*>(dst) = v;
      }
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      outv[it][k] = ok ? v : zero4;
    }
  }
  STAMP(2)
  if (a.stats != nullptr) {
    // two-pass per-tile statistics over the valid pixels: thread (tid>>4, oq) holds 8 pixels of channels 4 oq .. 4 oq + 3
    __syncthreads();                                // everyone has read the exchange planes
    float* red = reinterpret_cast<float*>(smem);    // [16][64 ch] + [64] means
    const float cnt = (float)(vrows * vcols);
    f32x4 mean4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int ot = (tid >> 4) + 16 * it;
        const int py = 2 * (ot >> 3), px = 2 * (ot & 7);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool ok = (py + (k >> 1)) < vrows && (px + (k & 1)) < vcols;
          if (pass == 0) sacc += outv[it][k];
          else if (ok) { const f32x4 d = outv[it][k] - mean4; sacc += d * d; }
        }
      }
      *reinterpret_cast<f32x4*>(red + (tid >> 4) * 64 + oq * 4) = sacc;
      __syncthreads();
      if (tid < 64) {                               // channel c = tid: sum over the 16 thread rows
        float tsum = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) tsum += red[q * 64 + tid];
        if (pass == 0) red[16 * 64 + tid] = tsum / cnt;
        else a.stats[(size_t)bx * a.Cout_pad + nb * 64 + tid] = make_float4(red[16 * 64 + tid], tsum, cnt, 0.f);
      }
      __syncthreads();
      if (pass == 0) mean4 = *reinterpret_cast<const f32x4*>(red + 16 * 64 + oq * 4);
      __syncthreads();
    }
  }
  if (a.bn_part != nullptr) {
    // BatchNorm-backward partial sums of this tile (see Wino4Args): thread (tid>>4, oq) holds g at 8 pixels x 4 channels
    f32x4 sc4, sh4, mu4, is4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int n = min(n0 + c, a.Cout - 1);
      sc4[c] = a.bn_scale[n]; sh4[c] = a.bn_shift[n]; mu4[c] = a.bn_mean[n]; is4[c] = a.bn_invstd[n];
    }
    f32x4 xv[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int ot = (tid >> 4) + 16 * it;
      const int py = 2 * (ot >> 3), px = 2 * (ot & 7);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int yy = min(py + (k >> 1), vrows - 1), xx = min(px + (k & 1), vcols - 1);
        xv[it][k] = *reinterpret_cast<const f32x4*>(a.bn_x + ((size_t)(img * a.H + Y0 + yy) * a.W + X0 + xx) * a.bn_x_cs + a.bn_x_coff +
                                                    min(n0, a.Cout_pad - 4));
      }
    }
    f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          // outv is zero outside the image and beyond Cout, so those positions add nothing
          const float gj = (!a.bn_relu || (xv[it][k][c] * sc4[c] + sh4[c] > 0.f)) ? outv[it][k][c] : 0.f;
          t1[c] += gj;
          t2[c] += gj * ((xv[it][k][c] - mu4[c]) * is4[c]);
        }
    __syncthreads();                                // everyone has read the exchange planes (and the statistics scratch)
    float* red = reinterpret_cast<float*>(smem);    // [2][16][64 ch]
    *reinterpret_cast<f32x4*>(red + (tid >> 4) * 64 + oq * 4) = t1;
    *reinterpret_cast<f32x4*>(red + 1024 + (tid >> 4) * 64 + oq * 4) = t2;
    __syncthreads();
    if (tid < 128) {                                // (sum, channel) = (tid >> 6, tid & 63): fixed order over the 16 thread rows
      const int which = tid >> 6, c = tid & 63;
      float tsum = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) tsum += red[which * 1024 + q * 64 + c];
      if (nb * 64 + c < a.bn_cpart) a.bn_part[((size_t)bx * 2 + which) * a.bn_cpart + nb * 64 + c] = tsum;
    }
  }
  STAMP(3)
}

// ---- filter transform: U = G g G^T, packed with k innermost: [Cin_pad/8][16][Ncols_pad][8] ---------------------------------
// mode 0: forward       g(k = c, col = n)[t] = W[n][c][t]            (W: [Cout][Cin][3][3], src_d1 = Cin)
// mode 1: data gradient g(k = n, col = c)[t] = W[n][c][8 - t]        (K = Cout, columns = Cin)
__global__ void wino4_pack_kernel(const float* __restrict__ w, float* __restrict__ up, int mode, int K, int Ncols, int Ncols_pad,
                                  int stages, int src_d1, const float* __restrict__ colscale) {
  const size_t total = (size_t)stages * 8 * Ncols_pad;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int kk = (int)(idx % 8);
    const int col = (int)((idx / 8) % Ncols_pad);
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
      const float u[4] = {gg[i][0], 0.5f * (gg[i][0] + gg[i][1] + gg[i][2]), 0.5f * (gg[i][0] - gg[i][1] + gg[i][2]), gg[i][2]};
#pragma unroll
      for (int j = 0; j < 4; ++j) up[(((size_t)st * 16 + i * 4 + j) * Ncols_pad + col) * 8 + kk] = u[j];
    }
  }
}

// Same size as hpri_wino_packed_floats; the layout differs (k innermost).
extern "C" int hpri_wino4_pack(const float* w, float* up, const float* colscale, int mode, int K, int Ncols, int Ncols_pad,
                               int src_d1, hipStream_t stream) {
  HPRI_REQUIRE(w && up, "wino4_pack: null pointer");
  HPRI_REQUIRE((mode == 0 || mode == 1) && K > 0 && Ncols > 0 && Ncols_pad >= Ncols && Ncols_pad % 64 == 0, "wino4_pack: bad arguments");
  const int stages = hpri_cdiv(K, 8);
  const size_t total = (size_t)stages * 8 * Ncols_pad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wino4_pack_kernel, dim3(blocks), dim3(256), 0, stream, w, up, mode, K, Ncols, Ncols_pad, stages, src_d1, colscale);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// Stamp buffer of the diagnostic builds (-DHPRI_STAMPS, tools/build_wino4_diag.sh; declared in include/hyperpri_hip_diag.h).  The
// product library keeps no state and has no such entry point.
#ifdef HPRI_STAMPS
static unsigned long long* hpri_wino4_stamps = nullptr;
extern "C" int hpri_wino4_set_stamps(unsigned long long* p) { hpri_wino4_stamps = p; return HPRI_OK; }
#elif defined(HPRI_DIAG_KERNELS)
extern "C" int hpri_wino4_set_stamps(unsigned long long*) {
  return hpri_set_error(HPRI_ERR_UNSUPPORTED, "wino4_set_stamps: this library was built without -DHPRI_STAMPS");
}
#endif
#define W4_STAGGER_CYCLES 20000   // a little more than one epilogue with a partner on the CU (16.4 k cycles, tools/wino4_stamps.py)

extern "C" int hpri_conv_wino4_plan(int N, int H, int W, int* stat_tiles) {
  *stat_tiles = N * hpri_cdiv(H, 8) * hpri_cdiv(W, 16);
  return HPRI_OK;
}

extern "C" int hpri_conv_wino4_bnred(const float* x, int x_cs, int x_coff, const float* up, float* y, int y_cs, int y_coff, int N, int H,
                                     int W, int Cin_pad, int Cout, int Cout_pad, int y_cw, const float* bn_x, int bn_x_cs, int bn_x_coff,
                                     const float* bn_mean, const float* bn_invstd, const float* bn_scale, const float* bn_shift,
                                     int bn_relu, float* bn_part, int bn_cpart, hipStream_t stream);

// 3x3 / pad 1 / stride 1 convolution, Winograd F(2x2,3x3), fp32.  x: fp32 NHWC view with channels [Cin, Cin_pad) zero
// (Cin_pad a multiple of 8); up from hpri_wino4_pack; accumulate bit 0: y += result, bit 1: ReLU epilogue; statistics:
// one (mean, M2, count, 0) record per 16 x 8-pixel tile and channel (hpri_conv_wino4_plan).
static int wino4_launch(const float* x, int x_cs, int x_coff, const float* up, const float* bias, float* y, int y_cs,
                        int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw,
                        int accumulate, const float* bn_x, int bn_x_cs, int bn_x_coff, const float* bn_mean,
                        const float* bn_invstd, const float* bn_scale, const float* bn_shift, int bn_relu, float* bn_part,
                        int bn_cpart, hipStream_t stream) {
  HPRI_REQUIRE(x && up && y, "conv_wino4: null pointer");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0, "conv_wino4: empty image");
  HPRI_REQUIRE(Cin_pad > 0 && Cin_pad % 8 == 0, "conv_wino4: Cin_pad must be a positive multiple of 8");
  HPRI_REQUIRE(Cout_pad % 64 == 0 && Cout <= Cout_pad && Cout > 0, "conv_wino4: Cout_pad must be a multiple of 64 >= Cout");
  HPRI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0 && x_coff + Cin_pad <= x_cs, "conv_wino4: input channel stride/offset");
  HPRI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)up & 15) == 0, "conv_wino4: pointers must be 16-byte aligned");
  HPRI_REQUIRE((long long)H * W * x_cs * 4 < (1ll << 32) - 65536, "conv_wino4: one image of the input view exceeds 4 GiB (32-bit buffer offsets)");
  HPRI_REQUIRE((long long)(Cin_pad / 8) * 16 * 8 * Cout_pad * 4 < (1ll << 31), "conv_wino4: packed weights exceed 2 GiB");
  HPRI_REQUIRE(!((accumulate & 1) && stats != nullptr), "conv_wino4: statistics are not available together with accumulate");
  Wino4Args a;
  a.x = x; a.x_cs = x_cs; a.x_coff = x_coff; a.up = up; a.bias = bias; a.y = y; a.y_cs = y_cs; a.y_coff = y_coff;
  a.stats = reinterpret_cast<float4*>(stats);
  a.N = N; a.H = H; a.W = W; a.Cin_pad = Cin_pad; a.Cout = Cout; a.Cout_pad = Cout_pad;
  a.y_cw = y_cw < Cout ? Cout : y_cw; a.accumulate = accumulate & 1; a.relu = (accumulate >> 1) & 1;
  HPRI_REQUIRE(a.y_cw + y_coff <= y_cs, "conv_wino4: output channels exceed the channel stride");
  HPRI_REQUIRE(y_cs % 4 == 0 && y_coff % 4 == 0 && a.y_cw % 4 == 0 && ((uintptr_t)y & 15) == 0,
               "conv_wino4: the output view must be float4-aligned (stride, offset and written width multiples of 4)");
  a.tiles_x = hpri_cdiv(W, 16); a.tiles_y = hpri_cdiv(H, 8);
  a.bn_x = bn_x; a.bn_x_cs = bn_x_cs; a.bn_x_coff = bn_x_coff; a.bn_mean = bn_mean; a.bn_invstd = bn_invstd; a.bn_scale = bn_scale;
  a.bn_shift = bn_shift; a.bn_part = bn_part; a.bn_cpart = bn_cpart; a.bn_relu = bn_relu;
  if (bn_part != nullptr) {
    HPRI_REQUIRE(bn_x && bn_mean && bn_invstd && bn_scale && bn_shift, "conv_wino4_bnred: null pointer");
    HPRI_REQUIRE(!(accumulate & 1) && stats == nullptr, "conv_wino4_bnred: not together with accumulate or statistics");
    HPRI_REQUIRE(bn_x_cs % 4 == 0 && bn_x_coff % 4 == 0 && bn_x_coff + Cout_pad <= bn_x_cs && ((uintptr_t)bn_x & 15) == 0,
                 "conv_wino4_bnred: the pre-BN view must be float4-aligned and hold Cout_pad channels");
    HPRI_REQUIRE(bn_cpart >= Cout && bn_cpart % 4 == 0, "conv_wino4_bnred: partial width must cover the channels");
  }
  a.ncu = hpri_cu_count(); a.stagger_cycles = W4_STAGGER_CYCLES;
#ifdef HPRI_STAMPS
  a.stamps = hpri_wino4_stamps;
#endif
  const long long items = (long long)N * a.tiles_x * a.tiles_y * (Cout_pad / 64);
  HPRI_REQUIRE(items < (1ll << 28), "conv_wino4: too many work items");
  a.items = (int)items; a.per_xcd = (int)((items + 7) / 8);
  a.banded = (Cout_pad / 64) <= 4;
  dim3 grid((unsigned)(a.banded ? a.per_xcd * 8 : a.items), 1u, 1u);
  hipLaunchKernelGGL(conv_wino4_kernel, grid, dim3(256), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_conv_wino4(const float* x, int x_cs, int x_coff, const float* up, const float* bias, float* y, int y_cs,
                               int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw,
                               int accumulate, hipStream_t stream) {
  return wino4_launch(x, x_cs, x_coff, up, bias, y, y_cs, y_coff, stats, N, H, W, Cin_pad, Cout, Cout_pad, y_cw, accumulate, nullptr, 0,
                      0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, stream);
}

// The same convolution as a DATA GRADIENT (mode-1 pack, no bias) that also leaves the BatchNorm-backward partial sums of the stage
// whose output gradient it writes: bn_x = that stage's pre-BN tensor (same pixels / channels as y), per-channel mean / invstd /
// scale / shift, bn_relu = the stage ends in a ReLU; bn_part[N * tiles][2][bn_cpart] (tiles: hpri_conv_wino4_plan) receives
// sum g*[y>0] and sum g*[y>0]*xhat per 16 x 8-pixel tile; finish with hpri_bn_relu_bwd_fused.
extern "C" int hpri_conv_wino4_bnred(const float* x, int x_cs, int x_coff, const float* up, float* y, int y_cs, int y_coff, int N, int H,
                                     int W, int Cin_pad, int Cout, int Cout_pad, int y_cw, const float* bn_x, int bn_x_cs, int bn_x_coff,
                                     const float* bn_mean, const float* bn_invstd, const float* bn_scale, const float* bn_shift,
                                     int bn_relu, float* bn_part, int bn_cpart, hipStream_t stream) {
  HPRI_REQUIRE(bn_part != nullptr, "conv_wino4_bnred: null partial buffer");
  return wino4_launch(x, x_cs, x_coff, up, nullptr, y, y_cs, y_coff, nullptr, N, H, W, Cin_pad, Cout, Cout_pad, y_cw, 0, bn_x, bn_x_cs,
                      bn_x_coff, bn_mean, bn_invstd, bn_scale, bn_shift, bn_relu, bn_part, bn_cpart, stream);
}
