// Plane-fed GEMM for the 1x1 forms of the bf16 precision mode: nn.Linear / Conv2d(k=1) forward and data gradient
// (reference models.py:105-115, 143 -- SpectralUNET's per-pixel MLP -- and model_parts.py:96), ConvTranspose2d(k=2, s=2) forward
// (model_parts.py:63-64: one GEMM Cin -> 4*Cup per input pixel whose epilogue scatters the 2x2 patches) and its data gradient
// (K = 4*Cup gathered from the four parities).  Operands are bf16 PLANES resident in HBM, both staged by LDS-DMA; the round-1
// kernels these forms ran until now (conv_fwd_bf16.hip) read fp32 activations and convert while staging.
//
//   workgroup  256 threads = 4 waves, 256 pixels x 128 columns, 73 KB of LDS: two independent workgroups per CU, persistent
//              (2 x CUs workgroups, fixed item lists; the next item's first two stages load under the current item's epilogue).
//              Items of one XCD (workgroup id mod 8) are the column blocks of ONE pixel tile back to back, pixel tiles in order:
//              the tile's rows are read from HBM once and served to the other column blocks by that XCD's L2.
//   MFMA       v_mfma_f32_16x16x32_bf16, weights as the A operand (a lane's four accumulator registers are four consecutive
//              columns of one pixel: 16-byte stores, no LDS transpose), wave tile 64 px x 128 columns = 4 x 8 accumulator tiles
//              (128 VGPRs): 12 fragment reads per 32 MFMAs (the 3x3 kernel's 64 x 64 wave tile: 8 per 16).
//   stages     one 32-deep k-chunk per stage: 256 px x 64 B + 128 columns x 64 B = 24 KB, TRIPLE-buffered (two stages in flight
//              behind the one being multiplied), one counted vmcnt + one barrier per stage of 32 MFMAs per wave; k-slots
//              XOR-swizzled through the DMA source address exactly as in conv_bf16v3.hip (conflict-free ds_read_b128).
//   epilogue   bias, ReLU, accumulate, fp32 and / or bf16 output views, per-tile BatchNorm partials (mode 0; tiles never
//              straddle an image, so per-image statistics -- SpectralUNET's groups -- are sums of whole records), the
//              depth-to-space scatter of the transposed convolution (mode 1).
#include "common.h"

typedef h16_t bf16x8 __attribute__((ext_vector_type(8)));

#ifndef G3_NT
#define G3_NT 8                                 // column tiles of 16 per wave
#endif
#define G3_BN (16 * G3_NT)                      // columns per workgroup
#define G3_BQ (G3_NT / 4)                       // weight DMA pieces per wave and stage
#define G3_A_BYTES (16 * 1024)
#define G3_B_BYTES (G3_BN * 64)
#define G3_STAGE_BYTES (G3_A_BYTES + G3_B_BYTES)

struct GemmV3Args {
  const h16_t* xp; int x_cs, x_coff;        // A planes: elements per pixel row (multiple of 8), first channel (multiple of 8)
  const h16_t* wp;                          // packed [chunk][Ncols_pad][32] (hpri_pack_weight_bf16, T = 1)
  const float* bias;                         // per output channel (mode 1: per Cup channel), or nullptr
  float* y; int y_cs, y_coff, y_cw;          // fp32 output view (nullptr: none)
  h16_t* y16; int y16_cs, y16_coff;         // bf16 output view (nullptr: none)
  int acc16;                                 // mode 0: add the bf16 view's old contents (accumulate flag with no fp32 view)
  float4* stats; int stat_cp;                // mode 0: [tile][stat_cp] (mean, M2, count) records, or nullptr
  int N, HW;                                 // images, GEMM rows per image (mode 1 / 2: H * W of the low-resolution grid)
  int W, H2, W2, py0, px0, cup;              // transposed-convolution geometry (modes 1, 2)
  int nchunks, Ncols, Ncols_pad;             // k-chunks of 32, output columns, packed column count (multiple of 64)
  int accumulate, relu;
  int tiles_img, ntiles, nb_count, per_xcd;
  int ncu, stagger_cycles;
  unsigned *queue, *queue_clear;             // item counters of this launch and the half it zeroes for the next one (common.h), or nullptr = fixed item lists
};

__device__ __forceinline__ float g3_row_sum(float v) {
#define G3_DPP_ADD(ctrl_)                                                                                \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl_, 0xF, 0xF, true))
  G3_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
  G3_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
  G3_DPP_ADD(0x141);   // row_half_mirror
  G3_DPP_ADD(0x140);   // row_mirror
#undef G3_DPP_ADD
  return v;
}

struct G3Tile { int img, p0, valid, nb, bx; };

// MODE 0: rows = pixels of x, output row = the same pixel.  MODE 1: the same rows, output scattered depth-to-space.
// MODE 2: row p of the low-resolution grid gathers its K = 4*cup values from the four parities of the high-resolution planes.
template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_bf16v3_kernel(GemmV3Args a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * G3_STAGE_BYTES + 2 * G3_BN * 4 + 16];
  float* bias_lds = reinterpret_cast<float*>(smem + 3 * G3_STAGE_BYTES);     // [2 slots][G3_BN]
  int* next_lds = reinterpret_cast<int*>(smem + 3 * G3_STAGE_BYTES + 2 * G3_BN * 4);      // [2 slots]: the next item's index in the band (item queue)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int xcd = blockIdx.x & 7, nloc = (int)(gridDim.x >> 3);
  const int items_all = a.ntiles * a.nb_count;
  auto tile_of = [&](int k, G3Tile& t) -> bool {
    if (k >= a.per_xcd) return false;
    const int item = xcd * a.per_xcd + k;
    if (item >= items_all) return false;
    t.bx = item / a.nb_count; t.nb = item - t.bx * a.nb_count;
    t.img = t.bx / a.tiles_img;
    t.p0 = (t.bx - t.img * a.tiles_img) * 256;
    t.valid = min(256, a.HW - t.p0);
    return true;
  };

  constexpr unsigned OOB = HPRI_DMA_OOB;
  unsigned aoff[4], goff[G3_BQ];
  hpri_rsrc_t rs_a = HPRI_MAKE_RSRC(a.xp, 0x7FFFFF00);
  const hpri_rsrc_t rs_b = HPRI_MAKE_RSRC(a.wp, 0x7FFFFF00);
  const int chunk_bytes = a.Ncols_pad * 64;
  const int cpt = MODE == 2 ? (a.cup >> 5) : 1;            // mode 2: 32-channel chunks per parity
  auto setup_dma = [&](const G3Tile& t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pix = (q * 4 + wave) * 16 + (lane >> 2);
      const unsigned sw = (unsigned)(((lane & 3) ^ (((pix >> 2) & 1) << 1)) << 3);
      unsigned off = OOB;
      if (pix < t.valid) {
        if (MODE == 2) {
          const int p = t.p0 + pix, yy = p / a.W, xx = p - yy * a.W;
          off = (unsigned)(((a.py0 + 2 * yy) * a.W2 + a.px0 + 2 * xx) * a.x_cs + (int)sw) * 2u;
        } else {
          off = (unsigned)(pix * a.x_cs + (int)sw) * 2u;
        }
      }
      aoff[q] = off;
    }
#pragma unroll
    for (int q = 0; q < G3_BQ; ++q) {
      const int row = (q * 4 + wave) * 16 + (lane >> 2);
      const int n = min(t.nb * G3_BN + row, a.Ncols_pad - 1);      // rows beyond the pack: any packed row (their columns are never written)
      goff[q] = (unsigned)(n * 32 + (((lane & 3) ^ (((row >> 2) & 1) << 1)) << 3)) * 2u;
    }
    // base of the rows this item reads: mode 0 / 1 the tile's first pixel (offsets stay below 2^22 whatever the tensor size),
    // mode 2 the image of the high-resolution planes (host: one image < 2 GiB); wave-uniform by construction, and said so
    const size_t row0 = MODE == 2 ? (size_t)t.img * a.H2 * a.W2 : (size_t)t.img * a.HW + t.p0;
    const unsigned long long pb = (unsigned long long)(uintptr_t)(a.xp + row0 * a.x_cs + a.x_coff);
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pb), phi = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32));
    (void)plo; (void)phi;
    rs_a = HPRI_MAKE_RSRC((((unsigned long long)phi << 32) | plo), 0x7FFFFF00);
  };
  (void)goff; (void)chunk_bytes; (void)rs_b; (void)cpt;
  // scalar byte offset of k-chunk c inside a pixel row of the A planes
  auto a_soff = [&](int c) -> int {
    if (MODE == 2) { const int tap = c / cpt, cc = c - tap * cpt; return (((tap >> 1) * a.W2 + (tap & 1)) * a.x_cs + cc * 32) * 2; }
    return c * 64;
  };
#define G3_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define G3_BARRIER()                         \
  __builtin_amdgcn_sched_barrier(0);         \
  __builtin_amdgcn_s_barrier();              \
  __builtin_amdgcn_sched_barrier(0)
#define G3_DMA_A_AT(bo_, c_, q_) HPRI_LDS_DMA16(rs_a, smem + (bo_) + ((q_) * 4 + wave) * 1024, aoff[q_], a_soff(c_))
#define G3_DMA_B_AT(bo_, c_, q_) HPRI_LDS_DMA16(rs_b, smem + (bo_) + G3_A_BYTES + ((q_) * 4 + wave) * 1024, goff[q_], (c_) * chunk_bytes)
#define G3_DMA_A(buf_, c_, q_) G3_DMA_A_AT((buf_) * G3_STAGE_BYTES, c_, q_)
#define G3_DMA_B(buf_, c_, q_) G3_DMA_B_AT((buf_) * G3_STAGE_BYTES, c_, q_)
#define G3_ISSUE(buf_, c_)                                           \
  {                                                                  \
    _Pragma("unroll") for (int q = 0; q < G3_BQ; ++q) G3_DMA_B(buf_, c_, q); \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) G3_DMA_A(buf_, c_, q); \
  }

  // fragment addresses: pixel (wave*4 + mt)*16 + li resp. column row nt*16 + li, logical k-slot lq
  const int fofs = li * 64 + ((lq ^ (((li >> 2) & 1) << 1)) << 4);
  const int S = a.nchunks;

  G3Tile cur, nxt;
  // first item: see conv_bf16v3.hip (fixed lists, or the item queue of common.h)
  int k = (int)(blockIdx.x >> 3);
  unsigned* const q = a.queue;
  if (blockIdx.x == 0) hpri_q_clear(a.queue_clear, tid);
  const int nstatic = min(nloc, a.ncu >> 3);
  const bool late_start = (unsigned)(blockIdx.x - a.ncu) < (unsigned)a.ncu;
  unsigned pend = 0u;                          // (thread 0) the ticket of the NEXT item, drawn one item ahead
  if (q != nullptr && tid == 0) pend = hpri_q_draw(q, xcd, k < nstatic ? 1u : 2u);
  if (late_start && a.stagger_cycles > 0) {      // see conv_wino4.hip: the CU's second occupant starts late once
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < (long long)a.stagger_cycles) __builtin_amdgcn_s_sleep(32);
  }
  if (q != nullptr && k >= nstatic) {
    if (tid == 0) { next_lds[0] = nstatic + (int)pend; pend += 1u; }
    __syncthreads();
    k = __builtin_amdgcn_readfirstlane(next_lds[0]);
    __syncthreads();
  }
  bool have = tile_of(k, cur);
  if (!have) return;
  setup_dma(cur);
  float bias_next[1] = {0.f};
  auto load_bias = [&](const G3Tile& t) {
    if (tid < G3_BN) {
      const int col = t.nb * G3_BN + tid;
      const int ch = MODE == 1 ? col % a.cup : col;
      bias_next[0] = (a.bias != nullptr && col < a.Ncols) ? a.bias[ch] : 0.f;
    }
  };
  load_bias(cur);
  G3_ISSUE(0, 0)
  if (S > 1) G3_ISSUE(1, 1)
  int slot = 0;

  while (have) {
    if (tid < G3_BN) bias_lds[slot * G3_BN + tid] = bias_next[0];      // visible behind the first stage's barrier
    if (q != nullptr && tid == 0) next_lds[slot] = nstatic + (int)pend;      // the next item's ticket, parked (conv_bf16v3.hip)

    f32x4 acc[4][G3_NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < G3_NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Stage s multiplies buffer s % 3; at its top the wave's own pieces of stage s have landed when at most the 4 + G3_BQ of
    // stage s+1 are still in flight; behind the barrier every wave has left stage s-1, whose buffer then takes stage s+2.
    // ONE loop with the buffer index in a scalar register (a three-times unrolled loop with two early exits made hipcc keep a
    // second copy of the accumulators: 180-340 spilled registers for the 4 x 8 tile).
    int bo = 0;                                  // byte offset of the buffer of stage s
    for (int s = 0; s < S; ++s) {
      if (s + 1 < S) G3_WAIT_VM(4 + G3_BQ); else G3_WAIT_VM(0);
      G3_BARRIER();
      const unsigned char* ab_ = smem + bo + wave * 4096 + fofs;
      const unsigned char* bb_ = smem + bo + G3_A_BYTES + fofs;
      bf16x8 fa_[4], fb_[G3_NT];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) fa_[mt] = *reinterpret_cast<const bf16x8*>(ab_ + mt * 1024);
#pragma unroll
      for (int nt = 0; nt < G3_NT; ++nt) fb_[nt] = *reinterpret_cast<const bf16x8*>(bb_ + nt * 1024);
      const bool more_ = s + 2 < S;
      const int bo2 = bo == 0 ? 2 * G3_STAGE_BYTES : bo - G3_STAGE_BYTES;          // buffer (s + 2) % 3
      __builtin_amdgcn_s_setprio(1);
      // (ONE straight-line MFMA sequence: two copies of it under an if / else made hipcc merge the accumulators through copies
      // and spill ~200 registers; a scalar branch around each of the six DMA instructions costs two instructions)
#pragma unroll
      for (int nt = 0; nt < G3_NT; ++nt) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          acc[mt][nt] = HPRI_MFMA_16X16X32(fb_[nt], fa_[mt], acc[mt][nt], 0, 0, 0);
          if (more_ && nt == 0 && mt < G3_BQ) { G3_DMA_B_AT(bo2, s + 2, mt); }
          if (more_ && nt == 1) { G3_DMA_A_AT(bo2, s + 2, mt); }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      bo = bo == 2 * G3_STAGE_BYTES ? 0 : bo + G3_STAGE_BYTES;
    }
    G3_BARRIER();                                // every wave has left the main loop: all three buffers are free
    if (q != nullptr) k = __builtin_amdgcn_readfirstlane(next_lds[slot]);
    else k += nloc;
    have = tile_of(k, nxt);
    if (have) {
      setup_dma(nxt);
      G3_ISSUE(0, 0)
      if (S > 1) G3_ISSUE(1, 1)
      load_bias(nxt);
      if (q != nullptr && tid == 0) pend = hpri_q_draw(q, xcd, 1u);      // the ticket of the item after that one
    }

    // ------------------------------- epilogue -------------------------------
    // acc[mt][nt][r]: pixel (wave*4 + mt)*16 + li of the tile, column nb*128 + nt*16 + 4*lq + r
    const int col0 = cur.nb * G3_BN + 4 * lq;
    const float relu_floor = a.relu ? 0.f : -__builtin_huge_valf();
#pragma unroll
    for (int nt = 0; nt < G3_NT; ++nt) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_lds + slot * G3_BN + nt * 16 + 4 * lq);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        f32x4 v = acc[mt][nt] + b4;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], relu_floor);
        acc[mt][nt] = v;
      }
    }
    unsigned vmask = 0u;
    long long orow[4];                           // output pixel row of this lane's pixel of M-tile mt (mode 1: of parity 0)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int pix = (wave * 4 + mt) * 16 + li;
      if (pix < cur.valid) vmask |= 1u << mt;
      const int p = cur.p0 + min(pix, cur.valid - 1);
      if (MODE == 1) {
        const int yy = p / a.W, xx = p - yy * a.W;
        orow[mt] = ((long long)cur.img * a.H2 + a.py0 + 2 * yy) * a.W2 + a.px0 + 2 * xx;
      } else {
        orow[mt] = (long long)cur.img * a.HW + p;
      }
    }
    const int ncol_lim = MODE == 1 ? a.Ncols : a.y_cw;       // columns that are written
    // two halves of four column tiles each (an accumulating half holds 64 registers of old values next to the 128 accumulators)
#pragma unroll
    for (int h = 0; h < G3_NT / 4; ++h) {
      long long coff[4];                         // element offset of (row 0, first column of column tile nt) for this lane
      bool cok[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nt = h * 4 + q, col = col0 + nt * 16;
        cok[q] = col < ncol_lim;
        if (MODE == 1) {
          const int cb = cur.nb * G3_BN + nt * 16, tap = cb / a.cup, co = cb - tap * a.cup + 4 * lq;
          coff[q] = (long long)((tap >> 1) * a.W2 + (tap & 1));          // rows to add
          coff[q] = (coff[q] << 20) | (long long)co;                    // (row delta, channel) packed: both < 2^20
        } else {
          coff[q] = col;
        }
      }
      if (a.y != nullptr) {
        if (MODE != 1 && a.accumulate) {          // (the transposed convolution's forward never accumulates: hpri_convt_fwd_bf16v3; with the
                                                  //  64 registers of old values compiled in, that form spilled 10)
          f32x4 old_[4][4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const long long rr = MODE == 1 ? orow[mt] + (coff[q] >> 20) : orow[mt];
              const long long cc = MODE == 1 ? (coff[q] & 0xFFFFF) : coff[q];
              old_[mt][q] = (((vmask >> mt) & 1u) && cok[q]) ? *reinterpret_cast<const f32x4*>(a.y + rr * a.y_cs + a.y_coff + cc)
                                                            : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[mt][h * 4 + q] += old_[mt][q];
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          if ((vmask >> mt) & 1u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const long long rr = MODE == 1 ? orow[mt] + (coff[q] >> 20) : orow[mt];
              const long long cc = MODE == 1 ? (coff[q] & 0xFFFFF) : coff[q];
              if (cok[q]) *reinterpret_cast<f32x4*>(a.y + rr * a.y_cs + a.y_coff + cc) = acc[mt][h * 4 + q];
            }
          }
        }
      }
      if (a.y16 != nullptr) {
        if (MODE == 0 && a.acc16) {
          // round 4: the bf16 rows hold an earlier contribution to the same gradient (SpectralUNET's skips have two consumers):
          // read, add in fp32, round once on the way back.  All loads of the half before its stores (32 registers of old values).
          bf16x4_t o16[4][4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const bf16x4_t z = {(h16_t)0.f, (h16_t)0.f, (h16_t)0.f, (h16_t)0.f};
              o16[mt][q] = (((vmask >> mt) & 1u) && cok[q]) ? *reinterpret_cast<const bf16x4_t*>(a.y16 + orow[mt] * a.y16_cs + a.y16_coff + coff[q]) : z;
            }
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[mt][h * 4 + q][r] += (float)o16[mt][q][r];
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          if ((vmask >> mt) & 1u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const long long rr = MODE == 1 ? orow[mt] + (coff[q] >> 20) : orow[mt];
              const long long cc = MODE == 1 ? (coff[q] & 0xFFFFF) : coff[q];
              if (cok[q]) {
                bf16x4_t hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = (h16_t)acc[mt][h * 4 + q][r];
                *reinterpret_cast<bf16x4_t*>(a.y16 + rr * a.y16_cs + a.y16_coff + cc) = hv;
              }
            }
          }
        }
      }
    }
    if (MODE == 0 && a.stats != nullptr) {
      // per-tile, per-column (mean, M2, count): exact two-pass record per wave, Chan merge of the four waves (conv_bf16v3.hip)
      float cntl = (float)__builtin_popcount(vmask);
      const float cntw = g3_row_sum(cntl);
      const float inv = cntw > 0.f ? 1.f / cntw : 0.f;
      float* red = reinterpret_cast<float*>(smem + 2 * G3_STAGE_BYTES);      // buffer 2: [4 waves][G3_BN columns][2] + [4] counts
#pragma unroll
      for (int nt = 0; nt < G3_NT; ++nt) {
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          if ((vmask >> mt) & 1u) s1 += acc[mt][nt];
        f32x4 mw, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) mw[r] = g3_row_sum(s1[r]) * inv;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          if ((vmask >> mt) & 1u) { const f32x4 d = acc[mt][nt] - mw; s2 += d * d; }
#pragma unroll
        for (int r = 0; r < 4; ++r) s2[r] = g3_row_sum(s2[r]);
        if (li == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            red[(wave * G3_BN + nt * 16 + 4 * lq + r) * 2 + 0] = mw[r];
            red[(wave * G3_BN + nt * 16 + 4 * lq + r) * 2 + 1] = s2[r];
          }
        }
      }
      if (lane == 0) red[4 * 2 * G3_BN + wave] = cntw;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      G3_BARRIER();
      if (tid < G3_BN) {
        float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
          const float nb_ = red[4 * 2 * G3_BN + w2];
          if (nb_ > 0.f) {
            const float mb = red[(w2 * G3_BN + tid) * 2 + 0], qb = red[(w2 * G3_BN + tid) * 2 + 1];
            const float tot = n + nb_, delta = mb - mean, f = __builtin_amdgcn_rcpf(tot) * nb_;
            mean += delta * f;
            m2 += qb + delta * delta * (n * f);
            n = tot;
          }
        }
        if (cur.nb * G3_BN + tid < a.stat_cp) a.stats[(size_t)cur.bx * a.stat_cp + cur.nb * G3_BN + tid] = make_float4(mean, m2, n, 0.f);
      }
      // (buffer 2 is refilled by stage 2 of the next item, i.e. behind that item's first barriers)
    }
    cur = nxt;
    slot ^= 1;
  }
#undef G3_ISSUE
#undef G3_DMA_A
#undef G3_DMA_B
#undef G3_DMA_A_AT
#undef G3_DMA_B_AT
#undef G3_WAIT_VM
#undef G3_BARRIER
}

// ---------------------------------------------------------------------------------------------------------------------
#define G3_STAGGER_CYCLES 4000

extern "C" int hpri_gemm_bf16v3_plan(int N, long long HW, int* stat_tiles) {
  HPRI_REQUIRE(N > 0 && HW > 0 && stat_tiles, "gemm_bf16v3_plan: bad arguments");
  *stat_tiles = (int)(N * ((HW + 255) / 256));
  return HPRI_OK;
}

static int g3_launch(int mode, const void* xp, int x_cs, int x_coff, const void* wp, const float* bias, float* y, int y_cs, int y_coff,
                     int y_cw, void* y16, int y16_cs, int y16_coff, float* stats, int stat_cp, int N, long long HW, int W, int H2,
                     int W2, int py0, int px0, int cup, int K_pad, int Ncols, int Ncols_pad, int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(xp && wp && (y || y16), "gemm_bf16v3: null pointer");
  HPRI_REQUIRE(N > 0 && HW > 0 && HW < (1ll << 30), "gemm_bf16v3: bad row counts");
  HPRI_REQUIRE(K_pad > 0 && K_pad % 32 == 0, "gemm_bf16v3: K_pad must be a positive multiple of 32");
  HPRI_REQUIRE(Ncols > 0 && Ncols_pad % 64 == 0 && Ncols <= Ncols_pad, "gemm_bf16v3: Ncols_pad must be a multiple of 64 >= Ncols");
  HPRI_REQUIRE(x_cs % 8 == 0 && x_coff % 8 == 0 && x_cs > 0 && x_cs <= 8192, "gemm_bf16v3: plane row stride / offset must be multiples of 8 (stride <= 8192)");
  HPRI_REQUIRE(((uintptr_t)xp & 15) == 0 && ((uintptr_t)wp & 15) == 0, "gemm_bf16v3: pointers must be 16-byte aligned");
  HPRI_REQUIRE((long long)(K_pad / 32) * Ncols_pad * 64 < 0x7FFFFF00ll, "gemm_bf16v3: packed weights exceed 2 GiB");
  GemmV3Args a;
  a.xp = reinterpret_cast<const h16_t*>(xp); a.x_cs = x_cs; a.x_coff = x_coff;
  a.wp = reinterpret_cast<const h16_t*>(wp); a.bias = bias;
  a.y = y; a.y_cs = y_cs; a.y_coff = y_coff; a.y_cw = y_cw;
  a.y16 = reinterpret_cast<h16_t*>(y16); a.y16_cs = y16_cs; a.y16_coff = y16_coff;
  a.stats = reinterpret_cast<float4*>(stats); a.stat_cp = stat_cp;
  a.N = N; a.HW = (int)HW; a.W = W; a.H2 = H2; a.W2 = W2; a.py0 = py0; a.px0 = px0; a.cup = cup;
  a.nchunks = K_pad / 32; a.Ncols = Ncols; a.Ncols_pad = Ncols_pad;
  a.accumulate = accumulate & 1; a.relu = (accumulate >> 1) & 1;
  if (y != nullptr)
    HPRI_REQUIRE(y_cs % 4 == 0 && y_coff % 4 == 0 && ((uintptr_t)y & 15) == 0, "gemm_bf16v3: the fp32 output view must be float4-aligned");
  if (y16 != nullptr)
    HPRI_REQUIRE(y16_cs % 4 == 0 && y16_coff % 4 == 0 && ((uintptr_t)y16 & 7) == 0, "gemm_bf16v3: the bf16 output view must be 8-byte aligned");
  a.acc16 = 0;
  if (a.accumulate && y == nullptr) {
    // accumulate with a bf16 view only: the plain row GEMM adds into the bf16 rows (a gradient with two producers stored as bf16)
    HPRI_REQUIRE(mode == 0 && y16 != nullptr && !a.relu && bias == nullptr, "gemm_bf16v3: accumulating into bf16 rows is the plain data-gradient form (mode 0, no bias, no ReLU)");
    a.acc16 = 1; a.accumulate = 0;
  }
  HPRI_REQUIRE(!(a.accumulate && mode == 1), "gemm_bf16v3: the depth-to-space form does not accumulate");
  if (mode == 0 || mode == 2) {
    HPRI_REQUIRE(y_cw % 4 == 0 && y_cw >= Ncols && y_cw <= ((Ncols_pad + G3_BN - 1) / G3_BN) * G3_BN, "gemm_bf16v3: written width must be a multiple of 4 in [Ncols, column blocks]");
    if (y != nullptr) HPRI_REQUIRE(y_cw + y_coff <= y_cs, "gemm_bf16v3: written width exceeds the fp32 row stride");
    if (y16 != nullptr) HPRI_REQUIRE(y_cw + y16_coff <= y16_cs, "gemm_bf16v3: written width exceeds the bf16 row stride");
    HPRI_REQUIRE(stats == nullptr || (mode == 0 && stat_cp >= Ncols && !a.accumulate), "gemm_bf16v3: statistics only for plain, non-accumulating launches");
  }
  if (mode == 0) HPRI_REQUIRE(x_coff + K_pad <= x_cs, "gemm_bf16v3: plane rows narrower than K_pad");
  if (mode == 1 || mode == 2) {
    HPRI_REQUIRE(W > 0 && HW % W == 0 && cup > 0 && H2 > 0 && W2 > 0 && py0 >= 0 && px0 >= 0 && py0 + 2 * (HW / W) <= H2 && px0 + 2 * W <= W2,
                 "gemm_bf16v3: transposed-convolution geometry out of range");
    HPRI_REQUIRE((long long)N * H2 * W2 < (1ll << 31), "gemm_bf16v3: too many output pixels");
  }
  if (mode == 1) {
    HPRI_REQUIRE(cup % 16 == 0 && Ncols == 4 * cup && cup < (1 << 20), "gemm_bf16v3: depth-to-space needs Cup % 16 == 0 and 4*Cup columns");
    HPRI_REQUIRE(x_coff + K_pad <= x_cs && stats == nullptr, "gemm_bf16v3: plane rows narrower than K_pad");
    if (y != nullptr) HPRI_REQUIRE(cup + y_coff <= y_cs, "gemm_bf16v3: Cup channels exceed the fp32 row stride");
    if (y16 != nullptr) HPRI_REQUIRE(cup + y16_coff <= y16_cs, "gemm_bf16v3: Cup channels exceed the bf16 row stride");
  }
  if (mode == 2) {
    HPRI_REQUIRE(cup % 32 == 0 && K_pad == 4 * cup && x_coff + cup <= x_cs, "gemm_bf16v3: space-to-depth needs Cup % 32 == 0 and K = 4*Cup");
    HPRI_REQUIRE((long long)H2 * W2 * x_cs * 2 < 0x7FFFFF00ll, "gemm_bf16v3: one image of the gradient planes exceeds 2 GiB (32-bit DMA offsets)");
  }
  a.tiles_img = (int)((HW + 255) / 256); a.ntiles = N * a.tiles_img; a.nb_count = (Ncols_pad + G3_BN - 1) / G3_BN;
  const long long items = (long long)a.ntiles * a.nb_count;
  HPRI_REQUIRE(items < (1ll << 28), "gemm_bf16v3: too many work items");
  a.per_xcd = (int)((items + 7) / 8);
  a.ncu = hpri_cu_count(); a.stagger_cycles = G3_STAGGER_CYCLES;
  a.queue = a.queue_clear = nullptr;
  int nloc = (2 * a.ncu) / 8;
  if (nloc < 1) nloc = 1;
  if (nloc > a.per_xcd) nloc = a.per_xcd;
  dim3 grid((unsigned)(nloc * 8));
  if (a.per_xcd >= 2 * nloc) { const HpriQueueHalves qh = hpri_item_queue_take(stream); a.queue = qh.use; a.queue_clear = qh.clear; }
  if (mode == 0) hipLaunchKernelGGL(gemm_bf16v3_kernel<0>, grid, dim3(256), 0, stream, a);
  else if (mode == 1) hipLaunchKernelGGL(gemm_bf16v3_kernel<1>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(gemm_bf16v3_kernel<2>, grid, dim3(256), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// y[p, n] (+)= sum_k x[p, k] * w[n, k] + bias[n] over bf16 planes x (P = N * HW rows of x_cs elements, K_pad of them read from
// x_coff on; pad channels must hold zeros or meet zero weights) and the packed weights of hpri_pack_weight_bf16 (T = 1, modes 0 / 1).
// Outputs: fp32 view y and / or bf16 view y16 (either may be nullptr), y_cw columns written (pad columns: exact zeros + bias 0);
// `accumulate` bit 0: add to y, bit 1: ReLU.  stats: per 256-row tile (hpri_gemm_bf16v3_plan; tiles do not straddle images)
// records of stat_cp columns, as the 3x3 kernels leave them for hpri_bn_finalize.
extern "C" int hpri_gemm_bf16v3(const void* xp, int x_cs, int x_coff, const void* wp, const float* bias, float* y, int y_cs, int y_coff,
                                void* y16, int y16_cs, int y16_coff, float* stats, int stat_cp, int N, long long HW, int K_pad, int Ncols,
                                int Ncols_pad, int y_cw, int accumulate, hipStream_t stream) {
  return g3_launch(0, xp, x_cs, x_coff, wp, bias, y, y_cs, y_coff, y_cw, y16, y16_cs, y16_coff, stats, stat_cp, N, HW, 0, 0, 0, 0, 0, 0,
                   K_pad, Ncols, Ncols_pad, accumulate, stream);
}

// ConvTranspose2d(k = 2, s = 2) forward from bf16 planes x [N, H, W, Cin]: column tap*Cup + co of the GEMM (pack mode 2) goes to
// output pixel (py0 + 2y + tap/2, px0 + 2x + tap%2), channel co of the [N, H2, W2] views y (fp32) and / or y16 (bf16).
extern "C" int hpri_convt_fwd_bf16v3(const void* xp, int x_cs, int x_coff, const void* wp, const float* bias, float* y, int y_cs, int y_coff,
                                     void* y16, int y16_cs, int y16_coff, int N, int H, int W, int K_pad, int Cup, int Ncols_pad, int H2,
                                     int W2, int py0, int px0, hipStream_t stream) {
  return g3_launch(1, xp, x_cs, x_coff, wp, bias, y, y_cs, y_coff, 0, y16, y16_cs, y16_coff, nullptr, 0, N, (long long)H * W, W, H2, W2, py0,
                   px0, Cup, K_pad, 4 * Cup, Ncols_pad, 0, stream);
}

// Its data gradient: dx[n, y, x, ci] (+)= sum over tap, co of dy[n, py0 + 2y + tap/2, px0 + 2x + tap%2, co] * w[ci, co, tap] with dy as
// bf16 planes [N, H2, W2, dy_cs] (Cup channels from dy_coff on) and the mode-3 pack (K = 4*Cup); Cup must be a multiple of 32.
extern "C" int hpri_convt_dgrad_bf16v3(const void* dyp, int dy_cs, int dy_coff, const void* wp, float* dx, int dx_cs, int dx_coff, int N,
                                       int H, int W, int Cup, int Cin, int Cin_pad, int dx_cw, int H2, int W2, int py0, int px0,
                                       int accumulate, hipStream_t stream) {
  return g3_launch(2, dyp, dy_cs, dy_coff, wp, nullptr, dx, dx_cs, dx_coff, dx_cw, nullptr, 0, 0, nullptr, 0, N, (long long)H * W, W, H2, W2,
                   py0, px0, Cup, 4 * Cup, Cin, Cin_pad, accumulate & 1, stream);
}

// ... written as bf16 rows (no accumulate): the input of a decoder stage has this one reader of its gradient, the BatchNorm backward
// of the stage that produced it, which reads bf16 (hpri_bn_relu_bwd_x16_dy16)
extern "C" int hpri_convt_dgrad_bf16v3_y16(const void* dyp, int dy_cs, int dy_coff, const void* wp, void* dx16, int dx_cs, int dx_coff, int N,
                                           int H, int W, int Cup, int Cin, int Cin_pad, int dx_cw, int H2, int W2, int py0, int px0,
                                           hipStream_t stream) {
  HPRI_REQUIRE(dx16 != nullptr, "convt_dgrad_bf16v3_y16: null pointer");
  return g3_launch(2, dyp, dy_cs, dy_coff, wp, nullptr, nullptr, 0, 0, dx_cw, dx16, dx_cs, dx_coff, nullptr, 0, N, (long long)H * W, W, H2, W2,
                   py0, px0, Cup, 4 * Cup, Cin, Cin_pad, 0, stream);
}
