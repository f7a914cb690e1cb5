// fp32 GEMM for the 1x1 forms of the exact-fp32 mode, second form (round 4): ConvTranspose2d(k = 2, s = 2) forward
// (reference model_parts.py:63-64; models.py:198: one GEMM Cin -> 4*Cup per input pixel whose epilogue scatters the 2x2 patches)
// and its data gradient (K = 4*Cup gathered from the four parities), plus the plain row GEMM (nn.Linear / Conv2d(k=1) forward
// and data gradient without statistics).  These forms ran conv_fwd_kernel<1, 2, 2, ...> until now: 85-99 TFLOP/s of the 157.3 the
// fp32 MFMA has -- their K is short (128 ... 1024), so a workgroup spent a quarter of its life in a prologue that waited for its
// first operands and in an epilogue of 64 scalar stores per lane with a 64-bit index chain each, none of it under MFMAs.
//
// Structure = gemm_bf16v3.hip's, with fp32 operands and v_mfma_f32_32x32x2_f32:
//   workgroup  256 threads = 4 waves, 256 pixels x 128 columns, 73 KB of LDS: two independent workgroups per CU, persistent
//              (2 x CUs workgroups, fixed item lists): one workgroup's prologue and epilogue run under the other's MFMAs, and
//              the next item's first two stages are in flight while the current item is written out.
//   MFMA       weights as the A operand: D[channel][pixel], a lane's accumulator registers 4 q .. 4 q + 3 are four CONSECUTIVE
//              columns of one pixel -> 16-byte stores, no transposition.  Wave tile 64 px x 128 columns = 2 x 4 tiles of 32 x 32
//              (128 accumulator registers), 64 MFMAs of 64 cycles per 16-deep stage: 12 ds_read_b128 per stage.
//   stages     16 k per stage: 256 px x 64 B + 128 columns x 64 B = 24 KB, triple-buffered, both operands by LDS-DMA
//              (buffer_load ... lds), one counted vmcnt + one barrier per stage; the four 16-byte k-slots of a row are XOR-swizzled
//              with (row >> 2) & 3 through the DMA source address: conflict-free ds_read_b128 for the 32 x 32 x 2 lane map (lanes
//              0-31 = rows, lane half = k half; checked over the four lane groups of a b128 read).
//   k order    a lane half reads two slots (8 consecutive k); MFMA step j multiplies k = j of the lower half with k = 8 + j of
//              the upper half: the result is an fp32 fma chain like the direct kernel's, in another (fixed) order.
#include "common.h"

#define F2_BN 128                               // columns per workgroup
#define F2_KC 16                                // k per stage (floats): 64-byte rows
#define F2_A_BYTES (256 * 64)
#define F2_B_BYTES (F2_BN * 64)
#define F2_STAGE_BYTES (F2_A_BYTES + F2_B_BYTES)

struct GemmF2Args {
  const float* x; int x_cs, x_coff;           // A rows: floats per pixel row (multiple of 4), first channel (multiple of 4)
  const float* wp;                            // packed [chunk16][Ncols_pad][16] (hpri_pack_weight_f32k16)
  const float* bias;                          // per output column (mode 1: per Cup channel), or nullptr
  float* y; int y_cs, y_coff, y_cw;
  int N, HW;                                  // images, GEMM rows per image (modes 1 / 2: H * W of the low-resolution grid)
  int W, H2, W2, py0, px0, cup;               // transposed-convolution geometry (modes 1, 2)
  int nchunks, Ncols, Ncols_pad;              // k-chunks of 16, output columns, packed column count (multiple of 64)
  int accumulate;
  int tiles_img, ntiles, nb_count, per_xcd;
  int ncu, stagger_cycles;
  unsigned *queue, *queue_clear;             // item counters of this launch and the half it zeroes for the next one (common.h), or nullptr = fixed item lists
};

struct F2Tile { int img, p0, valid, nb, bx; };

// MODE 0: rows = pixels of x, output row = the same pixel.  MODE 1: the same rows, output scattered depth-to-space.
// MODE 2: row p of the low-resolution grid gathers its K = 4*cup values from the four parities of the high-resolution tensor.
template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_f32v2_kernel(GemmF2Args a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * F2_STAGE_BYTES + 2 * F2_BN * 4 + 16];
  float* bias_lds = reinterpret_cast<float*>(smem + 3 * F2_STAGE_BYTES);     // [2 slots][F2_BN]
  int* next_lds = reinterpret_cast<int*>(smem + 3 * F2_STAGE_BYTES + 2 * F2_BN * 4);      // [2 slots]: the next item's index in the band (item queue)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int xcd = blockIdx.x & 7, nloc = (int)(gridDim.x >> 3);
  const int items_all = a.ntiles * a.nb_count;
  auto tile_of = [&](int k, F2Tile& t) -> bool {
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
  unsigned aoff[4], goff[2];
  hpri_rsrc_t rs_a = HPRI_MAKE_RSRC(a.x, 0x7FFFFF00);
  const hpri_rsrc_t rs_b = HPRI_MAKE_RSRC(a.wp, 0x7FFFFF00);
  const int chunk_bytes = a.Ncols_pad * 64;
  const int cpt = MODE == 2 ? (a.cup >> 4) : 1;            // mode 2: 16-channel chunks per parity
  auto setup_dma = [&](const F2Tile& t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pix = (q * 4 + wave) * 16 + (lane >> 2);
      const unsigned sw = (unsigned)(((lane & 3) ^ ((pix >> 2) & 3)) << 2);     // logical k-slot of this lane's physical slot, in floats
      unsigned off = OOB;
      if (pix < t.valid) {
        if (MODE == 2) {
          const int p = t.p0 + pix, yy = p / a.W, xx = p - yy * a.W;
          off = (unsigned)(((a.py0 + 2 * yy) * a.W2 + a.px0 + 2 * xx) * a.x_cs + (int)sw) * 4u;
        } else {
          off = (unsigned)(pix * a.x_cs + (int)sw) * 4u;
        }
      }
      aoff[q] = off;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = (q * 4 + wave) * 16 + (lane >> 2);
      const int n = min(t.nb * F2_BN + row, a.Ncols_pad - 1);      // rows beyond the pack: any packed row (their columns are never written)
      goff[q] = (unsigned)(n * 16 + (((lane & 3) ^ ((row >> 2) & 3)) << 2)) * 4u;
    }
    // base of the rows this item reads: modes 0 / 1 the tile's first pixel, mode 2 the image of the high-resolution tensor (host: one
    // image < 2 GiB); wave-uniform by construction, and said so (cdna_hip_programming.md T20)
    const size_t row0 = MODE == 2 ? (size_t)t.img * a.H2 * a.W2 : (size_t)t.img * a.HW + t.p0;
    const unsigned long long pb = (unsigned long long)(uintptr_t)(a.x + row0 * a.x_cs + a.x_coff);
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pb), phi = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32));
    (void)plo; (void)phi;
    rs_a = HPRI_MAKE_RSRC((((unsigned long long)phi << 32) | plo), 0x7FFFFF00);
  };
  (void)goff; (void)chunk_bytes; (void)rs_b; (void)cpt;
  auto a_soff = [&](int c) -> int {             // scalar byte offset of k-chunk c inside a pixel row of x
    if (MODE == 2) { const int tap = c / cpt, cc = c - tap * cpt; return (((tap >> 1) * a.W2 + (tap & 1)) * a.x_cs + cc * 16) * 4; }
    return c * 64;
  };
#define F2_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define F2_BARRIER()                         \
  __builtin_amdgcn_sched_barrier(0);         \
  __builtin_amdgcn_s_barrier();              \
  __builtin_amdgcn_sched_barrier(0)
#define F2_DMA_A_AT(bo_, c_, q_) HPRI_LDS_DMA16(rs_a, smem + (bo_) + ((q_) * 4 + wave) * 1024, aoff[q_], a_soff(c_))
#define F2_DMA_B_AT(bo_, c_, q_) HPRI_LDS_DMA16(rs_b, smem + (bo_) + F2_A_BYTES + ((q_) * 4 + wave) * 1024, goff[q_], (c_) * chunk_bytes)
#define F2_ISSUE(buf_, c_)                                                               \
  {                                                                                      \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) F2_DMA_B_AT((buf_) * F2_STAGE_BYTES, c_, q); \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) F2_DMA_A_AT((buf_) * F2_STAGE_BYTES, c_, q); \
  }

  // fragment addresses: row (column row ct*32 + li resp. pixel wave*64 + pt*32 + li), logical k-slot 2 lh + g
  const int fofs0 = li * 64 + (((2 * lh + 0) ^ ((li >> 2) & 3)) << 4);
  const int fofs1 = li * 64 + (((2 * lh + 1) ^ ((li >> 2) & 3)) << 4);
  const int S = a.nchunks;

  F2Tile cur, nxt;
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
  float bias_next = 0.f;
  auto load_bias = [&](const F2Tile& t) {
    if (tid < F2_BN) {
      const int col = t.nb * F2_BN + tid;
      const int ch = MODE == 1 ? col % a.cup : col;
      bias_next = (a.bias != nullptr && col < a.Ncols) ? a.bias[ch] : 0.f;
    }
  };
  load_bias(cur);
  F2_ISSUE(0, 0)
  if (S > 1) F2_ISSUE(1, 1)
  int slot = 0;

  while (have) {
    if (tid < F2_BN) bias_lds[slot * F2_BN + tid] = bias_next;      // visible behind the first stage's barrier
    if (q != nullptr && tid == 0) next_lds[slot] = nstatic + (int)pend;      // the next item's ticket, parked (conv_bf16v3.hip)

    f32x16 acc[4][2];                            // [column tile ct][pixel tile pt]
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;

    // Stage s multiplies buffer s % 3; at its top the wave's own pieces of stage s have landed when at most the six of stage s + 1
    // are still in flight; behind the barrier every wave has left stage s - 1, whose buffer then takes stage s + 2.  ONE loop with
    // the buffer offset in a scalar register and ONE straight-line MFMA sequence (gemm_bf16v3.hip: anything else made hipcc copy
    // the accumulators).
    int bo = 0;
    for (int s = 0; s < S; ++s) {
      if (s + 1 < S) F2_WAIT_VM(6); else F2_WAIT_VM(0);
      F2_BARRIER();
      const unsigned char* wb_ = smem + bo + F2_A_BYTES;
      const unsigned char* xb_ = smem + bo + wave * 4096;
      f32x4 wa[4][2], xb[2][2];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        wa[ct][0] = *reinterpret_cast<const f32x4*>(wb_ + ct * 2048 + fofs0);
        wa[ct][1] = *reinterpret_cast<const f32x4*>(wb_ + ct * 2048 + fofs1);
      }
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        xb[pt][0] = *reinterpret_cast<const f32x4*>(xb_ + pt * 2048 + fofs0);
        xb[pt][1] = *reinterpret_cast<const f32x4*>(xb_ + pt * 2048 + fofs1);
      }
      const bool more_ = s + 2 < S;
      const int bo2 = bo == 0 ? 2 * F2_STAGE_BYTES : bo - F2_STAGE_BYTES;          // buffer (s + 2) % 3
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
              acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ct][g][j], xb[pt][g][j], acc[ct][pt], 0, 0, 0);
          // the six DMA pieces of stage s + 2, one behind each of the first six k-steps' MFMA groups
          const int step = g * 4 + j;
          if (more_ && step < 2) { F2_DMA_B_AT(bo2, s + 2, step); }
          if (more_ && step >= 2 && step < 6) { F2_DMA_A_AT(bo2, s + 2, step - 2); }
        }
      __builtin_amdgcn_s_setprio(0);
      bo = bo == 2 * F2_STAGE_BYTES ? 0 : bo + F2_STAGE_BYTES;
    }
    F2_BARRIER();                                // every wave has left the main loop: all three buffers are free
    if (q != nullptr) k = __builtin_amdgcn_readfirstlane(next_lds[slot]);
    else k += nloc;
    have = tile_of(k, nxt);
    if (have) {
      setup_dma(nxt);
      F2_ISSUE(0, 0)
      if (S > 1) F2_ISSUE(1, 1)
      load_bias(nxt);
      if (q != nullptr && tid == 0) pend = hpri_q_draw(q, xcd, 1u);      // the ticket of the item after that one
    }

    // ------------------------------- epilogue -------------------------------
    // acc[ct][pt][4 q + e]: pixel wave*64 + pt*32 + li of the tile, column nb*128 + ct*32 + 8 q + 4 lh + e
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int pix = wave * 64 + pt * 32 + li;
      const bool ok = pix < cur.valid;
      const int p = cur.p0 + min(pix, cur.valid - 1);
      long long orow;                            // output pixel row (mode 1: of parity (0, 0))
      if (MODE == 1) {
        const int yy = p / a.W, xx = p - yy * a.W;
        orow = ((long long)cur.img * a.H2 + a.py0 + 2 * yy) * a.W2 + a.px0 + 2 * xx;
      } else {
        orow = (long long)cur.img * a.HW + p;
      }
      // all loads of an accumulating launch in front of this pixel tile's stores (a wait for a load behind a store would wait for
      // the store as well), in two halves of eight quads: 32 registers of old values at a time next to the 128 accumulators
      if (MODE != 1 && a.accumulate) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          f32x4 old[2][4];
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int col = cur.nb * F2_BN + (2 * hh + c2) * 32 + 8 * q + 4 * lh;
              old[c2][q] = (ok && col < a.y_cw) ? *reinterpret_cast<const f32x4*>(a.y + orow * a.y_cs + a.y_coff + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[2 * hh + c2][pt][4 * q + e] += old[c2][q][e];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cl = ct * 32 + 8 * q + 4 * lh;                  // column inside the block
          const int col = cur.nb * F2_BN + cl;
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_lds + slot * F2_BN + cl);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[ct][pt][4 * q + e] + b4[e];
          if (MODE == 1) {
            if (ok && col < a.Ncols) {
              const int tap = col / a.cup, co = col - tap * a.cup;
              const long long rr = orow + (long long)((tap >> 1) * a.W2 + (tap & 1));
              *reinterpret_cast<f32x4*>(a.y + rr * a.y_cs + a.y_coff + co) = v;
            }
          } else {
            if (ok && col < a.y_cw) {
#pragma unroll
              for (int e = 0; e < 4; ++e) if (col + e >= a.Ncols) v[e] = 0.f;
              *reinterpret_cast<f32x4*>(a.y + orow * a.y_cs + a.y_coff + col) = v;
            }
          }
        }
    }
    cur = nxt;
    slot ^= 1;
  }
#undef F2_ISSUE
#undef F2_DMA_A_AT
#undef F2_DMA_B_AT
#undef F2_WAIT_VM
#undef F2_BARRIER
}

// ---- packed weights [chunk16][Ncols_pad][16]: the four modes of hpri_pack_weight with k innermost in runs of 16 ---------------
__global__ void pack_weight_f32k16_kernel(const float* __restrict__ w, float* __restrict__ wp, int mode, int K, int Ncols,
                                          int Ncols_pad, int chunks, int Cup, int src_d1) {
  const size_t total = (size_t)chunks * Ncols_pad * 16;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int kk = (int)(idx & 15);
    const size_t r = idx >> 4;
    const int col = (int)(r % Ncols_pad), chunk = (int)(r / Ncols_pad);
    const int k = chunk * 16 + kk;
    float v = 0.f;
    if (k < K && col < Ncols) {
      if (mode == 0) v = w[(size_t)col * src_d1 + k];                           // Linear / 1x1 forward: W[n = col][c = k]
      else if (mode == 1) v = w[(size_t)k * src_d1 + col];                      // its data gradient: W[n = k][c = col]
      else if (mode == 2) { const int tap = col / Cup, co = col - tap * Cup; v = w[((size_t)k * Cup + co) * 4 + tap]; }
      else { const int tap = k / Cup, co = k - tap * Cup; v = w[((size_t)col * Cup + co) * 4 + tap]; }
    }
    wp[idx] = v;
  }
}

extern "C" size_t hpri_packed_weight_f32k16_floats(int K, int Ncols_pad) { return (size_t)hpri_cdiv(K, 16) * 16 * Ncols_pad; }

extern "C" int hpri_pack_weight_f32k16(const float* w, float* wp, int mode, int K, int Ncols, int Ncols_pad, int Cup, int src_d1,
                                       hipStream_t stream) {
  HPRI_REQUIRE(w && wp, "pack_weight_f32k16: null pointer");
  HPRI_REQUIRE(mode >= 0 && mode <= 3 && K > 0 && Ncols > 0 && Ncols_pad >= Ncols && Ncols_pad % 64 == 0, "pack_weight_f32k16: bad arguments");
  HPRI_REQUIRE(mode < 2 || Cup > 0, "pack_weight_f32k16: the transposed-convolution modes need Cup");
  const int chunks = hpri_cdiv(K, 16);
  const size_t total = (size_t)chunks * Ncols_pad * 16;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weight_f32k16_kernel, dim3(blocks), dim3(256), 0, stream, w, wp, mode, K, Ncols, Ncols_pad, chunks, Cup, src_d1);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

#define F2_STAGGER_CYCLES 8000

static int f2_launch(int mode, const float* x, int x_cs, int x_coff, const float* wp, const float* bias, float* y, int y_cs, int y_coff,
                     int y_cw, int N, long long HW, int W, int H2, int W2, int py0, int px0, int cup, int K_pad, int Ncols,
                     int Ncols_pad, int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(x && wp && y, "gemm_f32v2: null pointer");
  HPRI_REQUIRE(N > 0 && HW > 0 && HW < (1ll << 30), "gemm_f32v2: bad row counts");
  HPRI_REQUIRE(K_pad > 0 && K_pad % 16 == 0, "gemm_f32v2: K_pad must be a positive multiple of 16");
  HPRI_REQUIRE(Ncols > 0 && Ncols_pad % 64 == 0 && Ncols <= Ncols_pad, "gemm_f32v2: Ncols_pad must be a multiple of 64 >= Ncols");
  HPRI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0 && x_cs > 0 && x_cs <= 16384, "gemm_f32v2: row stride / offset must be multiples of 4 (stride <= 16384)");
  HPRI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)wp & 15) == 0 && ((uintptr_t)y & 15) == 0, "gemm_f32v2: pointers must be 16-byte aligned");
  HPRI_REQUIRE(y_cs % 4 == 0 && y_coff % 4 == 0, "gemm_f32v2: the output view must be float4-aligned");
  HPRI_REQUIRE((long long)(K_pad / 16) * Ncols_pad * 64 < 0x7FFFFF00ll, "gemm_f32v2: packed weights exceed 2 GiB");
  GemmF2Args a;
  a.x = x; a.x_cs = x_cs; a.x_coff = x_coff; a.wp = wp; a.bias = bias;
  a.y = y; a.y_cs = y_cs; a.y_coff = y_coff; a.y_cw = y_cw;
  a.N = N; a.HW = (int)HW; a.W = W; a.H2 = H2; a.W2 = W2; a.py0 = py0; a.px0 = px0; a.cup = cup;
  a.nchunks = K_pad / 16; a.Ncols = Ncols; a.Ncols_pad = Ncols_pad; a.accumulate = accumulate & 1;
  if (mode == 0 || mode == 2) {
    HPRI_REQUIRE(y_cw % 4 == 0 && y_cw >= Ncols && y_cw + y_coff <= y_cs, "gemm_f32v2: written width must be a multiple of 4 >= Ncols inside the row stride");
    HPRI_REQUIRE(y_cw <= ((Ncols_pad + F2_BN - 1) / F2_BN) * F2_BN, "gemm_f32v2: written width exceeds the column blocks");
  }
  if (mode == 0) HPRI_REQUIRE(x_coff + K_pad <= x_cs, "gemm_f32v2: rows narrower than K_pad");
  if (mode == 1 || mode == 2) {
    HPRI_REQUIRE(W > 0 && HW % W == 0 && cup > 0 && H2 > 0 && W2 > 0 && py0 >= 0 && px0 >= 0 && py0 + 2 * (HW / W) <= H2 && px0 + 2 * W <= W2,
                 "gemm_f32v2: transposed-convolution geometry out of range");
    HPRI_REQUIRE((long long)N * H2 * W2 < (1ll << 31), "gemm_f32v2: too many output pixels");
  }
  if (mode == 1) {
    HPRI_REQUIRE(cup % 4 == 0 && Ncols == 4 * cup && !a.accumulate, "gemm_f32v2: depth-to-space needs Cup % 4 == 0, 4*Cup columns and no accumulate");
    HPRI_REQUIRE(x_coff + K_pad <= x_cs && cup + y_coff <= y_cs, "gemm_f32v2: rows narrower than K_pad / Cup channels exceed the output stride");
  }
  if (mode == 2) {
    HPRI_REQUIRE(cup % 16 == 0 && K_pad == 4 * cup && x_coff + cup <= x_cs, "gemm_f32v2: space-to-depth needs Cup % 16 == 0 and K = 4*Cup");
    HPRI_REQUIRE((long long)H2 * W2 * x_cs * 4 < 0x7FFFFF00ll, "gemm_f32v2: one image of the gradient tensor exceeds 2 GiB (32-bit DMA offsets)");
  }
  a.tiles_img = (int)((HW + 255) / 256); a.ntiles = N * a.tiles_img; a.nb_count = (Ncols_pad + F2_BN - 1) / F2_BN;
  const long long items = (long long)a.ntiles * a.nb_count;
  HPRI_REQUIRE(items < (1ll << 28), "gemm_f32v2: too many work items");
  a.per_xcd = (int)((items + 7) / 8);
  a.ncu = hpri_cu_count(); a.stagger_cycles = F2_STAGGER_CYCLES;
  a.queue = a.queue_clear = nullptr;
  int nloc = (2 * a.ncu) / 8;
  if (nloc < 1) nloc = 1;
  if (nloc > a.per_xcd) nloc = a.per_xcd;
  dim3 grid((unsigned)(nloc * 8));
  if (a.per_xcd >= 2 * nloc) { const HpriQueueHalves qh = hpri_item_queue_take(stream); a.queue = qh.use; a.queue_clear = qh.clear; }
  if (mode == 0) hipLaunchKernelGGL(gemm_f32v2_kernel<0>, grid, dim3(256), 0, stream, a);
  else if (mode == 1) hipLaunchKernelGGL(gemm_f32v2_kernel<1>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(gemm_f32v2_kernel<2>, grid, dim3(256), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// y[p, n] (+)= sum_k x[p, k] * w[n, k] + bias[n] in exact fp32 (v_mfma_f32_32x32x2_f32) over fp32 rows x (P = N * HW rows of x_cs
// floats, K_pad of them read from x_coff on; channels [K, K_pad) must hold zeros or meet zero weights) and the packed weights of
// hpri_pack_weight_f32k16 (modes 0 / 1).  y_cw columns are written (columns >= Ncols: zeros); accumulate bit 0: y += result.
extern "C" int hpri_gemm_f32v2(const float* x, int x_cs, int x_coff, const float* wp, const float* bias, float* y, int y_cs, int y_coff,
                               int N, long long HW, int K_pad, int Ncols, int Ncols_pad, int y_cw, int accumulate, hipStream_t stream) {
  return f2_launch(0, x, x_cs, x_coff, wp, bias, y, y_cs, y_coff, y_cw, N, HW, 0, 0, 0, 0, 0, 0, K_pad, Ncols, Ncols_pad, accumulate, stream);
}

// ConvTranspose2d(k = 2, s = 2) forward in exact fp32 from x [N, H, W, Cin]: column tap*Cup + co of the GEMM (pack mode 2) goes to
// output pixel (py0 + 2y + tap/2, px0 + 2x + tap%2), channel co of the [N, H2, W2] view y.
extern "C" int hpri_convt_fwd_f32v2(const float* x, int x_cs, int x_coff, const float* wp, const float* bias, float* y, int y_cs,
                                    int y_coff, int N, int H, int W, int K_pad, int Cup, int Ncols_pad, int H2, int W2, int py0, int px0,
                                    hipStream_t stream) {
  return f2_launch(1, x, x_cs, x_coff, wp, bias, y, y_cs, y_coff, 0, N, (long long)H * W, W, H2, W2, py0, px0, Cup, K_pad, 4 * Cup, Ncols_pad,
                   0, stream);
}

// Its data gradient: dx[n, y, x, ci] (+)= sum over tap, co of dy[n, py0 + 2y + tap/2, px0 + 2x + tap%2, co] * w[ci, co, tap] with dy
// fp32 [N, H2, W2, dy_cs] (Cup channels from dy_coff on, Cup % 16 == 0) and the mode-3 pack (K = 4*Cup).
extern "C" int hpri_convt_dgrad_f32v2(const float* dy, int dy_cs, int dy_coff, const float* wp, float* dx, int dx_cs, int dx_coff, int N,
                                      int H, int W, int Cup, int Cin, int Cin_pad, int dx_cw, int H2, int W2, int py0, int px0,
                                      int accumulate, hipStream_t stream) {
  return f2_launch(2, dy, dy_cs, dy_coff, wp, nullptr, dx, dx_cs, dx_coff, dx_cw, N, (long long)H * W, W, H2, W2, py0, px0, Cup, 4 * Cup, Cin,
                   Cin_pad, accumulate & 1, stream);
}
