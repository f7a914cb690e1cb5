// 3x3 implicit-GEMM convolution on bf16 activation PLANES, third form (precision mode "bf16"; forward, and data gradient with the
// mode-1 pack; reference model_parts.py:22,25; models.py:169,177 and their autograd) -- conv_bf16v2.hip rebuilt on the occupancy
// recipe that worked for the fp32 Winograd kernel (conv_wino4.hip):
//
//   workgroup  256 threads = 4 waves, 256 pixels x 64 output channels, 72 KB of LDS: TWO independent workgroups per CU (one wave
//              of each on every SIMD).  v2 ran one 8-wave workgroup per CU: every barrier and the whole store + statistics
//              epilogue (~10 k cycles per item) stalled the CU's matrix pipes; here they run under the partner workgroup's
//              MFMAs.  Persistent (2 x CUs workgroups walking fixed item lists): a first, non-persistent form paid ~9 k cycles
//              of dispatch gap and ~5 k of exposed prologue per item (tools/v3_stamps.py); now the next item's first halo and
//              weight stage are in flight while the current item's results are written out.  Workgroup id mod 8 walks one
//              XCD's band of the image in raster order, so halos shared by neighbouring tiles meet in that XCD's L2.
//   MFMA       v_mfma_f32_16x16x32_bf16 (same cycles per flop as 32x32x16, but the chip holds a higher clock on it:
//              MI355X_MICROARCH.md, DVFS give-back 7), wave tile 64 px x 64 ch = 4 x 4 accumulator tiles (64 VGPRs).  The
//              WEIGHTS are the A operand and the pixels the B operand: D = W X^T has the channel on the register index and the
//              pixel on the lane, so a lane's four accumulator registers are four CONSECUTIVE channels of one pixel and go out
//              as one 16-byte store -- no transposition through LDS in the epilogue at all.
//   A (input)  halo of (TH+2) x (TW+2) pixels x 32 channels, 64-byte pixel rows, by LDS-DMA (buffer_load ... lds, out-of-image
//              pixels zero-filled by the descriptor's range check); the four 16-byte k-slots of a pixel are XOR-swizzled with
//              2*((pixel>>2)&1) through the DMA source address: conflict-free ds_read_b128 for the 16x16x32 fragment lane
//              groups at every tap offset (brute-forced over all halo positions; v2's (pixel>>2)&3 is 2-way conflicted for
//              this lane map).  Double-buffered per 32-channel chunk, the next chunk's halo issued in two halves.
//   B (weights) one kernel row (3 taps x 64 ch x 32 k = 12 KB) per stage, same swizzle, double-buffered, one stage ahead.
//   pipeline   one counted vmcnt + ONE barrier per stage of 48 MFMAs per wave; DMA pieces issued between the MFMAs.
//   statistics per-tile BatchNorm partials (exact two-pass per wave, Chan merge of the four waves) with DPP row reductions:
//              the 16 pixels of an accumulator column live on the 16 lanes of a DPP row.
// Epilogue contract (bias, ReLU, accumulate, split-K raw slabs, statistics records) as conv_bf16v2.hip.
#include "common.h"

typedef h16_t bf16x8 __attribute__((ext_vector_type(8)));

#define V3_MAXSEG 4
#define V3_A_PIECES 24                       // 1-KB DMA pieces per halo buffer: 384 pixel slots (largest halo: 10 x 34 = 340)
#define V3_A_BYTES (V3_A_PIECES * 1024)
#define V3_B_BYTES (12 * 1024)               // 3 taps x 64 channels x 32 k x 2 B

struct ConvV3Args {
  const h16_t* xp; int x_cs, x_coff;        // activation plane: elements per pixel (multiple of 32), first channel (multiple of 8)
  const h16_t* wp;                          // packed weights [chunk][tap][Cout_pad][32] (hpri_pack_weight_bf16)
  const float* bias;
  float* y; int y_cs, y_coff;
  float4* stats;
  int N, H, W, Cin_pad, Cout, Cout_pad, y_cw, accumulate, relu;
  int y16;                                   // the output view is bf16 (y points at h16_t, y_cs / y_coff in elements): the pre-BN tensor of
                                             // the bf16 mode at 2 bytes per element (statistics still from the fp32 sums)
  int ksplit; float* ws;
  int nseg, tiles_img, ntiles, nb_count, per_xcd, nb_major;
  int seg_twl[V3_MAXSEG], seg_xbeg[V3_MAXSEG], seg_ntx[V3_MAXSEG], seg_first[V3_MAXSEG];
  int ncu, stagger_cycles;                   // compute units of the device; one-off delay of each CU's second occupant
  unsigned *queue, *queue_clear;             // item counters of this launch and the half it zeroes for the next one (common.h), or nullptr = fixed item lists
  // conv_bf16v3_kernel<true> (a data-gradient launch that writes the ONLY contribution to dL/dx, x = ReLU(BN(bn_x))): the epilogue
  // also reads the pre-BN tensor bn_x (bf16, elements per pixel bn_x_cs, first channel bn_x_coff, bn_cw readable channels) at its
  // output positions and leaves per-tile partial sums of that BatchNorm's backward, bn_part[stat tile][2][bn_cpart] =
  // (sum g*[y>0], sum g*[y>0]*xhat): the stage that produced x then skips its two reduction sweeps (hpri_bn_relu_bwd_fused)
  // second output: channels [y2_c0, y2_c0 + y2_cw) (whole 64-channel blocks) of the result ALSO (y2_only: ONLY) as bf16 rows of
  // y2_cs elements from y2_coff on -- the gradient of the upsampled half of a decoder concat, which its readers (the transposed
  // convolution's data and weight gradient: gemm_bf16v3.hip, wgrad_bf16v3.hip) stage as planes
  h16_t* y2; int y2_cs, y2_coff, y2_c0, y2_cw, y2_only;
  const h16_t* bn_x; int bn_x_cs, bn_x_coff, bn_cw, bn_relu, bn_cpart;
  const float *bn_mean, *bn_invstd, *bn_scale, *bn_shift;
  float* bn_part;
#ifdef HPRI_STAMPS
  unsigned long long* stamps;                // diagnostic builds only: [workgroup][16] stamps of wave 0 (tools/v3_stamps.py)
#endif
};

#ifdef HPRI_STAMPS
#define V3_STAMP(i_)                                                                                     \
  {                                                                                                      \
    unsigned long long t_;                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if (a.stamps != nullptr && threadIdx.x == 0)                                                         \
      a.stamps[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16 + (i_)] = t_;                           \
  }
#else
#define V3_STAMP(i_)
#endif

// sum over the 16 lanes of a DPP row (= the 16 pixels of one accumulator column group); every lane gets the total
__device__ __forceinline__ float v3_row_sum(float v) {
#define V3_DPP_ADD(ctrl_)                                                                                \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl_, 0xF, 0xF, true))
  V3_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
  V3_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
  V3_DPP_ADD(0x141);   // row_half_mirror
  V3_DPP_ADD(0x140);   // row_mirror
#undef V3_DPP_ADD
  return v;
}

// Geometry of one work item (256-pixel tile x 64-channel block); wave-uniform.
struct V3Tile { int img, y0, x0, xlim, twl, nb, bx; };

template <bool BNRED>
__global__ __launch_bounds__(256, 2) void conv_bf16v3_kernel(ConvV3Args a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * V3_A_BYTES + 2 * V3_B_BYTES + 512 + 16 + 16];
  unsigned char* a_lds = smem;
  unsigned char* b_lds = smem + 2 * V3_A_BYTES;
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * V3_A_BYTES + 2 * V3_B_BYTES);   // [2 slots][64]: a block's biases (0 beyond Cout)
  unsigned* arrive_lds = reinterpret_cast<unsigned*>(smem + 2 * V3_A_BYTES + 2 * V3_B_BYTES + 512);   // waves that have left their statistics record
  int* next_lds = reinterpret_cast<int*>(smem + 2 * V3_A_BYTES + 2 * V3_B_BYTES + 512 + 16);           // [2 slots]: the next item's index in the band (item queue)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;

  // ---- persistent work list: workgroup id mod 8 labels the XCD (round-robin dispatch; speed only), XCD x owns the items
  //      [x*per_xcd, (x+1)*per_xcd) -- channel blocks of a pixel tile back to back, pixel tiles in raster order -- and its
  //      workgroups (two per CU) walk that band interleaved: neighbouring tiles run at the same time on one XCD, so the halo
  //      columns / rows they share meet in that XCD's L2.  Every workgroup runs a fixed list: the grid drains by itself. ----
  const int xcd = blockIdx.x & 7, nloc = (int)(gridDim.x >> 3);
  const int items_all = a.ntiles * a.nb_count;
  auto tile_of = [&](int k, V3Tile& t) -> bool {
    if (k >= a.per_xcd) return false;
    if (a.nb_major) {
      // wide layers (Cout_pad >= 512): XCD x owns the channel blocks nb = x (mod 8) of EVERY pixel tile, so the one or two packed
      // weight slices it needs (Cin_pad x 1152 B each) stay in its L2 and the small input is what gets re-read; banded (below)
      // such a layer streams all of its weights through every XCD once per tile (conv_wino4.hip measured 2-6 x the traffic)
      const int nbx = a.nb_count >> 3;
      t.bx = k / nbx; t.nb = (k - t.bx * nbx) * 8 + xcd;
      if (t.bx >= a.ntiles) return false;
    } else {
      const int item = xcd * a.per_xcd + k;
      if (item >= items_all) return false;
      t.bx = item / a.nb_count; t.nb = item - t.bx * a.nb_count;
    }
    t.img = t.bx / a.tiles_img;
    const int tin = t.bx - t.img * a.tiles_img;
    int seg = 0;
#pragma unroll
    for (int q = 1; q < V3_MAXSEG; ++q)
      if (q < a.nseg && tin >= a.seg_first[q]) seg = q;
    t.twl = a.seg_twl[seg];
    const int TW = 1 << t.twl, TH = 256 >> t.twl;
    const int tt = tin - a.seg_first[seg];
    const int ty = tt / a.seg_ntx[seg], tx = tt - ty * a.seg_ntx[seg];
    t.y0 = ty * TH; t.x0 = a.seg_xbeg[seg] + tx * TW;
    t.xlim = min(a.W, a.seg_xbeg[seg] + a.seg_ntx[seg] * TW);
    return true;
  };

  const int nchunks_all = a.Cin_pad >> 5;
  const int cps = (nchunks_all + a.ksplit - 1) / a.ksplit;
  const int chunk0 = blockIdx.z * cps;
  const int nchunks = min(nchunks_all, chunk0 + cps);
  unsigned* const q = a.queue != nullptr ? a.queue + blockIdx.z * (8 * HPRI_Q_STRIDE) : nullptr;   // this K slice's eight band counters
  if (blockIdx.x == 0 && blockIdx.z == 0) hpri_q_clear(a.queue_clear, tid);      // (for the next launch on this stream)
  if (chunk0 >= nchunks) return;
  const int S0 = chunk0 * 3, S = nchunks * 3;

  // ---- DMA source offsets of the item being LOADED (32-bit per-lane byte offsets against wave-uniform buffer descriptors;
  //      halo pixels outside the image get an offset beyond the descriptor's range: the hardware writes zeros) ----
  constexpr unsigned OOB = HPRI_DMA_OOB;
  unsigned aoff[6];                            // halo piece q*4 + wave: pixel slot 16*(q*4 + wave) + (lane>>2), k-slot lane&3
  unsigned goff;                               // weight piece q*4 + wave of a stage: tap q, rows wave*16 + (lane>>2) of the block
  hpri_rsrc_t rs_a = HPRI_MAKE_RSRC(a.xp, 0x7FFFFF00);
  const hpri_rsrc_t rs_b = HPRI_MAKE_RSRC(a.wp, 0x7FFFFF00);
  const int tap_bytes = a.Cout_pad * 64;       // one tap of one chunk in the packed weights
  auto setup_dma = [&](const V3Tile& t) {
    const int TW = 1 << t.twl, TH = 256 >> t.twl, HW = TW + 2, HP = (TH + 2) * HW;
    const unsigned hw_inv = (65536u + (unsigned)HW - 1u) / (unsigned)HW;   // exact quotient for pix < 2048
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int pix = (q * 4 + wave) * 16 + (lane >> 2);
      unsigned off = OOB;
      if (pix < HP) {
        const int hy = (int)(((unsigned)pix * hw_inv) >> 16), hx = pix - hy * HW;
        const int iy = t.y0 + hy - 1, ix = t.x0 + hx - 1;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
          off = (unsigned)((iy * a.W + ix) * a.x_cs + (((lane & 3) ^ (((pix >> 2) & 1) << 1)) << 3)) * 2u;
      }
      aoff[q] = off;
    }
    const int n = wave * 16 + (lane >> 2);
    goff = (unsigned)((t.nb * 64 + n) * 32 + (((lane & 3) ^ (((n >> 2) & 1) << 1)) << 3)) * 2u;
    // (the descriptor must be PROVABLY wave-uniform, or hipcc wraps every DMA that uses it in a waterfall loop:
    // cdna_hip_programming.md T20 -- readfirstlane on the two halves of the base pointer says so)
    const unsigned long long pb = (unsigned long long)(uintptr_t)(a.xp + (size_t)t.img * a.H * a.W * a.x_cs + a.x_coff);
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pb), phi = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32));
    (void)plo; (void)phi;
    rs_a = HPRI_MAKE_RSRC((((unsigned long long)phi << 32) | plo), 0x7FFFFF00);
  };
  (void)goff; (void)tap_bytes; (void)rs_b;
// Diagnostic builds (tools/build_v3_diag.sh; results are wrong by construction, only cycles matter): V3_DIAG bit 0 no weight
// DMA inside the loop, bit 1 no halo DMA inside the loop, bit 2 fragments read once per stage only (tap 0), bit 3 no waits and
// no barriers inside the loop.
#ifndef V3_DIAG
#define V3_DIAG 0
#endif
#ifndef V3_PREFETCH_A
#define V3_PREFETCH_A 0
#endif
#ifndef V3_FAST_EPILOGUE
#define V3_FAST_EPILOGUE 1
#endif
// (buffer indices are compile-time: the chunk loop is unrolled by two, so LDS addresses are immediates of the reads)
#define V3_DMA_B_(buf_, s_, q_)                                                                                       \
  HPRI_LDS_DMA16(rs_b, b_lds + (buf_) * V3_B_BYTES + ((q_) * 4 + wave) * 1024, goff, ((s_) * 3 + (q_)) * tap_bytes)
#define V3_DMA_A_(buf_, c_, q_)                                                                                       \
  HPRI_LDS_DMA16(rs_a, a_lds + (buf_) * V3_A_BYTES + ((q_) * 4 + wave) * 1024, aoff[q_], (c_) * 64)
#define V3_DMA_B(buf_, s_, q_) { if (!(V3_DIAG & 1) || !in_loop) { V3_DMA_B_(buf_, s_, q_); } }
#define V3_DMA_A(buf_, c_, q_) { if (!(V3_DIAG & 2) || !in_loop) { V3_DMA_A_(buf_, c_, q_); } }
#define V3_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define V3_BARRIER()                         \
  __builtin_amdgcn_sched_barrier(0);         \
  __builtin_amdgcn_s_barrier();              \
  __builtin_amdgcn_sched_barrier(0)
// the first halo (six pieces per wave) and the first weight stage of an item: halo buffer 0, weight buffer 0
#define V3_PROLOGUE_LOADS()                                                   \
  {                                                                           \
    constexpr bool in_loop = false;                                           \
    _Pragma("unroll") for (int q = 0; q < 6; ++q) V3_DMA_A(0, chunk0, q);     \
    _Pragma("unroll") for (int q = 0; q < 3; ++q) V3_DMA_B(0, S0, q);         \
  }

  // ---- fragment addresses.  B operand of the MFMA (pixels): lane (li, lq) reads k-slot lq of pixel li of M-tile mt at tap
  //      (dy, dx); the swizzle depends on the halo pixel, so the nine tap offsets of each M-tile are tabulated -- once per tile
  //      SHAPE (they do not depend on where the tile lies) ----
  int aofs[4][9];
  auto build_aofs = [&](int twl) {
    const int TW = 1 << twl, HW = TW + 2;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int p = (wave * 4 + mt) * 16 + li;
      const int row = p >> twl, col = p & (TW - 1);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hp = (row + t / 3) * HW + col + (t % 3);
        aofs[mt][t] = hp * 64 + ((lq ^ (((hp >> 2) & 1) << 1)) << 4);
      }
    }
  };
  // A operand (weights): k-slot lq of channel row nt*16 + li; nt and the tap are immediates
  const int bofs = li * 64 + ((lq ^ (((li >> 2) & 1) << 1)) << 4);

  V3Tile cur, nxt;
  // ---- first item.  Fixed lists: item k = id / 8 of the band, then k + nloc, ...  Item queue (common.h): the first occupants of
  //      the CUs (ids below the CU count) start on the same item and draw from their second item on (ticket t = item nstatic + t);
  //      the second occupants draw their first two items now, and the answer arrives while they sit out their stagger ----
  int k = (int)(blockIdx.x >> 3);
  const int nstatic = min(nloc, a.ncu >> 3);
  const bool late_start = (unsigned)(blockIdx.x - a.ncu) < (unsigned)a.ncu;
  unsigned pend = 0u;                          // (thread 0) the ticket of the NEXT item, drawn one item ahead
  if (q != nullptr && tid == 0) pend = hpri_q_draw(q, xcd, k < nstatic ? 1u : 2u);
  // Two workgroups share a CU and every workgroup of a launch runs the same program on items of the same size: the second
  // occupants (workgroups [ncu, 2 ncu) in dispatch order) start late once, so that from then on one workgroup's epilogue runs
  // under the other's main loop.  Placement is not promised by HIP: another dispatch order costs the overlap, never correctness.
  if (late_start && a.stagger_cycles > 0) {
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < (long long)a.stagger_cycles) __builtin_amdgcn_s_sleep(32);
  }
  if (q != nullptr && k >= nstatic) {
    if (tid == 0) { next_lds[0] = nstatic + (int)pend; pend += 1u; }
    __syncthreads();
    k = __builtin_amdgcn_readfirstlane(next_lds[0]);
    __syncthreads();                           // (slot 0 is written again at the top of the first item)
  }
  bool have = tile_of(k, cur);
  if (!have) return;
  setup_dma(cur);
  build_aofs(cur.twl);
  // the channel block's biases go through LDS (16 scalar loads per lane in the epilogue were 16 serialized round trips: hipcc
  // waits vmcnt(0) behind every conditional load): wave 0 loads the NEXT item's 64 values while the current item's results are
  // written out and stores them at the top of that item (two slots: no wave is more than one item behind)
  float bias_next = 0.f;
  if (tid < 64) {
    const int n = cur.nb * 64 + tid;
    bias_next = (a.bias != nullptr && a.ksplit == 1 && n < a.Cout) ? a.bias[n] : 0.f;
  }

#ifdef HPRI_STAMPS
  if (a.stamps != nullptr && threadIdx.x == 0) a.stamps[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16 + 7] = __builtin_amdgcn_s_memrealtime();
  long long wait_cycles = 0;                   // cycles wave 0 spent between the top of a stage and the end of its barrier
  int ntiles_done = 0;
#endif
  if (tid == 0) *arrive_lds = 0u;              // (visible behind the first stage's barrier, long before its first use)
  V3_STAMP(0)
  V3_PROLOGUE_LOADS()
  constexpr bool in_loop = true;
  int slot = 0;

  // One stage = one kernel row of one 32-channel chunk: 3 taps x (4 pixel + 4 weight fragments, 16 MFMAs) per wave.
  //   top     wait for this wave's pieces of stage s (vmcnt counts in issue order: what the previous stage issued AFTER them --
  //           half a halo -- may stay in flight), then the barrier: every wave's pieces are visible, and every wave has left
  //           stage s-1, whose weight buffer (and, on dy = 0, the halo buffer of chunk c-1) may now be refilled
  //   body    weight pieces of stage s+1 first, then (dy = 0, 1) one half of the next chunk's halo, issued between the MFMAs
  // A fragment is read one tap ahead of its MFMAs.
#define V3_READ_TAP_A(fa_, ab_, dy_, dx_)                                                                             \
  _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                                    \
      fa_[mt] = *reinterpret_cast<const bf16x8*>((ab_) + aofs[mt][(dy_) * 3 + (dx_)]);
#define V3_READ_TAP_B(fb_, bb_, dx_)                                                                                  \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                                    \
      fb_[nt] = *reinterpret_cast<const bf16x8*>((bb_) + (dx_) * 4096 + nt * 1024);
#define V3_READ_TAP(fa_, fb_, ab_, bb_, dy_, dx_) V3_READ_TAP_A(fa_, ab_, dy_, dx_) V3_READ_TAP_B(fb_, bb_, dx_)
#define V3_MFMA_TAP(fa_, fb_, dma0_, dma1_)                                                                           \
  _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                                  \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                                  \
        acc[mt][nt] = HPRI_MFMA_16X16X32(fb_[nt], fa_[mt], acc[mt][nt], 0, 0, 0);                \
    if (mt == 0) { dma0_ }                                                                                            \
    if (mt == 2) { dma1_ }                                                                                            \
  }
#define V3_STAGE(par_, dy_)    /* par_: parity of the chunk relative to chunk0 = its halo buffer; stage buffer (3 par_ + dy_) & 1 */ \
  {                                                                                                                   \
    const int s_ = c * 3 + (dy_);                                                                                     \
    constexpr int bb_i = ((par_) * 3 + (dy_)) & 1;                                                                    \
    const unsigned char* ab_ = a_lds + (par_) * V3_A_BYTES;                                                           \
    const unsigned char* bb_ = b_lds + bb_i * V3_B_BYTES + bofs;                                                      \
    bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];                                                                            \
    /* the chunk's halo became visible at its dy = 0 stage: on dy = 1, 2 the pixel fragments of the first tap can be on their way \
       while this wave waits for the stage's weights (with the partner workgroup in its epilogue nothing else covers that wait) */ \
    if (V3_PREFETCH_A && (dy_) > 0) { V3_READ_TAP_A(fa0, ab_, dy_, 0) }                                               \
    V3_TOP_BEGIN()                                                                                                    \
    if (!(V3_DIAG & 8) || s_ == S0) {                                                                                 \
      if ((dy_) == 0 || !more_c) V3_WAIT_VM(0); else V3_WAIT_VM(3);                                                   \
      V3_BARRIER();                                                                                                   \
    }                                                                                                                 \
    V3_TOP_END()                                                                                                      \
    const bool more_b = s_ + 1 < S;                                                                                   \
    const bool more_a = (dy_) < 2 && more_c;                                                                          \
    if (V3_PREFETCH_A && (dy_) > 0) { V3_READ_TAP_B(fb0, bb_, 0) } else { V3_READ_TAP(fa0, fb0, ab_, bb_, dy_, 0) }   \
    if (!(V3_DIAG & 4)) { V3_READ_TAP(fa1, fb1, ab_, bb_, dy_, 1) }                                                   \
    else { _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { fa1[i_] = fa0[i_]; fb1[i_] = fb0[i_]; } }                \
    V3_ISSUE_FIRST(V3_I0 V3_I1 V3_I2 V3_I3(par_, dy_) V3_I4(par_, dy_) V3_I5(par_, dy_))                              \
    V3_SETPRIO(1);                                                                                                    \
    V3_MFMA_TAP(fa0, fb0, V3_ISSUE_MID(V3_I0), V3_ISSUE_MID(V3_I1))                                                   \
    if (!(V3_DIAG & 4)) { V3_READ_TAP(fa0, fb0, ab_, bb_, dy_, 2) }                                                   \
    V3_MFMA_TAP(fa1, fb1, V3_ISSUE_MID(V3_I2), V3_ISSUE_MID(V3_I3(par_, dy_)))                                        \
    V3_MFMA_TAP(fa0, fb0, V3_ISSUE_MID(V3_I4(par_, dy_)), V3_ISSUE_MID(V3_I5(par_, dy_)))                             \
    V3_SETPRIO(0);                                                                                                    \
  }
// the six DMA slots of a stage, in issue order: the weight pieces of stage s+1, then half of the next chunk's halo
#define V3_I0 if (more_b) { V3_DMA_B(bb_i ^ 1, s_ + 1, 0); }
#define V3_I1 if (more_b) { V3_DMA_B(bb_i ^ 1, s_ + 1, 1); }
#define V3_I2 if (more_b) { V3_DMA_B(bb_i ^ 1, s_ + 1, 2); }
#define V3_I3(par_, dy_) if (more_a) { V3_DMA_A((par_) ^ 1, c + 1, ((dy_) & 1) * 3 + 0); }
#define V3_I4(par_, dy_) if (more_a) { V3_DMA_A((par_) ^ 1, c + 1, ((dy_) & 1) * 3 + 1); }
#define V3_I5(par_, dy_) if (more_a) { V3_DMA_A((par_) ^ 1, c + 1, ((dy_) & 1) * 3 + 2); }
// (measured and dropped in round 3, each neutral on one box: all six DMA pieces right behind the barrier instead of between the
// MFMAs, no s_setprio at all, the epilogue at raised priority: DESIGN.md 4)
#define V3_ISSUE_FIRST(x_)
#define V3_ISSUE_MID(x_) x_
#define V3_SETPRIO(p_) __builtin_amdgcn_s_setprio(p_)
#ifdef HPRI_STAMPS
#define V3_TOP_BEGIN() long long tb_; { __builtin_amdgcn_sched_barrier(0); tb_ = (long long)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define V3_TOP_END() { __builtin_amdgcn_sched_barrier(0); wait_cycles += (long long)__builtin_amdgcn_s_memtime() - tb_; __builtin_amdgcn_sched_barrier(0); }
#else
#define V3_TOP_BEGIN()
#define V3_TOP_END()
#endif

  while (have) {
    const int twl = cur.twl, TW = 1 << twl;
    if (tid < 64) bias_lds[slot * 64 + tid] = bias_next;      // visible to everyone behind the first stage's barrier
    // the NEXT item's ticket (drawn during the previous item's epilogue) is parked in LDS: it does not live in a register across
    // the main loop.  (The first stage waits for vmcnt(0) anyway: this wait is that one, a few instructions early.)
    if (q != nullptr && tid == 0) next_lds[slot] = nstatic + (int)pend;

    f32x4 acc[4][4];                           // [pixel tile mt][channel tile nt]: channels nt*16 + 4*lq + r, pixel mt*16 + li
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = chunk0; c < nchunks; ++c) {
      {
        const bool more_c = c + 1 < nchunks;
        V3_STAGE(0, 0)
        V3_STAGE(0, 1)
        V3_STAGE(0, 2)
        if (!more_c) break;
      }
      ++c;
      {
        const bool more_c = c + 1 < nchunks;
        V3_STAGE(1, 0)
        V3_STAGE(1, 1)
        V3_STAGE(1, 2)
      }
    }
#ifdef HPRI_STAMPS
    if (ntiles_done == 0) { V3_STAMP(1) }
#endif

    // ---- the next item's first halo and weight stage start to load now and land while this item's results are written out:
    //      every wave has left the main loop (barrier), so halo buffer 0 and weight buffer 0 are free whatever the chunk parity;
    //      the statistics scratch below lives in halo buffer 1 ----
    V3_BARRIER();
    if (q != nullptr) k = __builtin_amdgcn_readfirstlane(next_lds[slot]);      // (beyond the band: the workgroup is done)
    else k += nloc;
    have = tile_of(k, nxt);
    if (have) {
      setup_dma(nxt);
      V3_PROLOGUE_LOADS()
      if (tid < 64) {
        const int n = nxt.nb * 64 + tid;
        bias_next = (a.bias != nullptr && a.ksplit == 1 && n < a.Cout) ? a.bias[n] : 0.f;
      }
      if (q != nullptr && tid == 0) pend = hpri_q_draw(q, xcd, 1u);      // the ticket of the item after that one
    }

#ifdef HPRI_STAMPS
    if (ntiles_done == 0) { V3_STAMP(10) }
#endif
    // ------------------------------- epilogue -------------------------------
    // acc[mt][nt][r]: pixel (wave*4 + mt)*16 + li of the tile, channel nb*64 + nt*16 + 4*lq + r
    const bool raw = a.ksplit > 1;             // split-K: raw partial sums into the workspace slab of this K slice
    float* dst = raw ? a.ws + (size_t)blockIdx.z * ((size_t)a.N * a.H * a.W) * a.Cout_pad : a.y;
    const int dcs = raw ? a.Cout_pad : a.y_cs, dco = raw ? 0 : a.y_coff, dcw = raw ? a.Cout_pad : a.y_cw;
    const int nlane = cur.nb * 64 + 4 * lq;    // first channel of this lane in channel tile 0
    if (!raw) {
      // pad channels need no mask: their weights are zero (pack) and their bias slot is zero, so their sums are exact zeros
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_lds + slot * 64 + nt * 16 + 4 * lq);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          f32x4 v = acc[mt][nt] + b4;
          if (a.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          acc[mt][nt] = v;
        }
      }
    }
    unsigned vmask = 0u;                       // bit mt: this lane's pixel of M-tile mt lies inside the image
    float* prow[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int p = (wave * 4 + mt) * 16 + li;
      const int iy = cur.y0 + (p >> twl), ix = cur.x0 + (p & (TW - 1));
      if (iy < a.H && ix < cur.xlim) vmask |= 1u << mt;
      prow[mt] = dst + ((size_t)(cur.img * a.H + min(iy, a.H - 1)) * a.W + min(ix, a.W - 1)) * dcs + dco + nlane;
    }
    // The accumulate / plain decision is taken ONCE, outside the store sequence: a conditional load in front of each store
    // makes hipcc wait vmcnt(0) at the join, which also waits for every earlier STORE (16 serialized write round trips per
    // wave: 10.7 k cycles in the first version of this epilogue, tools/v3_stamps.py).  So is "all 64 channels of the block lie
    // inside the written width" (everywhere except the last block of a narrow tensor): the common path has no per-store
    // condition at all.
    const bool full = cur.nb * 64 + 64 <= dcw;
    // BatchNorm-backward partial sums (see ConvV3Args): the pre-BN values at this lane's 4 pixels x 16 channels and the block's 64
    // parameter quadruples are requested BEFORE the stores (vmcnt counts in order: a load behind a store waits for the store)
    bf16x4_t xq[4][4];
    f32x4 prm4 = {0.f, 0.f, 0.f, 0.f};
    if (BNRED) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int p = (wave * 4 + mt) * 16 + li;
        const int iy = min(cur.y0 + (p >> twl), a.H - 1), ix = min(cur.x0 + (p & (TW - 1)), a.W - 1);
        const h16_t* xr = a.bn_x + ((size_t)(cur.img * a.H + iy) * a.W + ix) * a.bn_x_cs + a.bn_x_coff;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)      // channels beyond the readable width: any in-bounds quad (their g is an exact zero)
          xq[mt][nt] = *reinterpret_cast<const bf16x4_t*>(xr + min(nlane + nt * 16, a.bn_cw - 4));
      }
      if (tid < 64) {
        const int n = min(cur.nb * 64 + tid, a.Cout - 1);
        prm4 = f32x4{a.bn_scale[n], a.bn_shift[n], a.bn_mean[n], a.bn_invstd[n]};
      }
    }
#define V3_STORE_LOOP(ACC_, COND_)                                                                                    \
  if (ACC_) {               /* all loads first: a wait for a load behind a store would wait for the store as well */  \
    f32x4 old_[4][4];                                                                                                 \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                                  \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                              \
            old_[mt][nt] = (((vmask >> mt) & 1u) && (COND_)) ? *reinterpret_cast<const f32x4*>(prow[mt] + nt * 16)    \
                                                             : f32x4{0.f, 0.f, 0.f, 0.f};                             \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                                  \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[mt][nt] += old_[mt][nt];                                 \
  }                                                                                                                   \
  _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                                  \
    if ((vmask >> mt) & 1u) {                                                                                         \
      _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                                \
          if (COND_) *reinterpret_cast<f32x4*>(prow[mt] + nt * 16) = acc[mt][nt];                                     \
    }                                                                                                                 \
  }
    const bool to_y2 = !BNRED && !raw && a.y2 != nullptr && cur.nb * 64 >= a.y2_c0 && cur.nb * 64 < a.y2_c0 + a.y2_cw;
    // ---- round 4: the common case (a whole 64-channel block inside the written width, no second output, no split-K slab) goes out
    //      through a per-wave transposition in LDS.  Straight from the accumulators a store instruction covered 16 pixel rows x 64
    //      bytes -- 16 partial lines per instruction, 11.7 k cycles of store issue per item with a partner on the CU
    //      (tools/v3_stamps.py), another 7.5 k for the statistics' 128 DPP row sums.  Transposed, a lane holds channels 4 c .. 4 c + 3
    //      (c = lane & 15) of pixel 4 k + (lane >> 4) of an M-tile: one store instruction writes FOUR WHOLE pixel rows (1 KB
    //      contiguous for fp32; consecutive pixels of a tile row are consecutive in memory), and the statistics need two cross-lane
    //      steps per value instead of four DPP steps per accumulator register.  Scratch: 16 rows of 272 bytes per wave in halo
    //      buffer 1 (free until the next item's second chunk), private to the wave: no barrier.
    const bool fast = !BNRED && !raw && full && !to_y2 && V3_FAST_EPILOGUE;
    if (fast) {
      // (the lane-derived LDS addresses of this block are recomputed per item from an opaque copy of the lane id: hoisted out of the
      //  persistent loop by hipcc they stayed live across the main loop -- 8 registers it does not have)
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int li_e = lane_e & 15, lq_e = lane_e >> 4;
      unsigned char* tb = smem + V3_A_BYTES + wave * (16 * 272);
      const int lg = lane_e >> 4, lc = lane_e & 15;
      // the transposed values take the accumulators' own registers (a second array next to them spilled 200 registers):
      // from here on acc[mt][k] = pixel (wave*4 + mt)*16 + 4 k + lg, channels nb*64 + 4 lc + (0..3)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4*>(tb + li_e * 272 + nt * 64 + lq_e * 16) = acc[mt][nt];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc[mt][kk] = *reinterpret_cast<const f32x4*>(tb + (4 * kk + lg) * 272 + lc * 16);
      }
#ifdef HPRI_STAMPS
      if (ntiles_done == 0) { V3_STAMP(11) }
#endif
      unsigned vm = 0u;                        // bit 4 mt + k: that pixel lies inside the image
      unsigned poff[4][4];                     // element offset inside this image's view (< 2^32: the launcher bounds one image)
      float* const ybase = dst + (size_t)cur.img * a.H * a.W * dcs + dco + cur.nb * 64 + 4 * lc;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int p = (wave * 4 + mt) * 16 + 4 * kk + lg;
          const int iy = cur.y0 + (p >> twl), ix = cur.x0 + (p & (TW - 1));
          if (iy < a.H && ix < cur.xlim) vm |= 1u << (4 * mt + kk);
          poff[mt][kk] = (unsigned)((min(iy, a.H - 1) * a.W + min(ix, a.W - 1)) * dcs);
        }
      if (a.y16) {
        h16_t* d16 = reinterpret_cast<h16_t*>(a.y) + (ybase - dst);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            if ((vm >> (4 * mt + kk)) & 1u) {
              bf16x4_t h;
#pragma unroll
              for (int r = 0; r < 4; ++r) h[r] = (h16_t)acc[mt][kk][r];
              *reinterpret_cast<bf16x4_t*>(d16 + poff[mt][kk]) = h;
            }
      } else {
        if (a.accumulate) {
          // in two halves of eight pixel rows: all loads of a half before its stores (a wait for a load behind a store would wait
          // for the store as well), and 32 instead of 64 registers of old values next to the accumulators
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            f32x4 old_[2][4];
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) {
                const int mt = 2 * hh + m2;
                old_[m2][kk] = ((vm >> (4 * mt + kk)) & 1u) ? *reinterpret_cast<const f32x4*>(ybase + poff[mt][kk]) : f32x4{0.f, 0.f, 0.f, 0.f};
              }
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) {
                const int mt = 2 * hh + m2;
                acc[mt][kk] += old_[m2][kk];
                if ((vm >> (4 * mt + kk)) & 1u) *reinterpret_cast<f32x4*>(ybase + poff[mt][kk]) = acc[mt][kk];
              }
          }
        } else {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
              if ((vm >> (4 * mt + kk)) & 1u) *reinterpret_cast<f32x4*>(ybase + poff[mt][kk]) = acc[mt][kk];
        }
      }
#ifdef HPRI_STAMPS
      if (ntiles_done == 0) { V3_STAMP(12) }
#endif
      if (a.stats != nullptr) {
        // exact two-pass record of this wave's 64 pixels per channel (sum, then squared deviations from the wave's own mean), the
        // four lane groups of a channel quad met by two butterfly steps; the four wave records are merged below as before
        // sum over the four 16-lane rows (lanes c, c + 16, c + 32, c + 48): two half-exchanges in the vector ALU instead of two trips
        // through the LDS crossbar (v_permlane16_swap: rows 1, 3 of vdst <-> rows 0, 2 of src; v_permlane32_swap: upper half of vdst
        // <-> lower half of src; s_nop 1 = the two wait states a VALU write needs in front of a permlane read, T21)
        auto xsum = [](float v) {
          float p = v, q = v;
          asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q));
          float t = p + q, u = t;
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(t), "+v"(u));
          return t + u;
        };
        const float cntw = xsum((float)__builtin_popcount(vm)) ;
        const float inv = cntw > 0.f ? 1.f / cntw : 0.f;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, mw;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            if ((vm >> (4 * mt + kk)) & 1u) s1 += acc[mt][kk];
#pragma unroll
        for (int r = 0; r < 4; ++r) mw[r] = xsum(s1[r]) * inv;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            if ((vm >> (4 * mt + kk)) & 1u) { const f32x4 d = acc[mt][kk] - mw; s2 += d * d; }
#pragma unroll
        for (int r = 0; r < 4; ++r) s2[r] = xsum(s2[r]);
        float* red = reinterpret_cast<float*>(smem + V3_A_BYTES + 4 * (16 * 272));      // behind the transposition rows
        if (lg == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            red[(wave * 64 + 4 * lc + r) * 2 + 0] = mw[r];
            red[(wave * 64 + 4 * lc + r) * 2 + 1] = s2[r];
          }
        }
        if (lane == 0) red[4 * 128 + wave] = cntw;
        // No workgroup barrier here (it cost every wave the skew of the slowest one, ~3-4 k cycles of a 19 k epilogue): the wave
        // that arrives LAST merges the four records, in wave order as before; the others go on to the next item.  Its first
        // stage barrier -- which the merging wave joins too -- comes before any DMA that could overwrite this scratch.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's record is in LDS before it counts itself in
        unsigned arrived = 0u;
        // release on the count (this wave's record happens-before it), acquire for the merging wave (the other three waves' records
        // happen-before its reads): at workgroup scope on LDS both are waitcnts only, but the compiler may no longer move the
        // red[] loads below above the atomic
        if (lane == 0) arrived = __hip_atomic_fetch_add(arrive_lds, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)arrived);
        asm volatile("" ::: "memory");
        if (arrived == 3u) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // (lanes 1..63 did not take part in the atomic)
          if (lane == 0) *arrive_lds = 0u;
          const int ch = lane;                   // one channel per lane
          float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) {
            const float nb_ = red[4 * 128 + w2];
            if (nb_ > 0.f) {
              const float mb = red[(w2 * 64 + ch) * 2 + 0], qb = red[(w2 * 64 + ch) * 2 + 1];
              const float tot = n + nb_, delta = mb - mean, f = __builtin_amdgcn_rcpf(tot) * nb_;
              mean += delta * f;
              m2 += qb + delta * delta * (n * f);
              n = tot;
            }
          }
          a.stats[(size_t)cur.bx * a.Cout_pad + cur.nb * 64 + ch] = make_float4(mean, m2, n, 0.f);
        }
      }
    }
    if (to_y2) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if ((vmask >> mt) & 1u) {
          const int p = (wave * 4 + mt) * 16 + li;
          const int iy = cur.y0 + (p >> twl), ix = cur.x0 + (p & (TW - 1));
          h16_t* q = a.y2 + ((size_t)(cur.img * a.H + iy) * a.W + ix) * a.y2_cs + a.y2_coff + (cur.nb * 64 - a.y2_c0) + 4 * lq;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            bf16x4_t h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (h16_t)acc[mt][nt][r];
            *reinterpret_cast<bf16x4_t*>(q + nt * 16) = h;
          }
        }
      }
    }
    if (fast) {
      // (written above)
    } else if (to_y2 && a.y2_only) {
      // (this block of the result exists as bf16 rows only)
    } else if (!BNRED && !raw && a.y16) {
      // bf16 output: a lane's four channels are one 8-byte store (round-to-nearest-even, v_cvt_pk_bf16_f32)
      h16_t* d16 = reinterpret_cast<h16_t*>(a.y);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if ((vmask >> mt) & 1u) {
          h16_t* q = d16 + (prow[mt] - dst);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            if (full || nlane + nt * 16 < dcw) {
              bf16x4_t h;
#pragma unroll
              for (int r = 0; r < 4; ++r) h[r] = (h16_t)acc[mt][nt][r];
              *reinterpret_cast<bf16x4_t*>(q + nt * 16) = h;
            }
        }
      }
    } else if (!BNRED && !raw && a.accumulate) {
      if (full) { V3_STORE_LOOP(true, true) } else { V3_STORE_LOOP(true, nlane + nt * 16 < dcw) }
    } else {
      if (full) { V3_STORE_LOOP(false, true) } else { V3_STORE_LOOP(false, nlane + nt * 16 < dcw) }
    }
#undef V3_STORE_LOOP
#ifdef HPRI_STAMPS
    if (ntiles_done == 0) { V3_STAMP(2) }
#endif
    if (!fast && !BNRED && !raw && a.stats != nullptr) {
      // per-tile, per-channel (mean, M2, count): each wave makes an exact two-pass record of its own 64 pixels (sum, then
      // squared deviations from its own mean); the four wave records of a channel are merged with Chan's update after one
      // barrier.  (raw barriers: __syncthreads() would also wait for the output stores above.)
      float cntl = (float)__builtin_popcount(vmask);
      const float cntw = v3_row_sum(cntl);     // valid pixels of this wave's 64 (each DPP row holds all 16 pixel columns)
      const float inv = cntw > 0.f ? 1.f / cntw : 0.f;
      float* red = reinterpret_cast<float*>(smem + V3_A_BYTES);         // [4 waves][64 channels][2] + [4] counts
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          if ((vmask >> mt) & 1u) s1 += acc[mt][nt];
        f32x4 mw, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) mw[r] = v3_row_sum(s1[r]) * inv;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          if ((vmask >> mt) & 1u) { const f32x4 d = acc[mt][nt] - mw; s2 += d * d; }
#pragma unroll
        for (int r = 0; r < 4; ++r) s2[r] = v3_row_sum(s2[r]);
        if (li == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            red[(wave * 64 + nt * 16 + 4 * lq + r) * 2 + 0] = mw[r];
            red[(wave * 64 + nt * 16 + 4 * lq + r) * 2 + 1] = s2[r];
          }
        }
      }
      if (lane == 0) red[4 * 128 + wave] = cntw;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      V3_BARRIER();
      if (tid < 64) {
        float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
          const float nb_ = red[4 * 128 + w2];
          if (nb_ > 0.f) {
            const float mb = red[(w2 * 64 + tid) * 2 + 0], qb = red[(w2 * 64 + tid) * 2 + 1];
            const float tot = n + nb_, delta = mb - mean, f = __builtin_amdgcn_rcpf(tot) * nb_;
            mean += delta * f;
            m2 += qb + delta * delta * (n * f);
            n = tot;
          }
        }
        a.stats[(size_t)cur.bx * a.Cout_pad + cur.nb * 64 + tid] = make_float4(mean, m2, n, 0.f);
      }
      // (the scratch is rewritten only after the next item's main loop, i.e. behind many barriers)
    }
    if (BNRED) {
      // scratch in halo buffer 1 behind the statistics': [64] parameter quads, then [4 waves][2 sums][64 channels]
      float* prm = reinterpret_cast<float*>(smem + V3_A_BYTES + 4096);
      float* red2 = prm + 256;
      if (tid < 64) *reinterpret_cast<f32x4*>(prm + tid * 4) = prm4;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      V3_BARRIER();
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(prm + (nt * 16 + 4 * lq + r) * 4);     // scale, shift, mean, invstd
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const float xf = (float)xq[mt][nt][r];
            const bool keep = ((vmask >> mt) & 1u) && (!a.bn_relu || (xf * q[0] + q[1] > 0.f));
            const float gj = keep ? acc[mt][nt][r] : 0.f;
            t1[r] += gj;
            t2[r] += gj * ((xf - q[2]) * q[3]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { t1[r] = v3_row_sum(t1[r]); t2[r] = v3_row_sum(t2[r]); }
        if (li == 0) {
          *reinterpret_cast<f32x4*>(red2 + (wave * 2 + 0) * 64 + nt * 16 + 4 * lq) = t1;
          *reinterpret_cast<f32x4*>(red2 + (wave * 2 + 1) * 64 + nt * 16 + 4 * lq) = t2;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      V3_BARRIER();
      if (tid < 128) {                           // (sum, channel) = (tid >> 6, tid & 63): the four wave records in a fixed order
        const int which = tid >> 6, c = tid & 63;
        float tsum = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) tsum += red2[(w2 * 2 + which) * 64 + c];
        if (cur.nb * 64 + c < a.bn_cpart) a.bn_part[((size_t)cur.bx * 2 + which) * a.bn_cpart + cur.nb * 64 + c] = tsum;
      }
    }
#ifdef HPRI_STAMPS
    if (ntiles_done == 0) { V3_STAMP(3) }
    ++ntiles_done;
#endif
    if (have && nxt.twl != cur.twl) build_aofs(nxt.twl);
    cur = nxt;
    slot ^= 1;
  }
#ifdef HPRI_STAMPS
  if (a.stamps != nullptr && threadIdx.x == 0) {
    unsigned long long* sp = a.stamps + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16;
    sp[4] = (unsigned long long)wait_cycles;
    sp[5] = __builtin_amdgcn_s_memrealtime();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    sp[6] = ((unsigned long long)xcc << 32) | hw;
    sp[8] = __builtin_amdgcn_s_memtime();
    sp[9] = (unsigned long long)ntiles_done;
  }
#endif
#undef V3_STAGE
#undef V3_I0
#undef V3_I1
#undef V3_I2
#undef V3_I3
#undef V3_I4
#undef V3_I5
#undef V3_ISSUE_FIRST
#undef V3_ISSUE_MID
#undef V3_MFMA_TAP
#undef V3_READ_TAP
#undef V3_READ_TAP_A
#undef V3_READ_TAP_B
#undef V3_SETPRIO
#undef V3_PROLOGUE_LOADS
#undef V3_DMA_A
#undef V3_DMA_B
#undef V3_DMA_A_
#undef V3_DMA_B_
#undef V3_TOP_BEGIN
#undef V3_TOP_END
#undef V3_WAIT_VM
#undef V3_BARRIER
}

// ---- host side -------------------------------------------------------------------------------------------------------
// column bands of tile width 32 / 16 / 8 (tile height 8 / 16 / 32: 256 pixels either way), chosen by padding cost.  The fragment
// swizzle is conflict-free for runs of 16 consecutive halo pixels; an 8-wide tile reads two runs of 8 (possibly 2-way conflicted), so
// at most ONE column of them is used, for a remainder of at most 8 columns -- round 5: W = 968 is 30 x 32 + 8, and with the last 8
// columns in 16-wide tiles the benched layer had 4636 items = 9.05 rounds of the 512 resident workgroups (a tenth round for 28 items);
// 32 x 8 tiles there make it 4598 = 8.98 rounds.  Plan option bf16v3_tile_width: 0 = this, 1 = no 8-wide column, 16 = 16 x 16 only.
struct V3Segs { int nseg, tiles_img, twl[V3_MAXSEG], xbeg[V3_MAXSEG], ntx[V3_MAXSEG], first[V3_MAXSEG]; };
static V3Segs v3_segments(int H, int W) {
  auto th = [&](int tw) { return 256 / tw; };
  auto slots = [&](int tw, int ntx) { return (long long)hpri_cdiv(H, th(tw)) * th(tw) * tw * ntx; };
  V3Segs best{};
  long long best_cost = -1;
  const int opt = hpri_option(3);
  const bool allow8 = opt != 1 && opt != 16;
  // candidates: n32 columns of 32-wide tiles, the rest in 16-wide tiles (the last of them 8 wide when it holds at most 8 columns)
  const int max32 = opt == 16 ? 0 : hpri_cdiv(W, 32);
  for (int n32 = 0; n32 <= max32; ++n32) {
    const int rem = W - n32 * 32;
    int n16 = rem > 0 ? hpri_cdiv(rem, 16) : 0, n8 = 0;
    if (rem <= 0 && n32 * 32 - W >= 32) continue;
    if (allow8 && n16 > 0 && rem - (n16 - 1) * 16 <= 8) { n16 -= 1; n8 = 1; }
    const long long cost = (n32 ? slots(32, n32) : 0) + (n16 ? slots(16, n16) : 0) + (n8 ? slots(8, n8) : 0);
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && n16 + n8 == 0)) {
      V3Segs g{};
      if (n32) { g.twl[g.nseg] = 5; g.xbeg[g.nseg] = 0; g.ntx[g.nseg] = n32; g.nseg++; }
      if (n16) { g.twl[g.nseg] = 4; g.xbeg[g.nseg] = n32 * 32; g.ntx[g.nseg] = n16; g.nseg++; }
      if (n8) { g.twl[g.nseg] = 3; g.xbeg[g.nseg] = n32 * 32 + n16 * 16; g.ntx[g.nseg] = 1; g.nseg++; }
      best = g; best_cost = cost;
    }
  }
  int first = 0;
  for (int k = 0; k < best.nseg; ++k) { best.first[k] = first; first += hpri_cdiv(H, th(1 << best.twl[k])) * best.ntx[k]; }
  best.tiles_img = first;
  return best;
}

// Split-K (host only), priced as in conv_bf16v2.hip: a slice costs 2 k output sizes of fp32 slab traffic against the partial
// round of workgroup slots (two per CU) it fills; only problems below half a round are cut.
static int v3_ksplit(int N, int H, int W, int Cin_pad, int Cout_pad) {
  const int ncu = hpri_cu_count();
  const long long blocks = (long long)N * v3_segments(H, W).tiles_img * (Cout_pad / 64);
  const int nchunks = Cin_pad / 32;
  if (blocks >= ncu) return 1;
  const double t_compute = 2.0 * N * H * W * (double)Cin_pad * Cout_pad * 9.0 / 900e12;
  const double out_bytes = 4.0 * N * H * W * (double)Cout_pad;
  int best = 1; double best_t = 1e30;
  for (int k = 1; k <= 4; ++k) {
    if (k > 1 && nchunks / k < 4) break;
    const double per_slot = (double)blocks * k / (2.0 * ncu);
    const double eff = per_slot / (double)((long long)(per_slot + 0.999999));
    const double t = t_compute / eff + (k > 1 ? 2.0 * k * out_bytes / 5e12 + 4e-6 : 0.0);
    if (t < best_t - 1e-12) { best_t = t; best = k; }
  }
  return best;
}

#define V3_SK_PIX 64
extern "C" int hpri_conv_bf16v3_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int* ksplit, int* stat_tiles,
                                     size_t* ws_floats) {
  const int k = v3_ksplit(N, H, W, Cin_pad, Cout_pad);
  *ksplit = k;
  if (k > 1) { *stat_tiles = N * hpri_cdiv(H * W, V3_SK_PIX); *ws_floats = (size_t)k * N * H * W * Cout_pad; }
  else { *stat_tiles = N * v3_segments(H, W).tiles_img; *ws_floats = 0; }
  return HPRI_OK;
}

// conv_fwd.hip
extern "C" int hpri_splitk_finish(const float* ws, int ksplit, int Cout_pad, const float* bias, float* y, int y_cs, int y_coff,
                                  float* stats, int N, int HW, int Cout, int y_cw, int accumulate, int relu, hipStream_t stream);

#define V3_STAGGER_CYCLES 6000      // about one store + statistics epilogue with a partner on the CU

struct V3Out2 { void* y2; int cs, coff, c0, cw, only; };
struct V3BnRed {
  const void* x16; int x_cs, x_coff; const float *mean, *invstd, *scale, *shift; int relu; float* part; int cpart;
};

static int v3_launch(const void* xp, long long x_plane, int x_cs, int x_coff, const void* wp, const float* bias,
                     float* y, int y_cs, int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout,
                     int Cout_pad, int y_cw, int accumulate, int split, float* ws, size_t ws_floats,
                     unsigned long long* stamps, int stagger_cycles, const V3BnRed* bn, hipStream_t stream,
                     const V3Out2* o2 = nullptr) {
  HPRI_REQUIRE(xp && wp && y, "conv_bf16v3: null pointer");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0, "conv_bf16v3: empty image");
  HPRI_REQUIRE(Cin_pad > 0 && Cin_pad % 32 == 0, "conv_bf16v3: Cin_pad must be a positive multiple of 32");
  HPRI_REQUIRE(Cout_pad % 64 == 0 && Cout <= Cout_pad && Cout > 0, "conv_bf16v3: Cout_pad must be a multiple of 64 >= Cout");
  HPRI_REQUIRE(x_cs % 8 == 0 && x_coff % 8 == 0 && x_coff + Cin_pad <= x_cs, "conv_bf16v3: plane channel stride/offset must be multiples of 8 and hold Cin_pad channels");
  HPRI_REQUIRE(((uintptr_t)xp & 15) == 0 && ((uintptr_t)wp & 15) == 0, "conv_bf16v3: pointers must be 16-byte aligned");
  HPRI_REQUIRE((long long)H * W * x_cs * 2 < 0x7FFFFF00ll, "conv_bf16v3: one image of the input planes exceeds 2 GiB (32-bit DMA offsets)");
  HPRI_REQUIRE((long long)(Cin_pad / 32) * 9 * Cout_pad * 64 < 0x7FFFFF00ll, "conv_bf16v3: packed weights exceed 2 GiB");
  HPRI_REQUIRE(split == 0, "conv_bf16v3: only plain bf16 planes (split 0) are built");
  (void)x_plane;
  ConvV3Args a;
  a.xp = reinterpret_cast<const h16_t*>(xp); a.x_cs = x_cs; a.x_coff = x_coff;
  a.wp = reinterpret_cast<const h16_t*>(wp); a.bias = bias; a.y = y; a.y_cs = y_cs; a.y_coff = y_coff;
  a.stats = reinterpret_cast<float4*>(stats);
  a.N = N; a.H = H; a.W = W; a.Cin_pad = Cin_pad; a.Cout = Cout; a.Cout_pad = Cout_pad;
  a.y_cw = y_cw < Cout ? Cout : y_cw; a.accumulate = accumulate & 1; a.relu = (accumulate >> 1) & 1; a.y16 = (accumulate >> 2) & 1;
  if (o2 != nullptr && o2->only && a.y16) {
    // a compact bf16 main output: it holds the channels below the second output's range and nothing else
    HPRI_REQUIRE(y_cw >= o2->c0 && o2->c0 + o2->cw >= Cout, "conv_bf16v3_y2: a bf16 main output must cover every channel below the second output's range, which must reach Cout");
    a.y_cw = y_cw;
  }
  HPRI_REQUIRE(a.y_cw + y_coff <= y_cs, "conv_bf16v3: output channels exceed the channel stride");
  HPRI_REQUIRE(!(a.y16 && a.accumulate), "conv_bf16v3: a bf16 output cannot accumulate");
  HPRI_REQUIRE(y_cs % 4 == 0 && y_coff % 4 == 0 && a.y_cw % 4 == 0 && ((uintptr_t)y & 15) == 0,
               "conv_bf16v3: the output view must be float4-aligned (stride, offset and written width multiples of 4)");
  a.ksplit = v3_ksplit(N, H, W, Cin_pad, Cout_pad);
  a.ws = ws;
  HPRI_REQUIRE(!(a.y16 && a.ksplit > 1), "conv_bf16v3: a bf16 output is not available for split-K problems (hpri_conv_bf16v3_plan: ksplit > 1)");
  if (a.ksplit > 1) {
    if (ws == nullptr || (size_t)a.ksplit * N * H * W * Cout_pad > ws_floats)
      return hpri_set_error(HPRI_ERR_WORKSPACE, "conv_bf16v3: split-K workspace too small (see hpri_conv_bf16v3_plan)");
    a.stats = nullptr; a.accumulate = 0;
  }
  const V3Segs sg = v3_segments(H, W);
  a.nseg = sg.nseg; a.tiles_img = sg.tiles_img; a.ntiles = N * sg.tiles_img; a.nb_count = Cout_pad / 64;
  for (int k = 0; k < V3_MAXSEG; ++k) { a.seg_twl[k] = sg.twl[k]; a.seg_xbeg[k] = sg.xbeg[k]; a.seg_ntx[k] = sg.ntx[k]; a.seg_first[k] = sg.first[k]; }
  const long long items = (long long)a.ntiles * a.nb_count;
  HPRI_REQUIRE(items < (1ll << 28), "conv_bf16v3: too many work items");
  a.nb_major = (a.nb_count % 8 == 0) ? 1 : 0;
  a.per_xcd = a.nb_major ? a.ntiles * (a.nb_count / 8) : (int)((items + 7) / 8);
  a.ncu = hpri_cu_count(); a.stagger_cycles = stagger_cycles;
  a.queue = a.queue_clear = nullptr;      // (taken right before the launch: the stream's parity advances with every real launch)
#ifdef HPRI_STAMPS
  a.stamps = stamps;
#else
  (void)stamps;
#endif
  // persistent workgroups: two per CU (fewer when there are fewer items), a multiple of 8 so that id mod 8 labels the XCD
  int nloc = (2 * a.ncu / a.ksplit) / 8;
  if (nloc < 1) nloc = 1;
  if (nloc > a.per_xcd) nloc = a.per_xcd;
  dim3 grid((unsigned)(nloc * 8), 1u, (unsigned)a.ksplit);
  a.y2 = nullptr; a.y2_cs = a.y2_coff = a.y2_c0 = a.y2_cw = a.y2_only = 0;
  if (o2 != nullptr) {
    HPRI_REQUIRE(o2->y2 != nullptr && bn == nullptr && a.ksplit == 1 && !a.accumulate,
                 "conv_bf16v3_y2: the second output is not for split-K problems (hpri_conv_bf16v3_plan) or accumulating launches");
    HPRI_REQUIRE(o2->c0 % 64 == 0 && o2->cw % 64 == 0 && o2->cw > 0 && o2->c0 + o2->cw <= Cout_pad, "conv_bf16v3_y2: the channel range must be whole 64-channel blocks");
    HPRI_REQUIRE(o2->cs % 4 == 0 && o2->coff % 4 == 0 && o2->coff + o2->cw <= o2->cs && ((uintptr_t)o2->y2 & 7) == 0,
                 "conv_bf16v3_y2: the bf16 view must be 8-byte aligned and hold the channel range");
    a.y2 = reinterpret_cast<h16_t*>(o2->y2); a.y2_cs = o2->cs; a.y2_coff = o2->coff; a.y2_c0 = o2->c0; a.y2_cw = o2->cw; a.y2_only = o2->only;
  }
  a.bn_part = nullptr;
  if (bn != nullptr) {
    HPRI_REQUIRE(bn->x16 && bn->mean && bn->invstd && bn->scale && bn->shift && bn->part, "conv_bf16v3_bnred: null pointer");
    HPRI_REQUIRE(a.ksplit == 1 && !a.accumulate && !a.y16 && stats == nullptr,
                 "conv_bf16v3_bnred: not for split-K problems (hpri_conv_bf16v3_plan), accumulating launches, bf16 outputs or launches that record statistics");
    HPRI_REQUIRE(bn->x_cs % 4 == 0 && bn->x_coff % 4 == 0 && ((uintptr_t)bn->x16 & 7) == 0 && bn->x_cs - bn->x_coff >= 4,
                 "conv_bf16v3_bnred: the pre-BN view must be 8-byte aligned (stride and offset multiples of 4)");
    HPRI_REQUIRE(bn->x_cs - bn->x_coff >= ((Cout + 3) & ~3) && bn->cpart >= Cout, "conv_bf16v3_bnred: pre-BN view / partial rows narrower than the channels");
    a.bn_x = reinterpret_cast<const h16_t*>(bn->x16); a.bn_x_cs = bn->x_cs; a.bn_x_coff = bn->x_coff;
    a.bn_cw = (bn->x_cs - bn->x_coff) & ~3;
    a.bn_mean = bn->mean; a.bn_invstd = bn->invstd; a.bn_scale = bn->scale; a.bn_shift = bn->shift;
    a.bn_relu = bn->relu; a.bn_part = bn->part; a.bn_cpart = bn->cpart;
#ifdef HPRI_DIAG_KERNELS
    if (a.ksplit <= HPRI_Q_SLICES && a.per_xcd >= 2 * nloc) { const HpriQueueHalves qh = hpri_item_queue_take(stream); a.queue = qh.use; a.queue_clear = qh.clear; }
    hipLaunchKernelGGL(conv_bf16v3_kernel<true>, grid, dim3(256), 0, stream, a);
#else
    return hpri_set_error(HPRI_ERR_UNSUPPORTED, "conv_bf16v3_bnred: diagnostics build only (HPRI_DIAG=1 python -m hyperpri_amd.build)");
#endif
  } else {
    if (a.ksplit <= HPRI_Q_SLICES && a.per_xcd >= 2 * nloc) { const HpriQueueHalves qh = hpri_item_queue_take(stream); a.queue = qh.use; a.queue_clear = qh.clear; }
    hipLaunchKernelGGL(conv_bf16v3_kernel<false>, grid, dim3(256), 0, stream, a);
  }
  HPRI_CHECK_LAUNCH();
  if (a.ksplit == 1) return HPRI_OK;
  return hpri_splitk_finish(ws, a.ksplit, Cout_pad, bias, y, y_cs, y_coff, stats, N, H * W, Cout, a.y_cw, accumulate & 1, a.relu, stream);
}

#ifdef HPRI_DIAG_KERNELS   // (stagger and stamp buffer given by the caller: tools/v3_bench.py, tools/v3_stamps.py)
extern "C" int hpri_conv_bf16v3_dbg(const void* xp, long long x_plane, int x_cs, int x_coff, const void* wp, const float* bias,
                                    float* y, int y_cs, int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout,
                                    int Cout_pad, int y_cw, int accumulate, int split, float* ws, size_t ws_floats,
                                    unsigned long long* stamps, int stagger_cycles, hipStream_t stream) {
  return v3_launch(xp, x_plane, x_cs, x_coff, wp, bias, y, y_cs, y_coff, stats, N, H, W, Cin_pad, Cout, Cout_pad, y_cw, accumulate,
                   split, ws, ws_floats, stamps, stagger_cycles, nullptr, stream);
}
#endif   // HPRI_DIAG_KERNELS

#ifdef HPRI_DIAG_KERNELS   // measured neutral to -3 % in round 3: kept for A/B in the diagnostics build only
// The data gradient of a 3x3 layer whose input x = ReLU(BN(bn_x16)) has no other consumer, with that BatchNorm's backward
// reduction taken in the epilogue (the bf16-mode counterpart of hpri_conv_wino4_bnred): bn_x16 = the pre-BN tensor as bf16 (same
// pixels as y; bn_x_cs / bn_x_coff in elements), its per-channel mean / invstd / scale / shift, bn_relu; bn_part[stat tiles][2][bn_cpart]
// (stat tiles: hpri_conv_bf16v3_plan, which must report ksplit 1) receives sum g*[y>0] and sum g*[y>0]*xhat per tile.  Finish with
// hpri_bn_relu_bwd_fused.
extern "C" int hpri_conv_bf16v3_bnred(const void* xp, int x_cs, int x_coff, const void* wp, float* y, int y_cs, int y_coff, int N, int H,
                                      int W, int Cin_pad, int Cout, int Cout_pad, int y_cw, const void* bn_x16, int bn_x_cs,
                                      int bn_x_coff, const float* bn_mean, const float* bn_invstd, const float* bn_scale,
                                      const float* bn_shift, int bn_relu, float* bn_part, int bn_cpart, hipStream_t stream) {
  const V3BnRed bn{bn_x16, bn_x_cs, bn_x_coff, bn_mean, bn_invstd, bn_scale, bn_shift, bn_relu, bn_part, bn_cpart};
  return v3_launch(xp, 0, x_cs, x_coff, wp, nullptr, y, y_cs, y_coff, nullptr, N, H, W, Cin_pad, Cout, Cout_pad, y_cw, 0, 0, nullptr, 0,
                   nullptr, V3_STAGGER_CYCLES, &bn, stream);
}

#endif   // HPRI_DIAG_KERNELS

// hpri_conv_bf16v3 (no accumulate, no split-K) whose result channels [c0, c0 + cw) -- whole 64-channel blocks -- are also
// (y2_only != 0: only) written as bf16 rows: y2 + pixel * y2_cs + y2_coff + (channel - c0).  The data gradient of the first
// convolution of a decoder stage leaves the gradient of the upsampled half of its concat input this way, for the plane-fed
// transposed-convolution kernels (hpri_convt_dgrad_bf16v3, hpri_wgrad_convt_bf16v3).
extern "C" int hpri_conv_bf16v3_y2(const void* xp, int x_cs, int x_coff, const void* wp, const float* bias, float* y, int y_cs, int y_coff,
                                   float* stats, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw, void* y2, int y2_cs,
                                   int y2_coff, int y2_c0, int y2_cw, int y2_only, hipStream_t stream) {
  // y2_only: bit 0 = the blocks of [c0, c0 + cw) exist as bf16 rows only; bit 1 (round 4) = the MAIN output y is bf16 rows too (y_cs in
  // elements; the gradient of a planes-only skip tensor: its readers -- pooling backward, BatchNorm backward -- read bf16)
  const V3Out2 o2{y2, y2_cs, y2_coff, y2_c0, y2_cw, y2_only & 1};
  return v3_launch(xp, 0, x_cs, x_coff, wp, bias, y, y_cs, y_coff, stats, N, H, W, Cin_pad, Cout, Cout_pad, y_cw, (y2_only & 2) ? 4 : 0, 0,
                   nullptr, 0, nullptr, V3_STAGGER_CYCLES, nullptr, stream, &o2);
}

// 3x3 pad-1 convolution (forward, or data gradient with the flipped pack) over bf16 activation planes: same argument
// contract as hpri_conv_bf16v2 (x_plane is unused: one plane); statistics records per 256-pixel tile (hpri_conv_bf16v3_plan).
extern "C" int hpri_conv_bf16v3(const void* xp, long long x_plane, int x_cs, int x_coff, const void* wp, const float* bias,
                                float* y, int y_cs, int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout,
                                int Cout_pad, int y_cw, int accumulate, int split, float* ws, size_t ws_floats,
                                hipStream_t stream) {
  return v3_launch(xp, x_plane, x_cs, x_coff, wp, bias, y, y_cs, y_coff, stats, N, H, W, Cin_pad, Cout, Cout_pad, y_cw, accumulate,
                   split, ws, ws_floats, nullptr, V3_STAGGER_CYCLES, nullptr, stream);
}
