// 3x3 implicit-GEMM convolution on bf16 activation PLANES (precision modes "bf16" / "bf16x3" / "bf16x6"), forward and
// data gradient (reference model_parts.py:22,25; models.py:169,177 and their autograd).
//
// Round-1's bf16 kernel read fp32 activations, rounded them in VGPRs and ds_write'd them into LDS on every staging:
// the 238->64 layer moved 4 B per input element over HBM and spent VALU / LDS-store time per staged element.  Here the
// PRODUCER of an activation (BN-apply, pooling, concat, ingest, BN-backward) has already written it as bf16 NHWC planes
// (hi | hi,lo | hi,mid,lo), so both operands reach LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no
// conversion, no ds_write, and half the HBM bytes.
//
//   workgroup  512 threads = 8 waves as WM x WN (8x1: 512 px x 64 ch, 4x2: 256 px x 128 ch); wave tile 64 px x 64 ch
//              = 2x2 tiles of v_mfma_f32_32x32x16_bf16 (64 accumulator VGPRs); ONE workgroup per CU, two waves per SIMD
//   A (input)  halo of (TH+2) x (TW+2) pixels x 32 channels per plane, 64-byte pixel rows, unpadded; the four 16-byte
//              slots of a pixel are XOR-swizzled with (pixel>>2)&3 (the DMA cannot permute its destination, but every
//              lane picks WHICH source slot it fetches) -> conflict-free ds_read_b128 fragments for any tap offset;
//              pixels outside the image get an out-of-range buffer offset (the descriptor's range check writes zeros into
//              their LDS slots); double-buffered per 32-channel chunk
//   B (weights) per stage = one kernel row (KS taps x planes x BN rows of 32 k), same swizzle, double-buffered
//   pipeline   stage s+1 (and, on a chunk's first row, the next chunk's halo) is in flight while stage s multiplies:
//              one vmcnt(0) + one barrier per stage of 24..48 MFMAs per wave
//   grid       1-D, XCD-aware: workgroup id mod 8 picks the XCD (round-robin dispatch), and each XCD walks a contiguous
//              band of the image in raster order, so the halo rows/columns shared by neighbouring tiles are served by
//              that XCD's L2 instead of HBM
// Epilogue as the fp32 kernel: bias, optional ReLU (folded eval), fp32 NHWC store (optionally accumulating), per-tile
// BatchNorm partial statistics, or raw split-K partial sums.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define HPRI_MAXSEG 4

struct ConvV2Args {
  const __bf16* xp; long long x_plane;   // activation planes: plane p at xp + p*x_plane (elements)
  int x_cs, x_coff;                      // elements per pixel (multiple of 32), first channel (multiple of 8)
  const __bf16* wp;                      // packed weights [chunk][tap][plane][Cout_pad][32] (hpri_pack_weight_bf16)
  const float* bias;
  float* y; int y_cs; int y_coff;
  float4* stats;
  int N, H, W, Cin_pad, Cout, Cout_pad, y_cw, accumulate, relu;
  int ksplit; float* ws;
  unsigned long long* stamps;            // diagnostic builds (-DHPRI_STAMPS) only: [workgroup][2 waves][8] s_memtime values
  int nseg, tiles_img, ntiles, nb_count;
  int seg_twl[HPRI_MAXSEG], seg_xbeg[HPRI_MAXSEG], seg_ntx[HPRI_MAXSEG], seg_first[HPRI_MAXSEG];
};

#ifdef HPRI_STAMPS
#define STAMP(i_)                                                                                          \
  {                                                                                                        \
    unsigned long long t_;                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                            \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if (a.stamps != nullptr && (threadIdx.x & 255) == 0)                                                   \
      a.stamps[((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 2 + (threadIdx.x >> 8)) * 8 + (i_)] = t_; \
  }
#else
#define STAMP(i_)
#endif

// Geometry of one work item (pixel tile x channel block); wave-uniform.
struct V2Tile { int img, y0, x0, xlim, twl, nb, bx; };

template <int KS, int WN, int SPLIT>
__global__ __launch_bounds__(512, 2) void conv_bf16v2_kernel(ConvV2Args a) {
  constexpr int T = KS * KS, PAD = KS / 2;
  constexpr int WM = 8 / WN;
  constexpr int BN = 64 * WN;
  constexpr int NPL = SPLIT + 1;
  static_assert(SPLIT == 0, "conv_bf16v2: the multi-plane products are not built yet");
  constexpr int TS = SPLIT ? 1 : KS;              // taps per stage
  constexpr int NST = SPLIT ? T : KS;             // stages per chunk
  constexpr int SR = TS * NPL;                    // [BN][32] row blocks per B stage
  constexpr int NIA = (WM == 8) ? 5 : 3;          // A DMA instructions per wave per plane per chunk (16 pixels each)
  constexpr int APIX = NIA * 8 * 16;              // staged pixel slots per plane (>= largest halo of this shape)
  constexpr int NPIECE = SR * BN / 16;            // 1-KB DMA pieces per B stage (12 for 64 channels, 24 for 128)
  constexpr int NB_E = (NPIECE + 7) / 8;          // ... issued by each of waves 0-3
  constexpr int NB_L = (NPIECE - 4 * NB_E) / 4;   // ... and by each of waves 4-7 (counted vmcnt waits differ per group)
  static_assert(4 * NB_E + 4 * NB_L == NPIECE, "B pieces must divide over the two wave groups");
  constexpr int NIB = NB_E;
  constexpr int NBB = 4;                          // B stage buffers: being read, landed, landing, being issued
  constexpr int A_BYTES = NPL * APIX * 64, B_BYTES = NPIECE * 1024;
  constexpr int NA_W = NPL * NIA;                 // vector-memory operations per wave per halo
  constexpr int EROWS = (WM == 8) ? 32 : 16;      // pixels of a 32x32 accumulator tile transposed through LDS at a time
  constexpr int ESCR = EROWS * 128;               // epilogue scratch per wave (bytes)
  static_assert(8 * ESCR + 8 * 128 * 4 + 64 <= A_BYTES, "epilogue scratch must fit the idle halo buffer");
  __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[2 * A_BYTES + NBB * B_BYTES];
  unsigned char* a_lds = smem_raw;
  unsigned char* b_lds = smem_raw + 2 * A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int nchunks_all = a.Cin_pad >> 5;
  const int cps = (nchunks_all + a.ksplit - 1) / a.ksplit;
  const int chunk0 = blockIdx.z * cps;
  const int nchunks = min(nchunks_all, chunk0 + cps);
  const int S0 = chunk0 * NST, S = nchunks * NST;

  // ---- persistent, XCD-aware work list: workgroup id mod 8 labels the XCD (round-robin dispatch); XCD x owns the
  // contiguous tile range [x*per_xcd, (x+1)*per_xcd) and its workgroups walk it in raster order, channel blocks of a
  // tile back to back.  Every workgroup runs a fixed list, so the grid drains without any inter-workgroup protocol.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int per_xcd = (a.ntiles + 7) >> 3;
  const int items = per_xcd * a.nb_count;

  auto tile_of = [&](int item, V2Tile& t) -> bool {
    if (item >= items) return false;
    const int tloc = item / a.nb_count;
    t.nb = item - tloc * a.nb_count;
    t.bx = xcd * per_xcd + tloc;
    if (t.bx >= a.ntiles) return false;
    t.img = t.bx / a.tiles_img;
    const int tin = t.bx - t.img * a.tiles_img;
    int seg = 0;
#pragma unroll
    for (int k = 1; k < HPRI_MAXSEG; ++k)
      if (k < a.nseg && tin >= a.seg_first[k]) seg = k;
    t.twl = a.seg_twl[seg];
    const int TW = 1 << t.twl, TH = 2 * WM * (32 >> t.twl);
    const int tt = tin - a.seg_first[seg];
    const int ty = tt / a.seg_ntx[seg], tx = tt - ty * a.seg_ntx[seg];
    t.y0 = ty * TH; t.x0 = a.seg_xbeg[seg] + tx * TW;
    t.xlim = min(a.W, a.seg_xbeg[seg] + a.seg_ntx[seg] * TW);
    return true;
  };

  // per-lane DMA source offsets of the item being loaded (B: packed-weight rows of channel block nb; A: halo pixels)
  // (buffer_load ... lds: 32-bit per-lane byte offsets against wave-uniform descriptors; halo pixels outside the image get an
  // offset beyond the descriptor's range and the hardware range check writes zeros into their LDS slots)
  constexpr unsigned OOB = 0xFFFFFFF0u;
  unsigned goff[NIB], aoff[NIA];
  int bpiece[NIB];
  {
    const bool late_ = wave >= 4;
#pragma unroll
    for (int q = 0; q < NIB; ++q)      // waves 0-3 take pieces [0, 4*NB_E), waves 4-7 the rest
      bpiece[q] = late_ ? (4 * NB_E + (q < NB_L ? q : 0) * 4 + (wave - 4)) : (q * 4 + wave);
  }
  const hpri_rsrc_t rs_b = HPRI_MAKE_RSRC(a.wp, 0x7FFFFF00);
  hpri_rsrc_t rs_a = HPRI_MAKE_RSRC(a.xp, 0x7FFFFF00);
  auto setup_loads = [&](const V2Tile& t) {
#pragma unroll
    for (int q = 0; q < NIB; ++q) {
      const int R = bpiece[q] * 16 + (lane >> 2);
      const int rb = R / BN, n = R - rb * BN;
      const int ls = (lane & 3) ^ ((n >> 2) & 3);
      goff[q] = (unsigned)((rb * a.Cout_pad + t.nb * BN + n) * 32 + ls * 8) * 2u;
    }
    const int TW = 1 << t.twl, TH = 2 * WM * (32 >> t.twl);
    const int HW = TW + KS - 1, HP = (TH + KS - 1) * HW;
    const unsigned hw_inv = (65536u + (unsigned)HW - 1u) / (unsigned)HW;   // exact for pix < 2048
#pragma unroll
    for (int q = 0; q < NIA; ++q) {
      const int pix = (q * 8 + wave) * 16 + (lane >> 2);
      unsigned off = OOB;
      if (pix < HP) {
        const int hy = (int)(((unsigned)pix * hw_inv) >> 16), hx = pix - hy * HW;
        const int iy = t.y0 + hy - PAD, ix = t.x0 + hx - PAD;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
          off = (unsigned)((iy * a.W + ix) * a.x_cs + (((lane & 3) ^ ((pix >> 2) & 3)) << 3)) * 2u;
      }
      aoff[q] = off;
    }
    rs_a = HPRI_MAKE_RSRC((a.xp + (size_t)t.img * a.H * a.W * a.x_cs + a.x_coff), 0x7FFFFF00);
  };
#define LOAD_B(s_)                                                                                                    \
  {                                                                                                                   \
    const int sb_ = (s_) * SR * a.Cout_pad * 64;                                  /* bytes */                        \
    unsigned char* lb_ = b_lds + ((s_) % NBB) * B_BYTES;                                                              \
    _Pragma("unroll") for (int q = 0; q < NIB; ++q)                                                                   \
        if (q < NB_L || !late)                                                                                        \
          HPRI_LDS_DMA16(rs_b, lb_ + bpiece[q] * 1024, goff[q], sb_);                                               \
  }
#define LOAD_A(chunk_)                                                                                                \
  {                                                                                                                   \
    unsigned char* la_ = a_lds + ((chunk_) & 1) * A_BYTES;                                                            \
    _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                                                                \
        _Pragma("unroll") for (int q = 0; q < NIA; ++q)                                                               \
            HPRI_LDS_DMA16(rs_a, la_ + pl * APIX * 64 + (q * 8 + wave) * 1024, aoff[q], (int)(pl * a.x_plane * 2) + (chunk_) * 64);   \
  }
#define PROLOGUE_LOADS()                       \
  LOAD_A(chunk0)                               \
  LOAD_B(S0)                                   \
  if (S0 + 1 < S) { LOAD_B(S0 + 1) }           \
  if (S0 + 2 < S) { LOAD_B(S0 + 2) }
#define WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define WAIT_VM2(e_, l_) do { if (late) WAIT_VM(l_); else WAIT_VM(e_); } while (0)   /* the wave groups issue different piece counts */
#define PHASE_BARRIER()                      \
  __builtin_amdgcn_sched_barrier(0);         \
  __builtin_amdgcn_s_barrier();              \
  __builtin_amdgcn_sched_barrier(0)

  const bool late = wave >= 4;
  const int bsw0 = ((0 + lh) ^ ((li >> 2) & 3)) * 16, bsw1 = ((2 + lh) ^ ((li >> 2) & 3)) * 16;
  const int b_base = (wn * 64 + li) * 64;

  V2Tile cur, nxt;
  int item = slot;
  bool have = tile_of(item, cur);
  if (!have) return;
  setup_loads(cur);
  PROLOGUE_LOADS()

  while (have) {
    STAMP(0)
    const int twl = cur.twl;
    const int TW = 1 << twl, RW = 32 >> twl;
    const int HW = TW + KS - 1;
    int hp0[2];                                  // lane -> halo pixel of its M-tile at tap (0,0)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) hp0[mt] = ((wm * 2 + mt) * RW + (li >> twl)) * HW + (li & (TW - 1));

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jn = 0; jn < 2; ++jn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

    // Pipeline (two wave groups, staggered by one phase).  A stage (one kernel row of one 32-channel chunk) is split
    // into a READ phase (its 24 fragments, LDS -> registers, then a counted vmcnt wait for this wave's DMA pieces of stage
    // s+1) and an MFMA phase (24 MFMAs from registers with the DMA pieces of stage s+3 -- and, on a chunk's first stage,
    // of the next chunk's halo -- issued between them), each closed by a workgroup barrier.  Waves 4-7, the second wave
    // on every SIMD, run one phase behind waves 0-3: on each SIMD one wave multiplies while its partner reads fragments,
    // so the matrix pipe and the LDS port are busy at the same time instead of taking turns.  (Measured alternatives,
    // tools/v2_stamps.py: both waves in lock-step with one barrier per stage 2490 cycles/stage, lock-step with register
    // double-buffered taps 2490, this form 2170, MFMA-bound 1536.)
    // Hazards: a wave waits for its own pieces of a stage (vmcnt counts in issue order) at least one all-wave barrier
    // before anyone reads that stage; a buffer is refilled no earlier than two phases after its last fragment read,
    // which lgkmcnt(0) retires before the closing barrier.  The first wait also retires the previous item's output
    // stores (issued after this item's prologue loads).
    if (S0 + 2 < S) { WAIT_VM2(2 * NB_E, 2 * NB_L); } else if (S0 + 1 < S) { WAIT_VM2(NB_E, NB_L); } else { WAIT_VM(0); }
    PHASE_BARRIER();                           // stage S0 and its halo are visible to every wave
    if (late) { PHASE_BARRIER(); }             // the stagger
    STAMP(1)
    // byte offsets of this lane's A fragments inside a halo buffer for the 9 taps (k16-step 0; step 1 is offset ^ 32):
    // the swizzle depends on the halo pixel, so they are tabulated once per item instead of re-derived per read
    int aofs[2][T];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int hp = hp0[mt] + (t / KS) * HW + (t % KS);
        aofs[mt][t] = hp * 64 + ((lh ^ ((hp >> 2) & 3)) << 4);
      }
    bool prev_a = false;                       // did the previous MFMA phase issue a halo?
#define STAGE(s_, st_)                                                                                                 \
    {                                                                                                                  \
      /* ---------------- READ phase ---------------- */                                                               \
      const unsigned char* ab = a_lds + (chunk & 1) * A_BYTES;                                                         \
      const unsigned char* bb = b_lds + ((s_) % NBB) * B_BYTES + b_base;                                               \
      bf16x8 af[KS][2][2], bf[KS][2][2];                                                                               \
      _Pragma("unroll") for (int dx = 0; dx < KS; ++dx) {                                                              \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                               \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                           \
                af[dx][kk][mt] = *reinterpret_cast<const bf16x8*>(ab + (aofs[mt][(st_) * KS + dx] ^ (kk << 5)));       \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                               \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                           \
                bf[dx][kk][nt] = *reinterpret_cast<const bf16x8*>(bb + (dx * BN + nt * 32) * 64 + (kk ? bsw1 : bsw0)); \
      }                                                                                                                \
      /* own pieces of stage s+1 landed: only what the previous MFMA phase issued (stage s+2, a halo) may be pending */ \
      if ((s_) + 2 < S) { if (prev_a) WAIT_VM2(NB_E + NA_W, NB_L + NA_W); else WAIT_VM2(NB_E, NB_L); }                 \
      else              { if (prev_a) WAIT_VM(NA_W); else WAIT_VM(0); }                                                \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
      PHASE_BARRIER();                                                                                                 \
      /* ---------------- MFMA phase ---------------- */                                                               \
      const bool more_b = (s_) + 3 < S, more_a = ((st_) == 0 && chunk + 1 < nchunks);                                  \
      prev_a = more_a;                                                                                                 \
      const int sb_ = ((s_) + 3) * SR * a.Cout_pad * 64;                                                               \
      unsigned char* lb_ = b_lds + (((s_) + 3) % NBB) * B_BYTES;                                                       \
      unsigned char* la_ = a_lds + ((chunk + 1) & 1) * A_BYTES;                                                        \
      __builtin_amdgcn_s_setprio(1);                                                                                   \
      _Pragma("unroll") for (int m = 0; m < KS * 8; ++m) {                                                             \
        const int dx = m >> 3, kk = (m >> 2) & 1, mt = (m >> 1) & 1, nt = m & 1;                                       \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[dx][kk][mt], bf[dx][kk][nt], acc[mt][nt], 0, 0, 0);   \
        if (m % 3 == 1) {                      /* after MFMAs 1, 4, 7, ...: one DMA piece (8 slots per stage) */       \
          const int k = m / 3;                                                                                         \
          if (k < NIB) {                                                                                               \
            if (more_b && (k < NB_L || !late))                                                                         \
              HPRI_LDS_DMA16(rs_b, lb_ + bpiece[k < NIB ? k : 0] * 1024, goff[k < NIB ? k : 0], sb_);                          \
          } else if (k - NIB < NA_W) {                                                                                 \
            const int pl = (k - NIB) / NIA, q = (k - NIB) % NIA;                                                       \
            if (more_a)                                                                                                \
              HPRI_LDS_DMA16(rs_a, la_ + pl * APIX * 64 + (q * 8 + wave) * 1024, aoff[q], (int)(pl * a.x_plane * 2) + (chunk + 1) * 64); \
          }                                                                                                            \
        }                                                                                                              \
      }                                                                                                                \
      __builtin_amdgcn_s_setprio(0);                                                                                   \
      PHASE_BARRIER();                                                                                                 \
    }
    static_assert(NIB + NA_W <= KS * 8 / 3, "more DMA pieces per stage than interleave slots");
    static_assert(NST == 3, "the chunk body is written for three stages");
    for (int chunk = chunk0; chunk < nchunks; ++chunk) {
      STAGE(chunk * NST + 0, 0)
      STAGE(chunk * NST + 1, 1)
      STAGE(chunk * NST + 2, 2)
    }
    if (!late) { PHASE_BARRIER(); }            // every wave has passed the same number of barriers; LDS is idle
#undef STAGE
#undef READ_TAP
    STAMP(2)

    // ---- the next item's first stages start to load now and land while this item's results are written out ----
    item += nslots;
    have = tile_of(item, nxt);
    if (have) {
      setup_loads(nxt);
      PROLOGUE_LOADS()
    }
    STAMP(5)

    // ------------------------------- epilogue -------------------------------
    // acc[mt][nt][r]: M-tile pixel m = (r&3) + 8*(r>>2) + 4*lh, channel nb*BN + wn*64 + nt*32 + li.  Each 32x32 tile goes
    // through a per-wave LDS scratch (the halo buffer the next item's prologue does not use) so that every lane stores
    // 16 contiguous bytes: 4x fewer store instructions than the accumulator layout allows directly.
    {
      const int TH = 2 * WM * RW;
      const bool raw = a.ksplit > 1;             // split-K: raw partial sums into the workspace slab of this K slice
      float* dst = raw ? a.ws + (size_t)blockIdx.z * ((size_t)a.N * a.H * a.W) * a.Cout_pad : a.y;
      const int dcs = raw ? a.Cout_pad : a.y_cs, dco = raw ? 0 : a.y_coff, dcw = raw ? a.Cout_pad : a.y_cw;
      float* scr = reinterpret_cast<float*>(a_lds + ((chunk0 & 1) ^ 1) * A_BYTES + wave * ESCR);
      float* red = reinterpret_cast<float*>(a_lds + ((chunk0 & 1) ^ 1) * A_BYTES + 8 * ESCR);   // [8 waves][64 channels]
      if (!raw) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int n = cur.nb * BN + wn * 64 + nt * 32 + li;
          const float b = (a.bias != nullptr && n < a.Cout) ? a.bias[n] : 0.f;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[mt][nt][r] += b; if (a.relu) acc[mt][nt][r] = fmaxf(acc[mt][nt][r], 0.f); }
        }
      }
      // store side: lane -> pixel (lane>>3) of an 8-pixel group, channels 4*(lane&7) .. +3
      const int sp = lane >> 3, sc = (lane & 7) * 4;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int nbase = cur.nb * BN + wn * 64 + nt * 32;
          const bool chan_ok = raw || (nbase + li < a.Cout);
#pragma unroll
          for (int hlf = 0; hlf < 32 / EROWS; ++hlf) {
            // accumulator -> scratch[pixel][channel] (pixels hlf*EROWS .. +EROWS of the tile)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
              if (EROWS == 32 || (m / EROWS) == hlf)
                scr[(m % EROWS) * 32 + li] = chan_ok ? acc[mt][nt][r] : 0.f;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private scratch: no barrier needed
#pragma unroll
            for (int it = 0; it < EROWS / 8; ++it) {
              const int m = hlf * EROWS + it * 8 + sp;
              const int iy = cur.y0 + (wm * 2 + mt) * RW + (m >> twl), ix = cur.x0 + (m & (TW - 1));
              f32x4 v = *reinterpret_cast<const f32x4*>(scr + (it * 8 + sp) * 32 + sc);
              if (iy < a.H && ix < cur.xlim && nbase + sc < dcw) {
                float* p = dst + ((size_t)(cur.img * a.H + iy) * a.W + ix) * dcs + dco + nbase + sc;
                if (!raw && a.accumulate) v += *reinterpret_cast<const f32x4*>(p);
                *reinterpret_cast<f32x4*>(p) = v;
              }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // scratch is rewritten by the next tile
          }
        }
      }
      STAMP(3)
      if (!raw && a.stats != nullptr) {
        // per-tile, per-channel (mean, M2, count).  Each wave makes an exact two-pass record of its own 64 pixels (sum,
        // then squared deviations from its own mean: no E[x^2]-E[x]^2 cancellation, no barrier); the WM wave records of a
        // channel are merged with Chan's update by one wave after a single barrier.
        const bool interior = (cur.y0 + TH <= a.H) && (cur.x0 + TW <= cur.xlim);
        unsigned vmask = 0xffffffffu;          // valid-pixel mask of this lane's 2 x 16 accumulator rows
        if (!interior) {
          vmask = 0u;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
              const int iy = cur.y0 + (wm * 2 + mt) * RW + (m >> twl), ix = cur.x0 + (m & (TW - 1));
              if (iy < a.H && ix < cur.xlim) vmask |= 1u << (mt * 16 + r);
            }
        }
        int nv = __builtin_popcount(vmask);
        nv += __shfl_xor(nv, 32);              // valid pixels of this wave's 64
        const float cntw = (float)nv, inv = nv > 0 ? 1.f / cntw : 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float s1 = 0.f;
          if (interior) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int r = 0; r < 16; ++r) s1 += acc[mt][nt][r];
          } else {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int r = 0; r < 16; ++r) s1 += ((vmask >> (mt * 16 + r)) & 1u) ? acc[mt][nt][r] : 0.f;
          }
          s1 += __shfl_xor(s1, 32);
          const float mw = s1 * inv;
          float s2 = 0.f;
          if (interior) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int r = 0; r < 16; ++r) { const float d = acc[mt][nt][r] - mw; s2 += d * d; }
          } else {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int r = 0; r < 16; ++r) { const float d = acc[mt][nt][r] - mw; s2 += ((vmask >> (mt * 16 + r)) & 1u) ? d * d : 0.f; }
          }
          s2 += __shfl_xor(s2, 32);
          if (lh == 0) {
            red[(wave * 64 + nt * 32 + li) * 2 + 0] = mw;
            red[(wave * 64 + nt * 32 + li) * 2 + 1] = s2;
          }
        }
        if (lane == 0) red[8 * 128 + wave] = cntw;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS only: the vector-memory queue (output stores, the next
        PHASE_BARRIER();                                     // item's prologue loads) keeps draining
        // merge: channel c of the block is handled by lane c % (BN/8) of wave c / (BN/8) -- all eight waves share the work
        if (lane < BN / 8) {
          const int c = wave * (BN / 8) + lane, wn_c = c >> 6, cc = c & 63;
          float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
          for (int m = 0; m < WM; ++m) {
            const int w2 = m * WN + wn_c;
            const float nb_ = red[8 * 128 + w2];
            if (nb_ > 0.f) {
              const float mb = red[(w2 * 64 + cc) * 2 + 0], qb = red[(w2 * 64 + cc) * 2 + 1];
              const float tot = n + nb_, delta = mb - mean, f = __builtin_amdgcn_rcpf(tot) * nb_;
              mean += delta * f;
              m2 += qb + delta * delta * (n * f);
              n = tot;
            }
          }
          a.stats[(size_t)cur.bx * a.Cout_pad + cur.nb * BN + c] = make_float4(mean, m2, n, 0.f);
        }
      }
      STAMP(4)
    }
    cur = nxt;
  }
#undef LOAD_A
#undef LOAD_B
#undef PROLOGUE_LOADS
#undef WAIT_VM
#undef WAIT_VM2
#undef PHASE_BARRIER
}

// ---- host side -------------------------------------------------------------------------------------------------------
// column bands of tile width 32 / 16 (/ 8 for the 256-pixel shape): same cover as conv_fwd.hip's conv_segments
struct V2Segs { int nseg, tiles_img, twl[HPRI_MAXSEG], xbeg[HPRI_MAXSEG], ntx[HPRI_MAXSEG], first[HPRI_MAXSEG]; };
static V2Segs v2_segments(int H, int W, int wm) {
  const int min_tw = (wm == 8) ? 16 : 8;         // halo of narrower tiles would not fit the staged pixel slots
  auto th = [&](int tw) { return 2 * wm * (32 / tw); };
  auto slots = [&](int tw, int ntx) { return (long long)hpri_cdiv(H, th(tw)) * th(tw) * tw * ntx; };
  V2Segs plain{}; plain.nseg = 1; plain.twl[0] = 5; plain.xbeg[0] = 0; plain.ntx[0] = hpri_cdiv(W, 32);
  const long long cost_plain = slots(32, plain.ntx[0]);
  V2Segs g{}; long long cost_g = 0; int x = 0;
  if (W / 32 > 0) { g.twl[0] = 5; g.xbeg[0] = 0; g.ntx[0] = W / 32; cost_g += slots(32, W / 32); x = (W / 32) * 32; g.nseg = 1; }
  int rem = W - x;
  for (int tw = 16; tw >= min_tw && rem > 0; tw >>= 1) {
    int n = rem / tw;
    if (tw == min_tw && rem % tw) n += 1;
    if (n > 0 && g.nseg < HPRI_MAXSEG) {
      int l = 0; while ((1 << l) < tw) ++l;
      g.twl[g.nseg] = l; g.xbeg[g.nseg] = x; g.ntx[g.nseg] = n; cost_g += slots(tw, n);
      x += n * tw; rem = W - x; g.nseg++;
    }
  }
  V2Segs r = (g.nseg > 0 && rem <= 0 && cost_g < cost_plain) ? g : plain;
  int first = 0;
  for (int k = 0; k < r.nseg; ++k) { r.first[k] = first; first += hpri_cdiv(H, th(1 << r.twl[k])) * r.ntx[k]; }
  r.tiles_img = first;
  return r;
}

static inline int v2_wn(int Cout_pad) { return (Cout_pad % 128 == 0) ? 2 : 1; }

// Split-K (host only): one workgroup per CU, so a grid that is not close to a multiple of 256 workgroups wastes whole
// rounds; K is cut (<= 4 ways, >= 4 chunks per slice) where that brings workgroups per CU closer to an integer.
// K slices for layers with few items.  A slice costs HBM traffic, not only a launch: the convolution writes k fp32 slabs
// instead of one output and splitk_finish reads them back and writes the output (2 k output sizes more than k = 1), against
// the whole rounds of 256 workgroups it buys.  Measured on the C2 step, interleaved on one box: the round-efficiency rule
// alone (k = 3-4 on every layer below 304 x 484: more slab traffic than arithmetic at 152 x 242) 138.7 cubes/s, slices priced
// at 850 TFLOP/s and 5 TB/s 145.2, no slices at all 147.7 -- with half a round or more of items (38 x 60: 144) the idle CUs
// of the last round are cheaper than the slabs, and in backward the weight gradients of the second stream use them anyway.
// Slices remain for problems below half a round, priced as above.
static int v2_ksplit(int N, int H, int W, int Cin_pad, int Cout_pad) {
  const int wn = v2_wn(Cout_pad);
  const long long blocks = (long long)N * v2_segments(H, W, 8 / wn).tiles_img * (Cout_pad / (64 * wn));
  const int nchunks = Cin_pad / 32;
  if (blocks >= 128) return 1;
  const double t_compute = 2.0 * N * H * W * (double)Cin_pad * Cout_pad * 9.0 / 850e12;
  const double out_bytes = 4.0 * N * H * W * (double)Cout_pad;
  int best = 1; double best_t = 1e30;
  for (int k = 1; k <= 4; ++k) {
    if (k > 1 && nchunks / k < 4) break;
    const double per_cu = (double)blocks * k / 256.0;
    double eff = per_cu / (double)((long long)(per_cu + 0.999999));
    const double t = t_compute / eff + (k > 1 ? 2.0 * k * out_bytes / 5e12 + 4e-6 : 0.0);
    if (t < best_t - 1e-12) { best_t = t; best = k; }
  }
  return best;
}

#define SK_PIX 64
extern "C" int hpri_conv_bf16v2_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int* ksplit, int* stat_tiles,
                                     size_t* ws_floats) {
  const int k = v2_ksplit(N, H, W, Cin_pad, Cout_pad);
  *ksplit = k;
  if (k > 1) { *stat_tiles = N * hpri_cdiv(H * W, SK_PIX); *ws_floats = (size_t)k * N * H * W * Cout_pad; }
  else { *stat_tiles = N * v2_segments(H, W, 8 / v2_wn(Cout_pad)).tiles_img; *ws_floats = 0; }
  return HPRI_OK;
}

// conv_fwd.hip
extern "C" int hpri_splitk_finish(const float* ws, int ksplit, int Cout_pad, const float* bias, float* y, int y_cs, int y_coff,
                                  float* stats, int N, int HW, int Cout, int y_cw, int accumulate, int relu, hipStream_t stream);

template <int WN, int SPLIT>
static int launch_v2(ConvV2Args& a, hipStream_t stream) {
  const V2Segs sg = v2_segments(a.H, a.W, 8 / WN);
  a.nseg = sg.nseg; a.tiles_img = sg.tiles_img; a.ntiles = a.N * sg.tiles_img; a.nb_count = a.Cout_pad / (64 * WN);
  for (int k = 0; k < HPRI_MAXSEG; ++k) { a.seg_twl[k] = sg.twl[k]; a.seg_xbeg[k] = sg.xbeg[k]; a.seg_ntx[k] = sg.ntx[k]; a.seg_first[k] = sg.first[k]; }
  // persistent workgroups: one per CU (256 / ksplit per K slice, a multiple of 8 so that id mod 8 labels the XCD), each
  // walking a fixed list of items
  const int per_xcd = hpri_cdiv(a.ntiles, 8);
  int slots = 32 / a.ksplit;
  if (slots < 1) slots = 1;
  if (slots > per_xcd * a.nb_count) slots = per_xcd * a.nb_count;
  dim3 grid((unsigned)(slots * 8), 1u, (unsigned)a.ksplit);
  hipLaunchKernelGGL((conv_bf16v2_kernel<3, WN, SPLIT>), grid, dim3(512), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_conv_bf16v2_dbg(const void* xp, long long x_plane, int x_cs, int x_coff, const void* wp, const float* bias,
                                    float* y, int y_cs, int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout,
                                    int Cout_pad, int y_cw, int accumulate, int split, float* ws, size_t ws_floats,
                                    unsigned long long* stamps, hipStream_t stream);

// 3x3 pad-1 convolution (forward, or data gradient with the flipped pack) over bf16 activation planes.
extern "C" int hpri_conv_bf16v2(const void* xp, long long x_plane, int x_cs, int x_coff, const void* wp, const float* bias,
                                float* y, int y_cs, int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout,
                                int Cout_pad, int y_cw, int accumulate, int split, float* ws, size_t ws_floats,
                                hipStream_t stream) {
  return hpri_conv_bf16v2_dbg(xp, x_plane, x_cs, x_coff, wp, bias, y, y_cs, y_coff, stats, N, H, W, Cin_pad, Cout, Cout_pad,
                              y_cw, accumulate, split, ws, ws_floats, nullptr, stream);
}

// the same with a stamp buffer ([workgroups][2][8] u64) for diagnostic builds (tools/v2_stamps.py); ignored otherwise
extern "C" int hpri_conv_bf16v2_dbg(const void* xp, long long x_plane, int x_cs, int x_coff, const void* wp, const float* bias,
                                    float* y, int y_cs, int y_coff, float* stats, int N, int H, int W, int Cin_pad, int Cout,
                                    int Cout_pad, int y_cw, int accumulate, int split, float* ws, size_t ws_floats,
                                    unsigned long long* stamps, hipStream_t stream) {
  HPRI_REQUIRE(xp && wp && y, "conv_bf16v2: null pointer");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0, "conv_bf16v2: empty image");
  HPRI_REQUIRE(Cin_pad > 0 && Cin_pad % 32 == 0, "conv_bf16v2: Cin_pad must be a positive multiple of 32");
  HPRI_REQUIRE(Cout_pad % 64 == 0 && Cout <= Cout_pad && Cout > 0, "conv_bf16v2: Cout_pad must be a multiple of 64 >= Cout");
  HPRI_REQUIRE(x_cs % 8 == 0 && x_coff % 8 == 0 && x_coff + Cin_pad <= x_cs, "conv_bf16v2: plane channel stride/offset must be multiples of 8 and hold Cin_pad channels");
  HPRI_REQUIRE(((uintptr_t)xp & 15) == 0 && ((uintptr_t)wp & 15) == 0 && (x_plane % 8) == 0, "conv_bf16v2: pointers must be 16-byte aligned");
  HPRI_REQUIRE((long long)H * W * x_cs * 2 < 0x7FFFFF00ll, "conv_bf16v2: one image of the input planes exceeds 2 GiB (32-bit DMA offsets)");
  HPRI_REQUIRE((long long)(Cin_pad / 32) * 9 * Cout_pad * 64 < 0x7FFFFF00ll, "conv_bf16v2: packed weights exceed 2 GiB");
  HPRI_REQUIRE(split == 0, "conv_bf16v2: only plain bf16 planes (split 0) are built in this version");
  ConvV2Args a;
  a.xp = reinterpret_cast<const __bf16*>(xp); a.x_plane = x_plane; a.x_cs = x_cs; a.x_coff = x_coff;
  a.wp = reinterpret_cast<const __bf16*>(wp); a.bias = bias; a.y = y; a.y_cs = y_cs; a.y_coff = y_coff;
  a.stats = reinterpret_cast<float4*>(stats); a.stamps = stamps;
  a.N = N; a.H = H; a.W = W; a.Cin_pad = Cin_pad; a.Cout = Cout; a.Cout_pad = Cout_pad;
  a.y_cw = y_cw < Cout ? Cout : y_cw; a.accumulate = accumulate & 1; a.relu = (accumulate >> 1) & 1;
  HPRI_REQUIRE(a.y_cw + y_coff <= y_cs, "conv_bf16v2: output channels exceed the channel stride");
  a.ksplit = v2_ksplit(N, H, W, Cin_pad, Cout_pad);
  a.ws = ws;
  if (a.ksplit > 1) {
    if (ws == nullptr || (size_t)a.ksplit * N * H * W * Cout_pad > ws_floats)
      return hpri_set_error(HPRI_ERR_WORKSPACE, "conv_bf16v2: split-K workspace too small (see hpri_conv_bf16v2_plan)");
    a.stats = nullptr; a.accumulate = 0;
  }
  const int rc = (v2_wn(Cout_pad) == 2) ? launch_v2<2, 0>(a, stream) : launch_v2<1, 0>(a, stream);
  if (rc != HPRI_OK || a.ksplit == 1) return rc;
  return hpri_splitk_finish(ws, a.ksplit, Cout_pad, bias, y, y_cs, y_coff, stats, N, H * W, Cout, a.y_cw, accumulate & 1, a.relu, stream);
}
