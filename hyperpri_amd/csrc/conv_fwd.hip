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
#include <stdlib.h>

#define HPRI_MAXSEG 4
struct ConvFwdArgs {
  const float* x; int x_cs; int x_coff;
  const float* wp;       // packed weights [chunks][T][32][Cout_pad]
  const float* bias;     // [Cout] or nullptr
  float* y; int y_cs; int y_coff;
  float4* stats;         // [N*tiles_img][Cout_pad] (mean, M2, count, 0) or nullptr
  int N, H, W;           // GEMM-M image: output pixels (DIRECT) / low-res pixels (S2D, D2S)
  int Cin_pad;           // K per tap, multiple of 8 (for S2D: 4*Cup)
  int Cout;              // valid output channels (for D2S: 4*Cup)
  int Cout_pad;          // multiple of BN
  int y_cw;              // channels written (>= Cout; extra ones get zeros)
  int accumulate;        // y += result instead of y = result
  int relu;              // epilogue max(.,0): eval-mode conv+BN+ReLU with BN folded into weights and bias
  int H2, W2, py0, px0, Cup;  // S2D / D2S geometry: hi-res image dims, pad offsets, channels per tap
  int ksplit;            // >1: blockIdx.z takes a slice of the K chunks and stores raw partial sums to ws
  float* ws;             // [ksplit][N*H*W][Cout_pad] partial sums (split-K only)
  h16_t* plp; int pl_cs, pl_coff;   // D2S only: the scattered result ALSO (y != nullptr) or ONLY (y == nullptr) as one bf16 plane
  // Tile segments: the image width is cut into column bands of tile width 32, 16, 8 or 4 (tile height grows as the
  // width shrinks, pixels per tile stay constant) so that W = 484 / 242 / 121 does not round up to 512 / 256 / 128.
  int nseg, tiles_img;
  int nbx;               // >0: XCD-aware 1-D grid (see conv_block_ids); = number of output-channel blocks
  int seg_twl[HPRI_MAXSEG];    // log2(tile width)
  int seg_xbeg[HPRI_MAXSEG];   // first column of the band
  int seg_ntx[HPRI_MAXSEG];    // tiles per row of the band
  int seg_first[HPRI_MAXSEG];  // index (within one image) of the band's first tile
};

// Block -> (pixel tile, output-channel block).  Legacy: grid (tiles, NB): channel blocks of one pixel tile are a whole
// grid row apart, so with many channel blocks and activations far larger than L2 (SpectralUNET, F = 1650: 13 blocks,
// 2.8 GB) every block re-reads its input tile from HBM.  nbx > 0: 1-D grid; workgroups go round-robin over the 8 XCDs
// in launch order, so id%8 picks the XCD and consecutive ids on one XCD walk the channel blocks of ONE pixel tile:
// the tile is fetched from HBM once and served to the other NB-1 blocks by that XCD's L2.
__device__ __forceinline__ bool conv_block_ids(const ConvFwdArgs& a, int& bx, int& nb) {
  bx = blockIdx.x; nb = blockIdx.y;
  if (a.nbx > 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    nb = j % a.nbx;
    bx = (j / a.nbx) * 8 + xcd;
    if (bx >= a.N * a.tiles_img) return false;
  }
  return true;
}


template <int KS, int WM, int WN, int AMODE, int EPI>
__global__ __launch_bounds__(256, (WM == 2) ? 3 : 2) void conv_fwd_kernel(ConvFwdArgs a) {
  constexpr int T = KS * KS, PAD = KS / 2;
  constexpr int TPIX = 64 * WM;                  // output pixels per tile
  // largest staged halo over the tile kinds this configuration may use (WM=2: widths 32..4, WM=4: 32..8)
  constexpr int MAXHP = (KS == 1) ? TPIX : (WM == 2 ? 6 * 34 : 10 * 34);
  constexpr int BN = 64 * WN;
  constexpr int CS = 36;                         // dwords per staged pixel (32 channels + 4 pad)
  constexpr int NLD_A = (MAXHP * 8 + 255) / 256; // float4 loads per thread per A chunk
  constexpr int NLD_B = (32 * BN / 4) / 256;     // float4 loads per thread per B panel
  // Panel buffers.  The 2x2 shape keeps ONE (45 KB of LDS, 168 VGPRs): three workgroups then share a CU, and a third
  // workgroup covers the panel's DMA latency better than the double buffer inside the workgroup did (+1..4 %, bit-identical).
  // (1x1 forms: the staged tile is 128 pixels, 18 KB, so TWO panel buffers still leave room for three workgroups per CU (50 KB each) and the
  //  panel of stage s+1 lands under the MFMAs of stage s instead of behind a second barrier: round 4)
  constexpr int NBUF = (WM == 2 && KS == 3) ? 1 : 2;
  __shared__ __attribute__((aligned(16))) float smem[MAXHP * CS + NBUF * 32 * BN];
  float* a_lds = smem;
  float* b_lds = smem + MAXHP * CS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  int bx, nb;
  if (!conv_block_ids(a, bx, nb)) return;

  // ---- which tile: image, column band (tile kind), position ----
  const int img = bx / a.tiles_img;
  const int tin = bx - img * a.tiles_img;
  int seg = 0;
#pragma unroll
  for (int k = 1; k < HPRI_MAXSEG; ++k)
    if (k < a.nseg && tin >= a.seg_first[k]) seg = k;
  const int twl = a.seg_twl[seg];                // log2 tile width
  const int TW = 1 << twl, RW = 32 >> twl;       // an MFMA M-tile (32 pixels) is RW rows x TW columns
  const int TH = 2 * WM * RW;                    // tile rows (each wave: 2 M-tiles stacked)
  const int HW = TW + KS - 1, HP = (TH + KS - 1) * HW;
  const int tt = tin - a.seg_first[seg];
  const int ty = tt / a.seg_ntx[seg], tx = tt - ty * a.seg_ntx[seg];
  const int y0 = ty * TH, x0 = a.seg_xbeg[seg] + tx * TW;
  const int xlim = min(a.W, a.seg_xbeg[seg] + a.seg_ntx[seg] * TW);   // columns >= xlim belong to the next band

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks_all = (a.Cin_pad + 31) >> 5;
  const int cps = (nchunks_all + a.ksplit - 1) / a.ksplit;          // chunks per K split
  const int chunk0 = blockIdx.z * cps;
  const int nchunks = min(nchunks_all, chunk0 + cps);               // this block runs chunks [chunk0, nchunks)
  const int S0 = chunk0 * T, S = nchunks * T;

  // ---- B panels: per-thread offsets computed once; per panel only the panel base moves ----
  const float* wpanel = a.wp + (size_t)nb * BN;
  // LDS-DMA: panel s goes straight from global memory into LDS buffer s&1 (no VGPR staging, no ds_write); wave w's
  // p-th instruction fills floats [(p*4+w)*256, +256) of the panel, lane l the 4 floats at +4*l
  int goff[NLD_B];
#pragma unroll
  for (int p = 0; p < NLD_B; ++p) {
    const int e = (p * 4 + wave) * 256 + lane * 4;
    goff[p] = (e / BN) * a.Cout_pad + (e % BN);
  }
#define LOAD_PANEL(s_)                                                                              \
  {                                                                                                 \
    const float* pb_ = wpanel + (size_t)(s_) * 32 * a.Cout_pad;                                     \
    float* lb_ = b_lds + ((s_) % NBUF) * 32 * BN;                                                   \
    _Pragma("unroll") for (int p = 0; p < NLD_B; ++p)                                               \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb_ + goff[p]),          \
                                         (__attribute__((address_space(3))) void*)(lb_ + (p * 4 + wave) * 256), 16, 0, 0); \
  }
#define STORE_PANEL(buf_)

  // ---- A halo: per-thread offsets inside this image (element units, < 2^31) computed ONCE per tile; -1 = zero fill ----
  const float* ximg = a.x + (size_t)img * (AMODE == HPRI_A_DIRECT ? (size_t)a.H * a.W : (size_t)a.H2 * a.W2) * a.x_cs;
  // slot f = tid + p*256 -> halo pixel f>>3, 16-byte quad f&7 of the 32-channel chunk
  int aoff[NLD_A];
  {
    const unsigned hw_inv = (65536u + (unsigned)HW - 1u) / (unsigned)HW;   // exact for pix < 2048
#pragma unroll
    for (int p = 0; p < NLD_A; ++p) {
      const int f = tid + p * 256;
      const int pix = f >> 3, q = f & 7;
      int off = -1;
      if (pix < HP) {
        const int hy = (int)(((unsigned)pix * hw_inv) >> 16), hx = pix - hy * HW;
        const int iy = y0 + hy - PAD, ix = x0 + hx - PAD;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
          if (AMODE == HPRI_A_DIRECT) off = (iy * a.W + ix) * a.x_cs + a.x_coff + q * 4;
          else off = ((2 * iy + a.py0) * a.W2 + 2 * ix + a.px0) * a.x_cs + a.x_coff;  // + tap/chan below
        }
      }
      aoff[p] = off;
    }
  }
  f32x4 areg[NLD_A];
#define LOAD_A(c0_)                                                                                   \
  {                                                                                                   \
    const int kq = min(8, (a.Cin_pad - (c0_)) >> 2); /* valid float4 per pixel in this chunk */       \
    _Pragma("unroll") for (int p = 0; p < NLD_A; ++p) {                                               \
      const int q = (tid + p * 256) & 7;                                                              \
      f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                                 \
      if (aoff[p] >= 0 && q < kq) {                                                                   \
        if (AMODE == HPRI_A_DIRECT) {                                                                 \
          v = *reinterpret_cast<const f32x4*>(ximg + (size_t)(unsigned)aoff[p] + (c0_));             \
        } else { /* S2D: k = tap*Cup + co ; source pixel (2*iy + t_y + py0, 2*ix + t_x + px0) */      \
          const int k4 = (c0_) + q * 4;                                                               \
          const int tp = k4 / a.Cup, co = k4 - tp * a.Cup;                                            \
          v = *reinterpret_cast<const f32x4*>(ximg + (size_t)(unsigned)aoff[p] +                      \
                                              ((tp >> 1) * a.W2 + (tp & 1)) * a.x_cs + co);           \
        }                                                                                             \
      }                                                                                               \
      areg[p] = v;                                                                                    \
    }                                                                                                 \
  }
#define STORE_A()                                                                                     \
  _Pragma("unroll") for (int p = 0; p < NLD_A; ++p) {                                                 \
    const int f = tid + p * 256;                                                                      \
    if ((f >> 3) < HP) *reinterpret_cast<f32x4*>(a_lds + (f >> 3) * CS + (f & 7) * 4) = areg[p];      \
  }

  // lane -> pixel of its M-tile: row li >> twl, column li & (TW-1)
  const int a_base = ((wm * 2 * RW + (li >> twl)) * HW + (li & (TW - 1))) * CS + lh * 4;
  const int a_mt = RW * HW * CS;                 // second M-tile of the wave: RW rows further down
  const int b_base = lh * 4 * BN + wn * 64 + li;

#define MFMA_GROUP(g_)                                                                                \
  {                                                                                                   \
    f32x4 af[2];                                                                                      \
    float bf[2][4];                                                                                   \
    _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                  \
        af[mt] = *reinterpret_cast<const f32x4*>(ap + mt * a_mt + (g_) * 8);                          \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) bf[nt][j] = bp[((g_) * 8 + j) * BN + nt * 32];  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                     \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                              \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                          \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][j], bf[nt][j], acc[mt][nt], 0, 0, 0); \
  }

  if (NBUF > 1) { LOAD_PANEL(S0) }
  LOAD_A(chunk0 * 32)
  for (int s = S0; s < S; ++s) {
    const int chunk = s / T, tap = s - chunk * T;
    if (tap == 0) {
      __syncthreads();                 // everyone is done reading the previous A chunk
      STORE_A()
    }
    if (NBUF == 1) {
      __syncthreads();
      if (tap == T - 1 && chunk + 1 < nchunks) { LOAD_A((chunk + 1) * 32) }
      LOAD_PANEL(s)
      __syncthreads();
    } else {
      STORE_PANEL(s & 1)
      __syncthreads();                   // panel s (and the A chunk) visible
      if (s + 1 < S) { LOAD_PANEL(s + 1) }
      if (tap == T - 1 && chunk + 1 < nchunks) { LOAD_A((chunk + 1) * 32) }   // lands during this panel's MFMAs
    }

    const int kg = min(4, (a.Cin_pad - chunk * 32) >> 3);
    const int dy = tap / KS, dx = tap - dy * KS;
    const float* ap = a_lds + a_base + (dy * HW + dx) * CS;
    const float* bp = b_lds + (s % NBUF) * 32 * BN + b_base;
    for (int g = 0; g < kg; ++g) MFMA_GROUP(g)
  }
#undef MFMA_GROUP
#undef LOAD_A
#undef STORE_A
#undef LOAD_PANEL
#undef STORE_PANEL

#define HPRI_EPI_NTW 2
#include "conv_fwd_epilogue.inc"
#undef HPRI_EPI_NTW
}

// -------------------------------------------------------------------------------------------------------------
// bf16 variant (precision mode "bf16": BASELINE.json config C5 names bf16 MFMA): same tiling, bands, split-K and
// epilogue, but operands are rounded to bf16 while they are staged into LDS and the contraction runs on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate, fp32 activations in HBM).  Differences to the fp32 kernel:
//   * A halo chunk in LDS: [pixel][32 ch bf16 + 8 pad] = 80-byte pitch (conflict-free ds_read_b128 of 8 k-values)
//   * B panels pre-packed bf16 [tap][(plane)][n][32 k] (k contiguous per output channel), brought in by LDS-DMA into an
//     unpadded, XOR-swizzled 64-byte-pitch layout
//   * one barrier per KERNEL ROW (KS taps x 32 channels = 6 k16-steps x 4 tiles = 24 MFMAs per wave), because a
//     bf16 MFMA retires 16x the flops of the fp32 one in half the cycles
//   * workgroup shapes and buffer counts are chosen for residency (three workgroups per CU where LDS allows)
// All forms of the fp32 kernel (3x3 / 1x1 forward and data gradient, ConvTranspose2d scatter / gather).
typedef h16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef h16_t bf16x4 __attribute__((ext_vector_type(4)));

// SPLIT = 1 is precision mode "bf16x3": every operand is carried as hi = bf16(x) and lo = bf16(x - hi) (16 mantissa
// bits) and each k16-step issues three MFMAs, hi*hi + hi*lo + lo*hi, into the same fp32 accumulator -- ~4e-5 on the
// logits instead of bf16's 2e-2, i.e. inside the reference's 1e-3 contract, at 3 bf16 MFMAs per product instead of
// one 16x slower fp32 MFMA.  SPLIT = 2 is "bf16x6": three planes hi, mid, lo = the fp32 operand exactly, six MFMAs
// (all products down to 2^-16 relative; lo*mid, mid*lo, lo*lo < 2^-24 are dropped): fp32-class accuracy.
// Stage = one tap (all planes) instead of one kernel row.
// NTW = 32-channel tiles per wave: 2 (wave tile 64 px x 64 ch) or 1 (narrow: 64 px x 32 ch, workgroup 128 px x 64 ch,
// 40.8 KB of LDS with one stage buffer and ~110 VGPRs: FOUR workgroups per CU)
template <int KS, int WM, int WN, int AMODE, int EPI, int SPLIT, int NTW>
__global__ __launch_bounds__(256, (SPLIT == 2) ? 2 : (NTW == 1) ? 4 : ((WM == 4 && !SPLIT) || (WM == 2 && SPLIT)) ? 3 : 2) void conv_fwd_bf16_kernel(ConvFwdArgs a) {
  constexpr int T = KS * KS, PAD = KS / 2;
  constexpr int TPIX = 64 * WM;
  constexpr int MAXHP = (KS == 1) ? TPIX : (WM == 2 ? 6 * 34 : 10 * 34);
  constexpr int BN = 32 * NTW * WN;
  constexpr int CS = 40;                          // halves per staged pixel / per staged weight row (32 + 8 pad)
  constexpr int NLD_A = (MAXHP * 8 + 255) / 256;  // float4 global loads per thread per A chunk
  constexpr int TS = SPLIT ? 1 : KS;              // taps per stage
  constexpr int NPL = SPLIT + 1;                  // operand planes: hi | hi, lo | hi, mid, lo
  // products issued per k16-step, smallest first (plane of A, plane of B); dropped terms are <= 2^-16 (two planes)
  // resp. 2^-24 (three planes) relative
  constexpr int NTERM = (SPLIT == 0) ? 1 : (SPLIT == 1) ? 3 : 6;
  constexpr int TA[6] = {2, 0, 1, 1, 0, 0};       // the LAST NTERM entries are used: (hi,hi) | (lo,hi) (hi,lo) (hi,hi) |
  constexpr int TB[6] = {0, 2, 1, 0, 1, 0};       // (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
  constexpr int SR = TS * NPL;                    // [BN][32] row blocks per B stage
  constexpr int NST = SPLIT ? T : KS;             // stages per 32-channel chunk
  constexpr int BS = 32;                          // halves per staged weight row: 64 bytes, unpadded, XOR-swizzled
  constexpr int NLD_B = (SR * BN) / 64;           // LDS-DMA instructions per wave per B stage (1 KB = 16 rows each)
  // Stage buffers.  The 2x2 split kernel keeps ONE: 48.6 KB of LDS then lets three workgroups share a CU, and a third
  // workgroup hides more than prefetching the next stage inside the workgroup did (measured +5..16 %; a three-buffer
  // ring with two stages in flight measured 0 %).
  constexpr int NBUF = ((SPLIT && WM == 2) || NTW == 1) ? 1 : 2;
  __shared__ __attribute__((aligned(16))) h16_t smem_h[NPL * MAXHP * CS + NBUF * SR * BN * BS];
  h16_t* a_lds = smem_h;
  h16_t* b_lds = smem_h + NPL * MAXHP * CS;
  float* smem = reinterpret_cast<float*>(smem_h);  // the statistics epilogue reuses the staging area as floats

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  int bx, nb;
  if (!conv_block_ids(a, bx, nb)) return;

  const int img = bx / a.tiles_img;
  const int tin = bx - img * a.tiles_img;
  int seg = 0;
#pragma unroll
  for (int k = 1; k < HPRI_MAXSEG; ++k)
    if (k < a.nseg && tin >= a.seg_first[k]) seg = k;
  const int twl = a.seg_twl[seg];
  const int TW = 1 << twl, RW = 32 >> twl;
  const int TH = 2 * WM * RW;
  const int HW = TW + KS - 1, HP = (TH + KS - 1) * HW;
  const int tt = tin - a.seg_first[seg];
  const int ty = tt / a.seg_ntx[seg], tx = tt - ty * a.seg_ntx[seg];
  const int y0 = ty * TH, x0 = a.seg_xbeg[seg] + tx * TW;
  const int xlim = min(a.W, a.seg_xbeg[seg] + a.seg_ntx[seg] * TW);

  f32x16 acc[2][NTW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks_all = (a.Cin_pad + 31) >> 5;
  const int cps = (nchunks_all + a.ksplit - 1) / a.ksplit;
  const int chunk0 = blockIdx.z * cps;
  const int nchunks = min(nchunks_all, chunk0 + cps);
  const int S0 = chunk0 * NST, S = nchunks * NST;        // stage = (chunk, kernel row) or (chunk, tap) when SPLIT

  // ---- B stages: SR row blocks (taps x planes) of BN rows x 64 bytes, contiguous per row block in the packed tensor.
  // LDS-DMA (global_load_lds_dwordx4): no VGPR staging, no ds_write.  A wave instruction fills 1 KB = 16 rows; lane l
  // lands on 16-byte slot l&3 of row l>>2.  Rows are unpadded (64-byte pitch), so the four 16-byte segments of a row are
  // XOR-swizzled with (row>>2)&3 to keep the ds_read_b128 of 32 consecutive rows conflict-free (ds_read_b128 is served
  // in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: (row>>1)&3 left them 2-way conflicted); the DMA cannot permute,
  // but every lane chooses WHICH global segment it fetches, which is the same thing.
  const h16_t* wpk = reinterpret_cast<const h16_t*>(a.wp);
  int goff[NLD_B];
#pragma unroll
  for (int p = 0; p < NLD_B; ++p) {
    const int R = (p * 4 + wave) * 16 + (lane >> 2);          // row within the stage
    const int rb = R / BN, n = R % BN;
    const int ls = (lane & 3) ^ ((n >> 2) & 3);               // logical segment stored at physical slot lane&3
    goff[p] = (rb * a.Cout_pad + nb * BN + n) * 32 + ls * 8;  // halves, relative to the stage's first row block
  }
#define LOAD_STAGE(s_)                                                                               \
  {                                                                                                  \
    const h16_t* pb_ = wpk + (size_t)(s_) * SR * a.Cout_pad * 32;                                   \
    h16_t* lb_ = b_lds + ((s_) % NBUF) * SR * BN * BS;                                              \
    _Pragma("unroll") for (int p = 0; p < NLD_B; ++p)                                                \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb_ + goff[p]),           \
                                         (__attribute__((address_space(3))) void*)(lb_ + (p * 4 + wave) * 512), 16, 0, 0); \
  }
#define STORE_STAGE(buf_)

  // ---- A halo: fp32 in HBM -> registers -> bf16 in LDS ----
  const float* ximg = a.x + (size_t)img * (AMODE == HPRI_A_DIRECT ? (size_t)a.H * a.W : (size_t)a.H2 * a.W2) * a.x_cs;
  int aoff[NLD_A];
  {
    const unsigned hw_inv = (65536u + (unsigned)HW - 1u) / (unsigned)HW;
#pragma unroll
    for (int p = 0; p < NLD_A; ++p) {
      const int f = tid + p * 256;
      const int pix = f >> 3, q = f & 7;
      int off = -1;
      if (pix < HP) {
        const int hy = (int)(((unsigned)pix * hw_inv) >> 16), hx = pix - hy * HW;
        const int iy = y0 + hy - PAD, ix = x0 + hx - PAD;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
          if (AMODE == HPRI_A_DIRECT) off = (iy * a.W + ix) * a.x_cs + a.x_coff + q * 4;
          else off = ((2 * iy + a.py0) * a.W2 + 2 * ix + a.px0) * a.x_cs + a.x_coff;   // + tap / channel per chunk
        }
      }
      aoff[p] = off;
    }
  }
  f32x4 areg[NLD_A];
#define LOAD_A(c0_)                                                                                  \
  {                                                                                                  \
    const int kq = min(8, (a.Cin_pad - (c0_)) >> 2);                                                 \
    _Pragma("unroll") for (int p = 0; p < NLD_A; ++p) {                                              \
      const int q = (tid + p * 256) & 7;                                                             \
      f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                                \
      if (aoff[p] >= 0 && q < kq) {                                                                  \
        if (AMODE == HPRI_A_DIRECT) {                                                                \
          v = *reinterpret_cast<const f32x4*>(ximg + (size_t)(unsigned)aoff[p] + (c0_));            \
        } else { /* S2D: k = tap*Cup + co ; source pixel (2*iy + t_y + py0, 2*ix + t_x + px0) */     \
          const int k4 = (c0_) + q * 4;                                                              \
          const int tp = k4 / a.Cup, co = k4 - tp * a.Cup;                                           \
          v = *reinterpret_cast<const f32x4*>(ximg + (size_t)(unsigned)aoff[p] +                     \
                                              ((tp >> 1) * a.W2 + (tp & 1)) * a.x_cs + co);          \
        }                                                                                            \
      }                                                                                              \
      areg[p] = v;                                                                                   \
    }                                                                                                \
  }
#define STORE_A()                                                                                    \
  _Pragma("unroll") for (int p = 0; p < NLD_A; ++p) {                                                \
    const int f = tid + p * 256;                                                                     \
    if ((f >> 3) < HP) {                                                                             \
      f32x4 rest_ = areg[p];                                                                         \
      _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                                           \
        const bf16x4 h_ = __builtin_convertvector(rest_, bf16x4);                                    \
        *reinterpret_cast<bf16x4*>(a_lds + pl * MAXHP * CS + (f >> 3) * CS + (f & 7) * 4) = h_;      \
        if (pl + 1 < NPL) rest_ = rest_ - __builtin_convertvector(h_, f32x4);                        \
      }                                                                                              \
    }                                                                                                \
  }

  const int a_base = ((wm * 2 * RW + (li >> twl)) * HW + (li & (TW - 1))) * CS + lh * 8;
  const int a_mt = RW * HW * CS;
  const int b_base = (wn * (32 * NTW) + li) * BS;
  const int bsw0 = ((0 + lh) ^ ((li >> 2) & 3)) * 8, bsw1 = ((2 + lh) ^ ((li >> 2) & 3)) * 8;   // swizzled k16-step offsets

  if (NBUF > 1) { LOAD_STAGE(S0) }
  LOAD_A(chunk0 * 32)
  for (int s = S0; s < S; ++s) {
    const int chunk = s / NST, st = s - chunk * NST;         // st = kernel row, or tap when SPLIT
    if (st == 0) {
      __syncthreads();
      STORE_A()
    }
    if (NBUF == 1) {
      __syncthreads();                                       // everyone is done with the previous stage
      if (st == NST - 1 && chunk + 1 < nchunks) { LOAD_A((chunk + 1) * 32) }
      LOAD_STAGE(s)
      __syncthreads();
    } else {
      __syncthreads();
      if (s + 1 < S) { LOAD_STAGE(s + 1) }
      if (st == NST - 1 && chunk + 1 < nchunks) { LOAD_A((chunk + 1) * 32) }
    }

    const h16_t* bp = b_lds + (s % NBUF) * SR * BN * BS + b_base;
    if (!SPLIT) {
      const h16_t* ap = a_lds + a_base + st * HW * CS;
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bf16x8 af[2], bf[NTW];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) af[mt] = *reinterpret_cast<const bf16x8*>(ap + mt * a_mt + dx * CS + kk * 16);
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) bf[nt] = *reinterpret_cast<const bf16x8*>(bp + (dx * BN + nt * 32) * BS + (kk ? bsw1 : bsw0));
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
              acc[mt][nt] = HPRI_MFMA_32X32X16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    } else {
      const int dy = st / KS, dx = st - dy * KS;
      const h16_t* ap = a_lds + a_base + (dy * HW + dx) * CS;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 afr[NPL][2], bfr[NPL][NTW];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            afr[pl][mt] = *reinterpret_cast<const bf16x8*>(ap + pl * MAXHP * CS + mt * a_mt + kk * 16);
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
            bfr[pl][nt] = *reinterpret_cast<const bf16x8*>(bp + (pl * BN + nt * 32) * BS + (kk ? bsw1 : bsw0));
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int t = 6 - NTERM; t < 6; ++t)
              acc[mt][nt] = HPRI_MFMA_32X32X16(afr[TA[t]][mt], bfr[TB[t]][nt], acc[mt][nt], 0, 0, 0);
      }
    }
  }
#undef LOAD_STAGE
#undef STORE_STAGE
#undef LOAD_A
#undef STORE_A

#define HPRI_EPI_NTW NTW
#include "conv_fwd_epilogue.inc"
#undef HPRI_EPI_NTW
}

// Split-K epilogue: y = sum_z ws[z] + bias (fixed order), NHWC store (optionally accumulating), and per-block BN
// partial statistics (mean, M2, count) over SK_PIX consecutive pixels of one image.
// grid = (pixel blocks per image, ceil(Cw4 / CQ), N); block = 256 = ROWS x CQ channel quads.
#define SK_PIX 64
__global__ void splitk_finish_kernel(const float* __restrict__ ws, int ksplit, int Cout_pad, const float* __restrict__ bias,
                                     float* __restrict__ y, int y_cs, int y_coff, float4* __restrict__ stats, int HW,
                                     long long P, int Cout, int y_cw, int CQ, int accumulate, int relu) {
  __shared__ float4 red[256];
  const int rows = 256 / CQ;
  const int cq = threadIdx.x % CQ, pr = threadIdx.x / CQ;
  const int c = (blockIdx.y * CQ + cq) * 4;
  const int img = blockIdx.z;
  const int q0 = blockIdx.x * SK_PIX, q1 = min(HW, q0 + SK_PIX);
  const float cnt = (float)(q1 - q0);
  const bool live = c < y_cw;
  float b[4] = {0.f, 0.f, 0.f, 0.f};
  if (live && bias != nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) if (c + j < Cout) b[j] = bias[c + j];
  }
  float sum[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    for (int q = q0 + pr; q < q1; q += rows) {
      const size_t p = (size_t)img * HW + q;
      float v[4] = {b[0], b[1], b[2], b[3]};
      for (int z = 0; z < ksplit; ++z) {
        const float4 t = *reinterpret_cast<const float4*>(ws + ((size_t)z * P + p) * Cout_pad + c);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { if (relu) v[j] = fmaxf(v[j], 0.f); if (c + j >= Cout) v[j] = 0.f; }
      float* o = y + p * y_cs + y_coff + c;
      if (accumulate) { const float4 old = *reinterpret_cast<const float4*>(o); v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w; }
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) sum[j] += v[j];
    }
  }
  if (stats == nullptr) return;
  red[threadIdx.x] = make_float4(sum[0], sum[1], sum[2], sum[3]);
  __syncthreads();
  float mean[4];
  {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < rows; ++r) { const float4 u = red[r * CQ + cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    mean[0] = t.x / cnt; mean[1] = t.y / cnt; mean[2] = t.z / cnt; mean[3] = t.w / cnt;
  }
  __syncthreads();
  float m2[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    for (int q = q0 + pr; q < q1; q += rows) {   // second pass re-reads what this thread just wrote
      const float4 t = *reinterpret_cast<const float4*>(y + ((size_t)img * HW + q) * y_cs + y_coff + c);
      const float d0 = t.x - mean[0], d1 = t.y - mean[1], d2 = t.z - mean[2], d3 = t.w - mean[3];
      m2[0] += d0 * d0; m2[1] += d1 * d1; m2[2] += d2 * d2; m2[3] += d3 * d3;
    }
  }
  red[threadIdx.x] = make_float4(m2[0], m2[1], m2[2], m2[3]);
  __syncthreads();
  if (pr == 0 && c < Cout_pad) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < rows; ++r) { const float4 u = red[r * CQ + cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    float4* o = stats + ((size_t)img * gridDim.x + blockIdx.x) * Cout_pad + c;
    o[0] = make_float4(mean[0], t.x, cnt, 0.f);
    o[1] = make_float4(mean[1], t.y, cnt, 0.f);
    o[2] = make_float4(mean[2], t.z, cnt, 0.f);
    o[3] = make_float4(mean[3], t.w, cnt, 0.f);
  }
}

// Launch of the split-K epilogue (shared by the fp32, bf16 and bf16-plane convolution launchers).
extern "C" int hpri_splitk_finish(const float* ws, int ksplit, int Cout_pad, const float* bias, float* y, int y_cs, int y_coff,
                                  float* stats, int N, int HW, int Cout, int y_cw, int accumulate, int relu, hipStream_t stream) {
  const int c4 = y_cw >> 2;
  int cq = 1; while (cq < c4 && cq < 64) cq <<= 1;
  dim3 grid((unsigned)hpri_cdiv(HW, SK_PIX), (unsigned)hpri_cdiv(hpri_cdiv(y_cw, 4), cq), (unsigned)N);
  hipLaunchKernelGGL(splitk_finish_kernel, grid, dim3(256), 0, stream, ws, ksplit, Cout_pad, bias, y, y_cs, y_coff,
                     reinterpret_cast<float4*>(stats), HW, (long long)N * HW, Cout, y_cw, cq, accumulate, relu);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// Tile-shape choice shared by the launcher and the sizing query.
// Workgroup shape: 2x2 waves (128 px x 128 ch) when the output channels allow it, else 4x1 (256 px x 64 ch).
// prec: 0 fp32, 1 bf16, 2 bf16x3, 3 bf16x6.  The plain bf16 kernel always takes 4x1: its 51 KB of LDS and 168 VGPRs let THREE
// workgroups share a CU, which beats the wider tile on every layer (measured +4..17 %); the waves of a workgroup leave
// each barrier in phase, so only other workgroups can fill the matrix pipe while one is reading its fragments.
static inline void conv_cfg(int Cout_pad, int* wm, int* wn, int prec = 0) {
  if (Cout_pad % 128 == 0 && prec != 1) { *wm = 2; *wn = 2; }
  else if (prec == 3) { *wm = 2; *wn = 1; }      // bf16x6, 64-channel remainder: three 256-pixel planes do not fit in LDS;
                                                 // the narrow 128 px x 64 ch workgroup (2x2 waves of 64 px x 32 ch) does
  else { *wm = 4; *wn = 1; }
}

// Cut the image width into column bands of tile width 32 / 16 / 8 / 4 (host only).  Candidates: plain 32-wide
// rounding up, or full 32-wide tiles followed by a greedy cover of the remainder; the cheaper one in tile
// pixel-slots (rows round up to the band's tile height) wins.  Returns tiles per image.
struct ConvSegs { int nseg, tiles_img, twl[HPRI_MAXSEG], xbeg[HPRI_MAXSEG], ntx[HPRI_MAXSEG], first[HPRI_MAXSEG]; };
static ConvSegs conv_segments(int H, int W, int wm) {
  const int min_tw = (wm == 2) ? 4 : 8;          // LDS budget of the 256-pixel configuration stops at width 8
  auto th = [&](int tw) { return 2 * wm * (32 / tw); };
  auto slots = [&](int tw, int ntx) { return (long long)hpri_cdiv(H, th(tw)) * th(tw) * tw * ntx; };
  ConvSegs plain{}; plain.nseg = 1; plain.twl[0] = 5; plain.xbeg[0] = 0; plain.ntx[0] = hpri_cdiv(W, 32);
  long long cost_plain = slots(32, plain.ntx[0]);
  ConvSegs g{}; long long cost_g = 0; int x = 0;
  if (W / 32 > 0) { g.twl[g.nseg] = 5; g.xbeg[g.nseg] = 0; g.ntx[g.nseg] = W / 32; cost_g += slots(32, W / 32); x = (W / 32) * 32; g.nseg++; }
  int rem = W - x;
  for (int tw = 16; tw >= min_tw && rem > 0; tw >>= 1) {
    int n = rem / tw;
    if (tw == min_tw && rem % tw) n += 1;        // last band rounds up
    if (n > 0 && g.nseg < HPRI_MAXSEG) {
      int l = 0; while ((1 << l) < tw) ++l;
      g.twl[g.nseg] = l; g.xbeg[g.nseg] = x; g.ntx[g.nseg] = n; cost_g += slots(tw, n);
      x += n * tw; rem = W - x; g.nseg++;
    }
  }
  ConvSegs r = (g.nseg > 0 && rem <= 0 && cost_g < cost_plain) ? g : plain;
  int first = 0;
  for (int k = 0; k < r.nseg; ++k) { r.first[k] = first; first += hpri_cdiv(H, th(1 << r.twl[k])) * r.ntx[k]; }
  r.tiles_img = first;
  return r;
}

// Channel-block count from which the XCD-aware 1-D grid is used (conv_block_ids).  Default 9: only shapes with more
// channel blocks than any CubeNET / UNet layer has (SpectralUNET's 13 and 26); option "conv_nbx_min" (api.cpp).
static int conv_nbx_min() { return hpri_option(0); }

template <int KS, int WM, int WN, int AMODE, int EPI>
static int launch_conv(const ConvFwdArgs& a0, hipStream_t stream) {
  ConvFwdArgs a = a0;
  constexpr int BN = 64 * WN;
  const ConvSegs sg = conv_segments(a.H, a.W, WM);
  a.nseg = sg.nseg; a.tiles_img = sg.tiles_img;
  for (int k = 0; k < HPRI_MAXSEG; ++k) { a.seg_twl[k] = sg.twl[k]; a.seg_xbeg[k] = sg.xbeg[k]; a.seg_ntx[k] = sg.ntx[k]; a.seg_first[k] = sg.first[k]; }
  const int NB = a.Cout_pad / BN, tiles = a.N * a.tiles_img;
  a.nbx = (NB >= conv_nbx_min()) ? NB : 0;
  dim3 grid((unsigned)tiles, (unsigned)NB, (unsigned)a.ksplit);
  if (a.nbx > 0) grid = dim3((unsigned)(hpri_cdiv(tiles, 8) * 8 * NB), 1u, (unsigned)a.ksplit);
  hipLaunchKernelGGL((conv_fwd_kernel<KS, WM, WN, AMODE, EPI>), grid, dim3(256), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// Split-K plan (host only).  A layer whose natural grid leaves CUs idle or badly balanced (38x60 and 76x121 levels) is cut
// along K so that the number of equal-sized workgroups per CU is close to an integer -- when that is worth its HBM traffic:
// the kernel writes k fp32 slabs instead of one output and splitk_finish reads them back (2 k output sizes more than k = 1).
// Priced with the rate the kernel family reaches (fp32 120, bf16 600, bf16x3 200, bf16x6 100 TFLOP/s) and 5 TB/s.
static inline int conv_ksplit(int N, int H, int W, int Cin_pad, int Cout_pad, int KS, int epi, int amode, int prec = 0) {
  if (epi != HPRI_E_DIRECT) return 1;
  int wm, wn; conv_cfg(Cout_pad, &wm, &wn, prec);
  const long long blocks = (long long)N * conv_segments(H, W, wm).tiles_img * (Cout_pad / (64 * wn));
  const int nchunks = hpri_cdiv(Cin_pad, 32);
  if (blocks >= 2048) return 1;                       // >= 8 workgroups per CU: balance is already fine
  const double rate = prec == 0 ? 120e12 : prec == 1 ? 600e12 : prec == 2 ? 200e12 : 100e12;
  const double t_compute = 2.0 * N * H * W * (double)Cin_pad * Cout_pad * KS * KS / rate;
  const double out_bytes = 4.0 * N * H * W * (double)Cout_pad;
  int best = 1; double best_t = 1e30;
  for (int k = 1; k <= 4; ++k) {
    if (k > 1 && nchunks / k < 4) break;              // keep >= 4 chunks (128 channels x taps) per split
    const double per_cu = (double)blocks * k / 256.0;
    const double eff = per_cu / (double)((long long)(per_cu + 0.999999));
    const double t = t_compute / eff + (k > 1 ? 2.0 * k * out_bytes / 5e12 + 4e-6 : 0.0);
    if (t < best_t - 1e-12) { best_t = t; best = k; }
  }
  return best;
}

static int conv_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int KS, int amode, int epi, int prec,
                     int* ksplit, int* stat_tiles, size_t* ws_floats) {
  const int k = conv_ksplit(N, H, W, Cin_pad, Cout_pad, KS, epi, amode, prec);
  *ksplit = k;
  if (k > 1) {
    *stat_tiles = N * hpri_cdiv(H * W, SK_PIX);
    *ws_floats = (size_t)k * N * H * W * Cout_pad;
  } else {
    int wm, wn; conv_cfg(Cout_pad, &wm, &wn, prec);
    *stat_tiles = N * conv_segments(H, W, wm).tiles_img;
    *ws_floats = 0;
  }
  return HPRI_OK;
}

extern "C" int hpri_conv_fwd_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int KS, int amode, int epi,
                                  int* ksplit, int* stat_tiles, size_t* ws_floats) {
  return conv_plan(N, H, W, Cin_pad, Cout_pad, KS, amode, epi, 0, ksplit, stat_tiles, ws_floats);
}

// the same query for hpri_conv_fwd_bf16 (split = 0: bf16, 1: bf16x3, 2: bf16x6), whose workgroup shapes differ
extern "C" int hpri_conv_fwd_bf16_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int KS, int amode, int epi,
                                       int split, int* ksplit, int* stat_tiles, size_t* ws_floats) {
  return conv_plan(N, H, W, Cin_pad, Cout_pad, KS, amode, epi, (split < 0 || split > 2) ? 1 : split + 1, ksplit, stat_tiles, ws_floats);
}

template <int KS, int WM, int WN, int AMODE, int EPI, int SPLIT, int NTW = 2>
static int launch_conv_bf16(const ConvFwdArgs& a0, hipStream_t stream) {
  ConvFwdArgs a = a0;
  constexpr int BN = 32 * NTW * WN;
  const ConvSegs sg = conv_segments(a.H, a.W, WM);
  a.nseg = sg.nseg; a.tiles_img = sg.tiles_img;
  for (int k = 0; k < HPRI_MAXSEG; ++k) { a.seg_twl[k] = sg.twl[k]; a.seg_xbeg[k] = sg.xbeg[k]; a.seg_ntx[k] = sg.ntx[k]; a.seg_first[k] = sg.first[k]; }
  const int NB = a.Cout_pad / BN, tiles = a.N * a.tiles_img;
  a.nbx = (NB >= conv_nbx_min()) ? NB : 0;
  dim3 grid((unsigned)tiles, (unsigned)NB, (unsigned)a.ksplit);
  if (a.nbx > 0) grid = dim3((unsigned)(hpri_cdiv(tiles, 8) * 8 * NB), 1u, (unsigned)a.ksplit);
  hipLaunchKernelGGL((conv_fwd_bf16_kernel<KS, WM, WN, AMODE, EPI, SPLIT, NTW>), grid, dim3(256), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// bf16-operand variant of hpri_conv_fwd (same modes); wp from hpri_pack_weight_bf16.
static int conv_fwd_bf16_impl(const float* x, int x_cs, int x_coff, const void* wp, const float* bias,
                              float* y, int y_cs, int y_coff, float* stats,
                              int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw,
                              int KS, int amode, int epi, int accumulate, int H2, int W2, int py0, int px0, int Cup,
                              int split, float* ws, size_t ws_floats, void* planes, int pl_cs, int pl_coff, hipStream_t stream) {
  HPRI_REQUIRE(x && wp && (y || planes), "conv_fwd_bf16: null pointer");
  if (planes != nullptr)
    HPRI_REQUIRE(epi == HPRI_E_D2S && split == 0 && !(accumulate & 1) && pl_cs % 8 == 0 && pl_coff % 8 == 0 && pl_coff + Cup <= pl_cs &&
                     ((uintptr_t)planes & 15) == 0, "conv_fwd_bf16: plane output is for the plain-bf16 transposed convolution (D2S), not accumulating");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0, "conv_fwd_bf16: empty image");
  HPRI_REQUIRE(Cin_pad > 0 && Cin_pad % 8 == 0, "conv_fwd_bf16: Cin_pad must be a positive multiple of 8");
  HPRI_REQUIRE(Cout_pad % 64 == 0 && Cout <= Cout_pad && Cout > 0, "conv_fwd_bf16: Cout_pad must be a multiple of 64 >= Cout");
  HPRI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0, "conv_fwd_bf16: input channel stride/offset must be multiples of 4");
  HPRI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)wp & 15) == 0, "conv_fwd_bf16: pointers must be 16-byte aligned");
  HPRI_REQUIRE(KS == 1 || KS == 3, "conv_fwd_bf16: kernel size must be 1 or 3");
  HPRI_REQUIRE((long long)(H2 > H ? H2 : H) * (W2 > W ? W2 : W) * x_cs < (1ll << 31), "conv_fwd_bf16: one image of the input view exceeds 2^31 elements");
  ConvFwdArgs a;
  a.x = x; a.x_cs = x_cs; a.x_coff = x_coff; a.wp = reinterpret_cast<const float*>(wp); a.bias = bias;
  a.y = y; a.y_cs = y_cs; a.y_coff = y_coff; a.stats = reinterpret_cast<float4*>(stats);
  a.N = N; a.H = H; a.W = W; a.Cin_pad = Cin_pad; a.Cout = Cout; a.Cout_pad = Cout_pad;
  a.y_cw = y_cw < Cout ? Cout : y_cw; a.accumulate = accumulate & 1; a.relu = (accumulate >> 1) & 1;
  a.H2 = H2; a.W2 = W2; a.py0 = py0; a.px0 = px0; a.Cup = Cup;
  a.plp = reinterpret_cast<h16_t*>(planes); a.pl_cs = pl_cs; a.pl_coff = pl_coff;
  if (epi == HPRI_E_DIRECT) HPRI_REQUIRE(a.y_cw + y_coff <= y_cs, "conv_fwd_bf16: output channels exceed the channel stride");
  if (amode == HPRI_A_S2D || epi == HPRI_E_D2S) {
    HPRI_REQUIRE(KS == 1, "conv_fwd_bf16: S2D/D2S need KS == 1");
    HPRI_REQUIRE(Cup > 0 && Cup % 4 == 0, "conv_fwd_bf16: Cup must be a positive multiple of 4");
    HPRI_REQUIRE(py0 >= 0 && px0 >= 0 && 2 * H + py0 <= H2 && 2 * W + px0 <= W2, "conv_fwd_bf16: 2x2 patch grid exceeds the hi-res image");
    HPRI_REQUIRE(stats == nullptr, "conv_fwd_bf16: statistics epilogue is only available for direct stores");
    if (amode == HPRI_A_S2D) HPRI_REQUIRE(Cin_pad == 4 * Cup, "conv_fwd_bf16: S2D needs Cin_pad == 4*Cup");
    if (epi == HPRI_E_D2S) HPRI_REQUIRE(Cout == 4 * Cup, "conv_fwd_bf16: D2S needs Cout == 4*Cup");
    HPRI_REQUIRE(!(amode == HPRI_A_S2D && epi == HPRI_E_D2S), "conv_fwd_bf16: S2D and D2S are exclusive");
  }
  HPRI_REQUIRE(split >= 0 && split <= 2, "conv_fwd_bf16: split must be 0 (bf16), 1 (bf16x3) or 2 (bf16x6)");
  a.ksplit = conv_ksplit(N, H, W, Cin_pad, Cout_pad, KS, epi, amode, split + 1);
  a.ws = ws;
  if (a.ksplit > 1) {
    if (ws == nullptr || (size_t)a.ksplit * N * H * W * Cout_pad > ws_floats)
      return hpri_set_error(HPRI_ERR_WORKSPACE, "conv_fwd_bf16: split-K workspace too small (see hpri_conv_fwd_plan)");
    a.stats = nullptr;
    a.accumulate = 0;
  }
  int wm, wn; conv_cfg(Cout_pad, &wm, &wn, split + 1);
  int rc;
#define HPRI_DISPATCH_B(KS_, AM_, EP_)                                                      \
  rc = (split == 2) ? ((wn == 2) ? launch_conv_bf16<KS_, 2, 2, AM_, EP_, 2>(a, stream) : launch_conv_bf16<KS_, 2, 2, AM_, EP_, 2, 1>(a, stream)) \
     : (split == 1) ? ((wm == 2) ? launch_conv_bf16<KS_, 2, 2, AM_, EP_, 1>(a, stream) : launch_conv_bf16<KS_, 4, 1, AM_, EP_, 1>(a, stream)) \
                    : ((wm == 2) ? launch_conv_bf16<KS_, 2, 2, AM_, EP_, 0>(a, stream) : launch_conv_bf16<KS_, 4, 1, AM_, EP_, 0>(a, stream))
  if (KS == 3) { HPRI_DISPATCH_B(3, HPRI_A_DIRECT, HPRI_E_DIRECT); }
  else if (amode == HPRI_A_S2D) { HPRI_DISPATCH_B(1, HPRI_A_S2D, HPRI_E_DIRECT); }
  else if (epi == HPRI_E_D2S) { HPRI_DISPATCH_B(1, HPRI_A_DIRECT, HPRI_E_D2S); }
  else { HPRI_DISPATCH_B(1, HPRI_A_DIRECT, HPRI_E_DIRECT); }
#undef HPRI_DISPATCH_B
  if (rc != HPRI_OK || a.ksplit == 1) return rc;
  return hpri_splitk_finish(ws, a.ksplit, Cout_pad, bias, y, y_cs, y_coff, stats, N, H * W, Cout, a.y_cw, accumulate & 1, a.relu, stream);
}

extern "C" int hpri_conv_fwd_bf16(const float* x, int x_cs, int x_coff, const void* wp, const float* bias,
                                  float* y, int y_cs, int y_coff, float* stats,
                                  int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw,
                                  int KS, int amode, int epi, int accumulate, int H2, int W2, int py0, int px0, int Cup,
                                  int split, float* ws, size_t ws_floats, hipStream_t stream) {
  return conv_fwd_bf16_impl(x, x_cs, x_coff, wp, bias, y, y_cs, y_coff, stats, N, H, W, Cin_pad, Cout, Cout_pad, y_cw, KS, amode, epi,
                            accumulate, H2, W2, py0, px0, Cup, split, ws, ws_floats, nullptr, 0, 0, stream);
}

// ConvTranspose2d(k2, s2) forward in the plain bf16 mode with the result written as ONE bf16 plane (channels [pl_coff, pl_coff +
// Cup) of a plane buffer with pl_cs elements per hi-res pixel: the decoder's concat planes) -- also as fp32 when y != NULL.
extern "C" int hpri_convt_fwd_bf16_pl(const float* x, int x_cs, int x_coff, const void* wp, const float* bias,
                                      float* y, int y_cs, int y_coff, int N, int H, int W, int Cin_pad, int Cout, int Cout_pad,
                                      int H2, int W2, int py0, int px0, int Cup, void* planes, int pl_cs, int pl_coff,
                                      hipStream_t stream) {
  HPRI_REQUIRE(planes != nullptr, "convt_fwd_bf16_pl: null plane pointer");
  return conv_fwd_bf16_impl(x, x_cs, x_coff, wp, bias, y, y_cs, y_coff, nullptr, N, H, W, Cin_pad, Cout, Cout_pad, Cout, 1,
                            HPRI_A_DIRECT, HPRI_E_D2S, 0, H2, W2, py0, px0, Cup, 0, nullptr, 0, planes, pl_cs, pl_coff, stream);
}

extern "C" int hpri_conv_fwd(const float* x, int x_cs, int x_coff, const float* wp, const float* bias,
                             float* y, int y_cs, int y_coff, float* stats,
                             int N, int H, int W, int Cin_pad, int Cout, int Cout_pad, int y_cw,
                             int KS, int amode, int epi, int accumulate,
                             int H2, int W2, int py0, int px0, int Cup, float* ws, size_t ws_floats,
                             hipStream_t stream) {
  HPRI_REQUIRE(x && wp && y, "conv_fwd: null pointer");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0, "conv_fwd: empty image");
  HPRI_REQUIRE(Cin_pad > 0 && Cin_pad % 8 == 0, "conv_fwd: Cin_pad must be a positive multiple of 8");
  HPRI_REQUIRE(Cout_pad % 64 == 0 && Cout <= Cout_pad && Cout > 0, "conv_fwd: Cout_pad must be a multiple of 64 >= Cout");
  HPRI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0, "conv_fwd: input channel stride/offset must be multiples of 4");
  HPRI_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)wp & 15) == 0, "conv_fwd: pointers must be 16-byte aligned");
  HPRI_REQUIRE(KS == 1 || KS == 3, "conv_fwd: kernel size must be 1 or 3");
  HPRI_REQUIRE((long long)(H2 > H ? H2 : H) * (W2 > W ? W2 : W) * x_cs < (1ll << 31), "conv_fwd: one image of the input view exceeds 2^31 elements");
  ConvFwdArgs a;
  a.x = x; a.x_cs = x_cs; a.x_coff = x_coff; a.wp = wp; a.bias = bias;
  a.y = y; a.y_cs = y_cs; a.y_coff = y_coff; a.stats = reinterpret_cast<float4*>(stats);
  a.N = N; a.H = H; a.W = W; a.Cin_pad = Cin_pad; a.Cout = Cout; a.Cout_pad = Cout_pad;
  a.y_cw = y_cw < Cout ? Cout : y_cw; a.accumulate = accumulate & 1; a.relu = (accumulate >> 1) & 1;
  a.H2 = H2; a.W2 = W2; a.py0 = py0; a.px0 = px0; a.Cup = Cup;
  a.plp = nullptr; a.pl_cs = 0; a.pl_coff = 0;
  a.ksplit = conv_ksplit(N, H, W, Cin_pad, Cout_pad, KS, epi, amode);
  a.ws = ws;
  if (a.ksplit > 1) {
    if (ws == nullptr || (size_t)a.ksplit * N * H * W * Cout_pad > ws_floats)
      return hpri_set_error(HPRI_ERR_WORKSPACE, "conv_fwd: split-K workspace too small (see hpri_conv_fwd_plan)");
    a.stats = nullptr;       // statistics come from the finish kernel
    a.accumulate = 0;
  }
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
  rc = (wm == 2) ? launch_conv<KS_, 2, 2, AM_, EP_>(a, stream) : launch_conv<KS_, 4, 1, AM_, EP_>(a, stream)
  int rc;
  if (KS == 3) { HPRI_DISPATCH(3, HPRI_A_DIRECT, HPRI_E_DIRECT); }
  else if (amode == HPRI_A_S2D) { HPRI_DISPATCH(1, HPRI_A_S2D, HPRI_E_DIRECT); }
  else if (epi == HPRI_E_D2S) { HPRI_DISPATCH(1, HPRI_A_DIRECT, HPRI_E_D2S); }
  else { HPRI_DISPATCH(1, HPRI_A_DIRECT, HPRI_E_DIRECT); }
#undef HPRI_DISPATCH
  if (rc != HPRI_OK || a.ksplit == 1) return rc;
  return hpri_splitk_finish(ws, a.ksplit, Cout_pad, bias, y, y_cs, y_coff, stats, N, H * W, Cout, a.y_cw, accumulate & 1, a.relu, stream);
}
