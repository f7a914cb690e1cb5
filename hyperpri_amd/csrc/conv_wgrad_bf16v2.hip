// Weight gradient of the 3x3 / pad 1 convolutions on bf16 activation PLANES (precision mode "bf16"; the autograd of
// reference model_parts.py:22,25 and models.py:169,177):   dW[n][c][ky][kx] = sum_pixels dY[p][n] * X[p + (ky-1, kx-1)][c]
//
// Round-1's bf16 weight-gradient kernel loaded fp32 X and dY into VGPRs, rounded them and ds_write'd them per staged unit,
// with no load in flight while it multiplied (144 accumulators left no staging registers): 0.15 of the bf16 MFMA peak.
// Here both operands are already bf16 NHWC planes in HBM (written by the producers: BN-apply, pooling, concat, ingest for
// X; the fused ReLU/BN backward for dY), so they reach LDS by LDS-DMA, double-buffered, one unit ahead of the MFMAs.
//
//   GEMM       rows n (64 output channels) x columns c (64 input channels) x 9 taps per workgroup, K = pixels
//   workgroup  512 threads = 8 waves = 2 pixel halves (kg) x 2 (n halves) x 2 (c halves); wave tile 32 n x 32 c x 9 taps
//              = 144 accumulator VGPRs; ONE workgroup per CU, two waves per SIMD
//   unit       4 image rows x 32 pixels (+1-pixel halo for X): X 6x34 pixels x 64 ch and dY 128 pixels x 64 ch as
//              [pixel][64 ch] bf16 rows of 128 B; wave group kg multiplies rows 2kg, 2kg+1 (4 k16-steps of 16 pixels)
//   operands   the MFMA wants 8 consecutive PIXELS per lane: ds_read_b64_tr_b16 transposes 4 pixels x 16 channels per
//              16-lane group on the way out of LDS.  Rows are unpadded (the DMA writes 1 KB per instruction linearly);
//              the two 64-byte halves of a pixel row are swapped when bit 1 of the LDS pixel index is set (the DMA
//              cannot permute its destination, but every lane picks WHICH 16 bytes it fetches), so any four consecutive
//              pixel rows -- every tap shifts the window -- fall on four disjoint 16-bank windows: conflict-free.
//              The swizzled lane addresses are four per-lane constants (window parity x flip); a tap's offset is an
//              immediate of the ds_read.
//   pipeline   unit u+1's 42 DMA pieces are issued right after the barrier that starts unit u; one vmcnt(0) + one
//              barrier per unit of 36 MFMAs per wave
//   grid       1-D, XCD-aware: workgroup id mod 8 is the XCD; an XCD walks the (c, n) tiles of ONE pixel split before the
//              next split, so the re-reads of that split's X / dY rows by the other tiles are served by its L2
//   output     deterministic split-K: the two pixel halves meet in LDS, then one slab ws[split][tap][n][c] per workgroup;
//              hpri_wgrad_reduce_ex sums the slabs in fixed order into OIHW
#include "common.h"

typedef h16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef h16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

struct WgV2Args {
  const h16_t* xp; int x_cs, x_coff, x_cvalid;       // plane 0 of the convolution input; channels >= x_cvalid read as zero
  const h16_t* dyp; int dy_cs, dy_coff, dy_cvalid;   // plane 0 of the gradient w.r.t. the convolution output
  float* ws;                                          // [splits][9][Nr][Cr]
  int N, H, W, Cr, Nr;
  int splits, tiles_c, tiles, units_x, units_y, total_units, units_per_split;
};

#define WG_HW 34                      // halo width in pixels
#define WG_HP (6 * WG_HW)             // 204 halo pixels
#define WG_XI 26                      // X DMA instructions per unit (8 pixel rows each; the last one half used)
#define WG_YI 16                      // dY DMA instructions per unit
#define WG_XB (WG_XI * 1024)
#define WG_YB (WG_YI * 1024)
#define WG_UB (WG_XB + WG_YB)

#ifdef WG_BUILTIN_TR      /* the form of rounds 2-3 (A/B): hipcc puts "s_waitcnt vmcnt(0)" in front of every group of these reads */
__device__ __forceinline__ bf16x8 wg_tr_frag(const unsigned char* p) {
  const bf16x4 lo = __builtin_bit_cast(bf16x4, HPRI_DS_READ_TR16_B64((hpri_lds_tr4_ptr)(p)));
  const bf16x4 hi = __builtin_bit_cast(bf16x4, HPRI_DS_READ_TR16_B64((hpri_lds_tr4_ptr)(p + 512)));     // pixel rows +4
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
#endif
// One fragment = two transposed reads (pixel rows L and L + 4) as inline assembly.  Through the builtin hipcc sees an LDS read it cannot
// tell apart from the LDS-DMA writes in flight and waits vmcnt(0) before every group of reads: each DMA piece issued between the MFMAs
// was waited for at once (wgrad_bf16v3.hip has the story).  The unit's counted vmcnt + barrier orders DMA writes and reads; the reads'
// own completion is waited for by WG_WAIT_* below, whose operands tie the wait to the registers it guards.
struct WgFrag { bf16x4 lo, hi; };
__device__ __forceinline__ void wg_tr_issue(WgFrag& f, unsigned lds_addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:512" : "=&v"(f.lo), "=&v"(f.hi) : "v"(lds_addr));
}
__device__ __forceinline__ bf16x8 wg_frag(const WgFrag& f) { return __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7); }
#define WG_WAIT5(a_, b0_, b1_, b2_, b3_)                                                                                  \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_.lo), "+v"(a_.hi), "+v"(b0_.lo), "+v"(b0_.hi), "+v"(b1_.lo), "+v"(b1_.hi), \
               "+v"(b2_.lo), "+v"(b2_.hi), "+v"(b3_.lo), "+v"(b3_.hi))

__global__ __launch_bounds__(512, 2) void conv_wgrad_bf16v2_kernel(WgV2Args a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * WG_UB];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2, wn = (wave >> 1) & 1, wc = wave & 1;

  // ---- work item: XCD-aware walk over (split, tile) ----
  const int per_xcd = gridDim.x >> 3;
  const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (item >= a.splits * a.tiles) return;
  const int split_id = item / a.tiles, tile = item - split_id * a.tiles;
  const int nb = tile / a.tiles_c, cb = tile - nb * a.tiles_c;
  const int c_blk = cb * 64, n_blk = nb * 64;
  const int u_begin = split_id * a.units_per_split;
  const int u_end = min(a.total_units, u_begin + a.units_per_split);

  // ---- DMA roles.  Instruction i covers LDS pixel rows 8i .. 8i+7; lane -> row 8i + (lane>>3), physical 16-byte slot lane&7,
  //      which holds logical slot (lane&7) ^ 4*((row>>1)&1) of that pixel ----
  const int sl = lane & 7;
  // (buffer_load ... lds: the descriptor base is the unit's first HALO pixel -- a wave-uniform pointer that may lie before the
  // tensor for border units; lanes outside the image or beyond the valid channels get an out-of-range offset, which the
  // range check turns into zeros in LDS)
  unsigned xoff[4]; int xhy[4], xhx[4];              // byte offset relative to the unit's first halo pixel; halo (row, column)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = wave + 8 * q, r = 8 * i + (lane >> 3);
    const int hy = r / WG_HW, hx = r - hy * WG_HW;
    const int ls = sl ^ (((r >> 1) & 1) << 2);
    const bool ok = i < WG_XI && r < WG_HP && c_blk + ls * 8 < a.x_cvalid;
    xoff[q] = (unsigned)((hy * a.W + hx) * a.x_cs + ls * 8) * 2u;
    xhy[q] = ok ? hy : (1 << 24);                    // never inside the image: zero-filled
    xhx[q] = hx;
  }
  unsigned yoff[2]; int yhy[2], yhx[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = wave + 8 * q, r = 8 * i + (lane >> 3);
    const int py = r >> 5, px = r & 31;
    const int ls = sl ^ (((r >> 1) & 1) << 2);
    const bool ok = n_blk + ls * 8 < a.dy_cvalid;
    yoff[q] = (unsigned)((py * a.W + px) * a.dy_cs + ls * 8) * 2u;
    yhy[q] = ok ? py : (1 << 24);
    yhx[q] = px;
  }

// A unit's DMA pieces: ISSUE_PREP computes the unit's wave-uniform descriptors and origin, ISSUE_X(q) / ISSUE_Y(q) issue this
// wave's piece q.  The first unit of a workgroup is issued in one go (ISSUE_UNIT); inside the loop the pieces of unit u+1 are
// issued BETWEEN the MFMAs of unit u (round 3: issued as one burst behind the barrier, every wave of the CU -- all in lock-step --
// stood in VMEM issue for 5-6 pieces while the matrix pipes idled).
#define ISSUE_PREP(u_, buf_)                                                                                           \
  int q_ = (u_);                                                                                                       \
  const int sx_ = q_ % a.units_x; q_ /= a.units_x;                                                                     \
  const int sy_ = q_ % a.units_y;                                                                                      \
  const int img_ = q_ / a.units_y;                                                                                     \
  const int y0_ = sy_ * 4, x0_ = sx_ * 32;                                                                             \
  const hpri_rsrc_t rx_ = HPRI_MAKE_RSRC(a.xp + ((long long)(img_ * a.H + y0_ - 1) * a.W + x0_ - 1) * a.x_cs + a.x_coff + c_blk, 0x7FFFFF00); \
  const hpri_rsrc_t ry_ = HPRI_MAKE_RSRC(a.dyp + ((long long)(img_ * a.H + y0_) * a.W + x0_) * a.dy_cs + a.dy_coff + n_blk, 0x7FFFFF00);     \
  unsigned char* lx_ = smem + (buf_) * WG_UB;                                                                          \
  (void)rx_; (void)ry_; (void)lx_;
#define ISSUE_X(q)                                                                                                     \
  {                                                                                                                    \
    const int i_ = wave + 8 * (q);                                                                                     \
    if (i_ < WG_XI) {                                                                                                  \
      const int iy_ = y0_ - 1 + xhy[q], ix_ = x0_ - 1 + xhx[q];                                                        \
      const bool in_ = (unsigned)iy_ < (unsigned)a.H && (unsigned)ix_ < (unsigned)a.W;                                 \
      HPRI_LDS_DMA16(rx_, lx_ + i_ * 1024, in_ ? xoff[q] : HPRI_DMA_OOB, 0);                                           \
    }                                                                                                                  \
  }
#define ISSUE_Y(q)                                                                                                     \
  {                                                                                                                    \
    const int i_ = wave + 8 * (q);                                                                                     \
    const int iy_ = y0_ + yhy[q], ix_ = x0_ + yhx[q];                                                                  \
    const bool in_ = iy_ < a.H && ix_ < a.W;                                                                           \
    HPRI_LDS_DMA16(ry_, lx_ + WG_XB + i_ * 1024, in_ ? yoff[q] : HPRI_DMA_OOB, 0);                                     \
  }
#define ISSUE_UNIT(u_, buf_)                                                                                           \
  {                                                                                                                    \
    ISSUE_PREP(u_, buf_)                                                                                               \
    ISSUE_X(0) ISSUE_X(1) ISSUE_X(2) ISSUE_X(3) ISSUE_Y(0) ISSUE_Y(1)                                                  \
  }

  // ---- transposed-read lane roles: 16-lane group = (k half lh, channel half lg); lane i of the group addresses pixel row
  //      lq = i>>2, channel quad lp = i&3 and receives channel i of its 16 ----
  const int lg = (lane >> 4) & 1, lh = lane >> 5, lq = (lane >> 2) & 3, lp = lane & 3;
  const int li = lane & 31;
  const int L = 8 * lh + lq;
  const int eL = (L >> 1) & 1, oL = ((L + 1) >> 1) & 1;
  const int inrow = lg * 32 + lp * 8;
  // X: LDS pixel row = T + L (T = the tap's window start, compile time).  Half swap of row T+L: bit 1 of (T+L) =
  // tbit ^ eL for even T, tbit ^ oL for odd T, with tbit = bit 1 of T resp. T-1
  const int xE = L * 128 + ((wc ^ eL) << 6) + inrow;
  const int xO = L * 128 + ((wc ^ oL) << 6) + inrow;
  const int yB = L * 128 + ((wn ^ eL) << 6) + inrow + WG_XB;      // dY windows start at multiples of 16: even, tbit 0

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  if (u_begin < u_end) ISSUE_UNIT(u_begin, 0)
  for (int u = u_begin; u < u_end; ++u) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // unit u has landed for everyone; everyone has finished reading unit u-1
    const int bo = ((u - u_begin) & 1) * WG_UB;
    const bool more = u + 1 < u_end;
    ISSUE_PREP(more ? u + 1 : u, ((u + 1 - u_begin) & 1))
#ifdef WG_BUILTIN_TR
    const unsigned char* sb = smem + bo;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int py_l = k4 >> 1, pxo = 16 * (k4 & 1);
      const unsigned char* yrow = sb + yB + kg * (64 * 128) + (py_l * 32 + pxo) * 128;
      const bf16x8 af = wg_tr_frag(yrow);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t - dy * 3;
        const int T0 = (py_l + dy) * WG_HW + pxo + dx;
        const int odd = T0 & 1, tbit = ((T0 - odd) >> 1) & 1;
        const int base = (odd ? xO : xE) ^ (tbit << 6);
        const bf16x8 bfr = wg_tr_frag(sb + base + kg * (2 * WG_HW * 128) + T0 * 128);
        acc[t] = HPRI_MFMA_32X32X16(af, bfr, acc[t], 0, 0, 0);
        if (more) {
          const int m = k4 * 9 + t;
          if (m == 2) ISSUE_X(0)
          if (m == 8) ISSUE_X(1)
          if (m == 14) ISSUE_X(2)
          if (m == 20) ISSUE_X(3)
          if (m == 26) ISSUE_Y(0)
          if (m == 32) ISSUE_Y(1)
        }
      }
    }
#else
    // k16-step 4 kg + k4 of the unit; kg is a run-time (wave-uniform) value: its offsets go into the bases, not the constants.
    // X window start T = (2 kg + py_l + dy) * 34 + pxo + dx: 68 kg is a multiple of 4 -> parity and bit 1 of T are those of
    // T0 = (py_l + dy) * 34 + pxo + dx; the half swap (^ 64) commutes with the additions (all multiples of 128).
    const unsigned ya = lds0 + bo + yB + kg * (64 * 128);
    const unsigned xe = lds0 + bo + xE + kg * (2 * WG_HW * 128), xo = lds0 + bo + xO + kg * (2 * WG_HW * 128);
#define WG_XADDR(K4_, T_)                                                                                              \
  ((((((K4_) >> 1) + (T_) / 3) * WG_HW + 16 * ((K4_) & 1) + (T_) % 3) & 1 ? xo : xe) ^                                  \
   (((((((K4_) >> 1) + (T_) / 3) * WG_HW + 16 * ((K4_) & 1) + (T_) % 3) >> 1) & 1) << 6)) +                             \
      ((((K4_) >> 1) + (T_) / 3) * WG_HW + 16 * ((K4_) & 1) + (T_) % 3) * 128
#define WG_YADDR(K4_) (ya + ((((K4_) >> 1) * 32 + 16 * ((K4_) & 1)) * 128))
    // Two read groups per k16-step, each waited for as a whole (lgkmcnt(0): scalar loads share the counter, so no counted waits):
    // group A = the dY fragment + taps 0-3, group B = taps 4-8.  B arrives under the MFMAs of taps 0-3, the next step's A under 4-8.
    WgFrag af[2], ba[4], bb[5];
    wg_tr_issue(af[0], WG_YADDR(0));
#pragma unroll
    for (int t = 0; t < 4; ++t) wg_tr_issue(ba[t], WG_XADDR(0, t));
    WG_WAIT5(af[0], ba[0], ba[1], ba[2], ba[3]);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
#pragma unroll
      for (int t = 4; t < 9; ++t) wg_tr_issue(bb[t - 4], WG_XADDR(k4, t));
      const bf16x8 a8 = wg_frag(af[k4 & 1]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t] = HPRI_MFMA_32X32X16(a8, wg_frag(ba[t]), acc[t], 0, 0, 0);
        if (more) {
          const int m = k4 * 9 + t;
          if (m == 2) ISSUE_X(0)
          if (m == 20) ISSUE_X(3)
        }
      }
      WG_WAIT5(bb[0], bb[1], bb[2], bb[3], bb[4]);
      if (k4 < 3) {
        wg_tr_issue(af[(k4 + 1) & 1], WG_YADDR(k4 + 1));
#pragma unroll
        for (int t = 0; t < 4; ++t) wg_tr_issue(ba[t], WG_XADDR(k4 + 1, t));
      }
#pragma unroll
      for (int t = 4; t < 9; ++t) {
        acc[t] = HPRI_MFMA_32X32X16(a8, wg_frag(bb[t - 4]), acc[t], 0, 0, 0);
        if (more) {
          const int m = k4 * 9 + t;
          if (m == 8) ISSUE_X(1)
          if (m == 14) ISSUE_X(2)
          if (m == 26) ISSUE_Y(0)
          if (m == 32) ISSUE_Y(1)
        }
      }
      if (k4 < 3) WG_WAIT5(af[(k4 + 1) & 1], ba[0], ba[1], ba[2], ba[3]);
    }
#undef WG_XADDR
#undef WG_YADDR
#endif
  }
#undef ISSUE_UNIT
#undef ISSUE_PREP
#undef ISSUE_X
#undef ISSUE_Y

  // ---- the two pixel halves meet in LDS: kg 1 hands taps 0-4 to kg 0, kg 0 hands taps 5-8 to kg 1 ----
  float* ex = reinterpret_cast<float*>(smem);          // [tap_local*16 + r][256 threads]
  static_assert(5 * 16 * 256 * 4 <= 2 * WG_UB, "exchange buffer must fit the staging LDS");
  const int t256 = tid & 255;
  __syncthreads();
  if (kg == 1) {
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) ex[(t * 16 + r) * 256 + t256] = acc[t][r];
  }
  __syncthreads();
  if (kg == 0) {
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] += ex[(t * 16 + r) * 256 + t256];
  }
  __syncthreads();
  if (kg == 0) {
#pragma unroll
    for (int t = 5; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) ex[((t - 5) * 16 + r) * 256 + t256] = acc[t][r];
  }
  __syncthreads();
  if (kg == 1) {
#pragma unroll
    for (int t = 5; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] += ex[((t - 5) * 16 + r) * 256 + t256];
  }
  // slab ws[split][t][n][c]: MFMA rows = n (A operand = dY), columns = c (B operand = X); 32 lanes = 128 contiguous bytes
  float* slab = a.ws + (size_t)split_id * 9 * a.Nr * a.Cr;
  const int c = c_blk + wc * 32 + li;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if ((t < 5) == (kg == 0)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n_blk + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        slab[((size_t)t * a.Nr + n) * a.Cr + c] = acc[t][r];
      }
    }
  }
}

static void wgv2_geometry(int N, int H, int W, int Cin_pad, int Cout_pad, WgV2Args* a) {
  a->Cr = hpri_cdiv(Cin_pad, 64) * 64; a->Nr = hpri_cdiv(Cout_pad, 64) * 64;
  a->tiles_c = a->Cr / 64; a->tiles = a->tiles_c * (a->Nr / 64);
  a->units_x = hpri_cdiv(W, 32); a->units_y = hpri_cdiv(H, 4); a->total_units = N * a->units_x * a->units_y;
  // one workgroup per CU: aim at one full round of 256 (split * tile) items, at least 4 units per item
  int s = a->tiles >= 256 ? 1 : (256 + a->tiles / 2) / a->tiles;
  if (s > a->total_units / 4) s = a->total_units / 4;
  if (s < 1) s = 1;
  a->units_per_split = hpri_cdiv(a->total_units, s);
  a->splits = hpri_cdiv(a->total_units, a->units_per_split);
}

// Workspace of hpri_conv_wgrad_bf16v2: splits * 9 * Cr * Nr floats.
extern "C" int hpri_wgrad_bf16v2_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int* splits, int* Cr, int* Nr) {
  WgV2Args a;
  wgv2_geometry(N, H, W, Cin_pad, Cout_pad, &a);
  *splits = a.splits; *Cr = a.Cr; *Nr = a.Nr;
  return HPRI_OK;
}

// Partial weight-gradient slabs of a 3x3 / pad 1 convolution from bf16 planes (plane 0 of X and of dY, NHWC, channel
// strides multiples of 8 elements, 16-byte aligned); finish with hpri_wgrad_reduce_ex(ws, dw, splits, Cr, Nr, ...).
extern "C" int hpri_conv_wgrad_bf16v2(const void* x_planes, int x_cs, int x_coff, int x_cvalid, const void* dy_planes, int dy_cs,
                                      int dy_coff, int dy_cvalid, float* ws, size_t ws_floats, int N, int H, int W, int Cin_pad,
                                      int Cout_pad, hipStream_t stream) {
  HPRI_REQUIRE(x_planes && dy_planes && ws, "conv_wgrad_bf16v2: null pointer");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0 && Cin_pad > 0 && Cout_pad > 0, "conv_wgrad_bf16v2: empty problem");
  HPRI_REQUIRE(x_cs % 8 == 0 && x_coff % 8 == 0 && x_cvalid % 8 == 0 && dy_cs % 8 == 0 && dy_coff % 8 == 0 && dy_cvalid % 8 == 0,
               "conv_wgrad_bf16v2: channel strides / offsets / valid widths must be multiples of 8 (16-byte DMA granules)");
  HPRI_REQUIRE(x_coff + x_cvalid <= x_cs && dy_coff + dy_cvalid <= dy_cs, "conv_wgrad_bf16v2: valid channels exceed the channel stride");
  HPRI_REQUIRE(((uintptr_t)x_planes & 15) == 0 && ((uintptr_t)dy_planes & 15) == 0, "conv_wgrad_bf16v2: planes must be 16-byte aligned");
  HPRI_REQUIRE((long long)6 * W * x_cs < (1ll << 31) && (long long)4 * W * dy_cs < (1ll << 31), "conv_wgrad_bf16v2: image rows too long");
  WgV2Args a;
  wgv2_geometry(N, H, W, Cin_pad, Cout_pad, &a);
  if ((size_t)a.splits * 9 * a.Cr * a.Nr > ws_floats) return hpri_set_error(HPRI_ERR_WORKSPACE, "conv_wgrad_bf16v2: workspace too small");
  a.xp = reinterpret_cast<const h16_t*>(x_planes); a.x_cs = x_cs; a.x_coff = x_coff; a.x_cvalid = x_cvalid;
  a.dyp = reinterpret_cast<const h16_t*>(dy_planes); a.dy_cs = dy_cs; a.dy_coff = dy_coff; a.dy_cvalid = dy_cvalid;
  a.ws = ws; a.N = N; a.H = H; a.W = W;
  const int items = a.splits * a.tiles;
  dim3 grid((unsigned)(hpri_cdiv(items, 8) * 8), 1u, 1u);
  hipLaunchKernelGGL(conv_wgrad_bf16v2_kernel, grid, dim3(512), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}
