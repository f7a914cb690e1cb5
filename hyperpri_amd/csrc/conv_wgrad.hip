// Weight-gradient GEMMs on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32):
//   dW[tap][c][n] = sum over pixels p of  X[p + off(tap)][c] * dY[p][n]
// i.e. the K dimension of the GEMM is the pixel index.  Serves the 3x3 convs (model_parts.py:22,25;
// models.py:169,177), Linear (models.py:108,143; KS=1) and ConvTranspose2d k2s2 (model_parts.py:63;
// KS=1 with the dY operand gathered as 2x2 stride-2 patches, B_S2D).
//
// Work split: grid.y x grid.z tiles the (c, n) output, grid.x splits the pixel range ("split-K") into
// contiguous runs of 2x32-pixel strips.  Every workgroup keeps its KS*KS x (c-tile x n-tile) output in
// accumulators for its whole run and writes ONE partial slab; hpri_wgrad_reduce sums the slabs in a fixed
// order (deterministic -- no float atomics) straight into the OIHW / (Cin,Cout,2,2) gradient tensor.
//
// Per strip the X halo ((2+KS-1) x (32+KS-1) pixels x BC channels) and the dY strip (64 pixels x BNW
// channels) are staged in LDS as [pixel][channel]; MFMA lanes index the channel, so every ds_read_b32 is
// 32 consecutive dwords (conflict-free) and the NHWC global loads are full 128-byte lines.
#include "common.h"
#include <stdlib.h>

struct WgradArgs {
  const float* x; int x_cs; int x_coff; int x_cvalid;   // conv input (NHWC), readable channels
  const float* dy; int dy_cs; int dy_coff; int dy_cvalid; // gradient of the conv output
  float* ws;             // [splits][T][Nr][Cr] partial slabs
  int N, H, W;
  int strips_x, strips_y, total_strips, strips_per_split;
  int Cr, Nr;            // padded slab dims (gridDim.y*BC, gridDim.z*BNW)
  int H2, W2, py0, px0, Cup;  // S2D geometry for dY (convT): hi-res dims, pad offsets, channels per tap
  int splits;            // pixel splits (= slabs)
  int xcd_tiles;         // >0: XCD-aware 1-D grid (wgrad_block_ids); = cblk * nblk
  int cblk;              // channel blocks along C (for the 1-D grid decode)
};

// Block -> (pixel split, C block, N block).  Legacy grid (splits, cblk, nblk).  With many (C, N) tiles (SpectralUNET:
// 13 x 13 and 26 x 13 tiles of 128 x 128) every tile streams the whole X and dY from HBM.  xcd_tiles > 0: 1-D grid of
// 8k * tiles workgroups; they go round-robin over the 8 XCDs in launch order, so id%8 picks the XCD; each XCD owns the
// pixel splits = xcd (mod 8) and walks the (C, N) tiles of one split back to back (C fastest): the ~64 workgroups
// resident on an XCD read the SAME pixel strips, which its L2 then serves 5-13 times.
__device__ __forceinline__ void wgrad_block_ids(const WgradArgs& a, int& split, int& cb, int& nb) {
  split = blockIdx.x; cb = blockIdx.y; nb = blockIdx.z;
  if (a.xcd_tiles > 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % a.xcd_tiles;
    split = (j / a.xcd_tiles) * 8 + xcd;
    cb = tile % a.cblk;
    nb = tile / a.cblk;
  }
}

template <int KS, int CT, int NT, int BMODE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int T = KS * KS, PAD = KS / 2;
  constexpr int SH = 2, SW = 32, HH = SH + KS - 1, HW = SW + KS - 1, HP = HH * HW;
  constexpr int BC = 64 * CT, BNW = 64 * NT;
  constexpr int NLD_X = (HP * (BC / 4) + 255) / 256;
  constexpr int NLD_Y = (64 * (BNW / 4)) / 256;
  __shared__ __attribute__((aligned(16))) float smem[HP * BC + 64 * BNW];
  float* x_lds = smem;
  float* y_lds = smem + HP * BC;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wc = wave >> 1, wn = wave & 1;
  int split_id, cb_id, nb_id;
  wgrad_block_ids(a, split_id, cb_id, nb_id);
  const int c_blk = cb_id * BC, n_blk = nb_id * BNW;

  f32x16 acc[T][CT][NT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

  const int s_begin = split_id * a.strips_per_split;
  const int s_end = min(a.total_strips, s_begin + a.strips_per_split);

  // KS == 1 (Linear / ConvTranspose2d weight gradients: no halo reuse, 128 MFMAs per staged strip) prefetches the next
  // strip into registers while the current one is being multiplied; KS == 3 has no registers to spare (144 accumulators)
  // and relies on the second resident workgroup to cover its loads.
  constexpr bool PREF = (KS == 1);
  f32x4 xr[NLD_X], yr[NLD_Y];
#define LOAD_STRIP(st_)                                                                               \
  {                                                                                                   \
    int q_ = (st_);                                                                                   \
    const int sx = q_ % a.strips_x; q_ /= a.strips_x;                                                 \
    const int sy = q_ % a.strips_y;                                                                   \
    const int img = q_ / a.strips_y;                                                                  \
    const int y0 = sy * SH, x0 = sx * SW;                                                             \
    _Pragma("unroll") for (int p = 0; p < NLD_X; ++p) {                                               \
      const int f = tid + p * 256;                                                                    \
      const int pix = f / (BC / 4), c4 = f % (BC / 4);                                                \
      f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                                 \
      if (pix < HP) {                                                                                 \
        const int hy = pix / HW, hx = pix - hy * HW;                                                  \
        const int iy = y0 + hy - PAD, ix = x0 + hx - PAD;                                             \
        const int c = c_blk + c4 * 4;                                                                 \
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && c < a.x_cvalid)                             \
          v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)(img * a.H + iy) * a.W + ix) * a.x_cs + a.x_coff + c); \
      }                                                                                               \
      xr[p] = v;                                                                                      \
    }                                                                                                 \
    _Pragma("unroll") for (int p = 0; p < NLD_Y; ++p) {                                               \
      const int f = tid + p * 256;                                                                    \
      const int pix = f / (BNW / 4), n4 = f % (BNW / 4);                                              \
      const int iy = y0 + (pix >> 5), ix = x0 + (pix & 31);                                           \
      const int n = n_blk + n4 * 4;                                                                   \
      f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                                 \
      if (iy < a.H && ix < a.W && n < a.dy_cvalid) {                                                  \
        if (BMODE == HPRI_A_DIRECT) {                                                                 \
          v = *reinterpret_cast<const f32x4*>(a.dy + ((size_t)(img * a.H + iy) * a.W + ix) * a.dy_cs + a.dy_coff + n); \
        } else {                                                                                      \
          const int tap = n / a.Cup, co = n - tap * a.Cup;                                            \
          const int yy = 2 * iy + (tap >> 1) + a.py0, xx = 2 * ix + (tap & 1) + a.px0;                \
          v = *reinterpret_cast<const f32x4*>(a.dy + ((size_t)(img * a.H2 + yy) * a.W2 + xx) * a.dy_cs + a.dy_coff + co); \
        }                                                                                             \
      }                                                                                               \
      yr[p] = v;                                                                                      \
    }                                                                                                 \
  }
  if (PREF && s_begin < s_end) LOAD_STRIP(s_begin)
  for (int st = s_begin; st < s_end; ++st) {
    if (!PREF) LOAD_STRIP(st)
    __syncthreads();   // previous strip's LDS reads are finished
#pragma unroll
    for (int p = 0; p < NLD_X; ++p) {
      const int f = tid + p * 256;
      if (f < HP * (BC / 4)) *reinterpret_cast<f32x4*>(x_lds + f * 4) = xr[p];
    }
#pragma unroll
    for (int p = 0; p < NLD_Y; ++p) {
      const int f = tid + p * 256;
      *reinterpret_cast<f32x4*>(y_lds + f * 4) = yr[p];
    }
    __syncthreads();
    if (PREF && st + 1 < s_end) LOAD_STRIP(st + 1)    // lands during this strip's MFMAs

    // k-step ks covers pixels 2*ks (lanes 0-31) and 2*ks+1 (lanes 32-63) of the 2x32 strip
    const float* xb = x_lds + lh * BC + wc * (CT * 32) + li;
    const float* yb = y_lds + lh * BNW + wn * (NT * 32) + li;
#pragma unroll
    for (int py = 0; py < SH; ++py) {
#pragma unroll 4
      for (int kx = 0; kx < 16; ++kx) {
        float bf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = yb[(py * 32 + 2 * kx) * BNW + j * 32];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int dy = t / KS, dx = t - dy * KS;
          float af[CT];
#pragma unroll
          for (int i = 0; i < CT; ++i) af[i] = xb[((py + dy) * HW + 2 * kx + dx) * BC + i * 32];
#pragma unroll
          for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j], af[i], acc[t][i][j], 0, 0, 0);
        }
      }
    }
  }

#undef LOAD_STRIP
  // partial slab: ws[split][t][n][c]; MFMA rows = n (A operand = dY), cols = c (B operand = X) so that
  // lanes store consecutive c -- the order the reduce kernel and the OIHW gradient want
  float* slab = a.ws + (size_t)split_id * T * a.Cr * a.Nr;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int c = c_blk + wc * (CT * 32) + i * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n_blk + wn * (NT * 32) + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          slab[((size_t)t * a.Nr + n) * a.Cr + c] = acc[t][i][j][r];
        }
      }
}

// -------------------------------------------------------------------------------------------------------------
// bf16 variant (precision mode "bf16"): X and dY are rounded to bf16 while staged into LDS as [pixel][channel]
// (192 / 320-byte pitch); the MFMA operands need 8 consecutive PIXELS per lane, i.e. the transpose of that image,
// which gfx950's ds_read_b64_tr_b16 delivers for free (4 pixels x 16 channels per 16-lane group, column-major).
// v_mfma_f32_32x32x16_bf16, fp32 accumulate; same slab layout and fixed-order reduce as the fp32 kernel.
typedef h16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef h16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

__device__ __forceinline__ bf16x8 tr_frag(const h16_t* p, int pitch4) {
  // p: this lane's address for pixel row (8h + q); rows +4 further down supply k-slots 4..7
  const bf16x4 lo = __builtin_bit_cast(bf16x4, HPRI_DS_READ_TR16_B64((hpri_lds_tr4_ptr)(p)));
  const bf16x4 hi = __builtin_bit_cast(bf16x4, HPRI_DS_READ_TR16_B64((hpri_lds_tr4_ptr)(p + pitch4)));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// SPLIT = 1 (precision mode "bf16x3"): X and dY are staged as two bf16 planes (hi, lo) and every k16-step issues
// dYl*Xh + dYh*Xl + dYh*Xh.
// KR = kernel rows per workgroup.  KR == KS: all taps in one workgroup (144 accumulators for 3x3: two workgroups per CU
// and no registers left to prefetch).  KR == 1: blockIdx picks ONE kernel row (3 taps, 48 accumulators, a 2-row input
// strip without vertical halo): 50.7 KB of LDS and ~160 VGPRs, i.e. THREE workgroups per CU plus a register prefetch of
// the next unit -- the 3x3 bf16 kernels were waiting for their global loads two thirds of the time.
template <int KS, int CT, int NT, int BMODE, int SPLIT, int KR>
__global__ __launch_bounds__(256, (KS == 3 && KR == 1 && SPLIT < 2) ? 3 : 2) void conv_wgrad_bf16_kernel(WgradArgs a) {
  constexpr int T = KR * KS, PAD = KS / 2, RS = KS / KR;   // T: taps of THIS workgroup; RS: kernel-row groups
  // one staged unit = SQ vertically adjacent 2x32 strips (SQ*64 pixels): the bf16 MFMAs retire so fast that the
  // staging + barrier cost must be amortised over more pixels than in the fp32 kernel
  constexpr int SQ = (KS == 3 && !SPLIT) ? 2 : 1;
  constexpr int NPL = SPLIT + 1;                   // operand planes: hi | hi, lo | hi, mid, lo
  constexpr int NTERM = (SPLIT == 0) ? 1 : (SPLIT == 1) ? 3 : 6;
  constexpr int TA[6] = {2, 0, 1, 1, 0, 0};       // (dY plane, X plane) per MFMA, smallest products first; the LAST NTERM
  constexpr int TB[6] = {0, 2, 1, 0, 1, 0};       // entries are used
  constexpr int SH = 2 * SQ, SW = 32, HH = SH + KR - 1, HW = SW + KS - 1, HP = HH * HW, NPIX = SH * SW;
  constexpr int BC = 64 * CT, BNW = 64 * NT;
  constexpr int PX = BC + 32, PY = BNW + 32;      // halves per staged pixel: data + 32 pad, i.e. a pitch of 48 / 80 dwords
                                                  // = 16 (mod 32): the 4 pixel rows of a transposed read hit disjoint banks
  constexpr int NLD_X = (HP * (BC / 4) + 255) / 256;
  constexpr int NLD_Y = (NPIX * (BNW / 4)) / 256;
  __shared__ __attribute__((aligned(16))) h16_t smem[NPL * (HP * PX + NPIX * PY)];
  h16_t* x_lds = smem;
  h16_t* y_lds = smem + NPL * HP * PX;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave >> 1, wn = wave & 1;
  int split_id, cb_id, nb_id;
  wgrad_block_ids(a, split_id, cb_id, nb_id);
  const int r0 = (RS > 1) ? nb_id % RS : 0;        // first kernel row of this workgroup
  if (RS > 1) nb_id /= RS;
  const int c_blk = cb_id * BC, n_blk = nb_id * BNW;
  // transposed-read lane roles: 16-lane group = (k half h, channel half g); lane i of the group addresses pixel row
  // q = i>>2, channel quad p = i&3 and receives channel i of the block
  const int lg = (lane >> 4) & 1, lh = lane >> 5, lq = (lane >> 2) & 3, lp = lane & 3;
  const int li = lane & 31;

  f32x16 acc[T][CT][NT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

  // strips_y counts 2-row strips; this kernel walks them SQ at a time: unit = (img, unit row, strip column)
  const int units_y = (a.strips_y + SQ - 1) / SQ;
  const int total_units = a.N * units_y * a.strips_x;
  const int units_per_split = (total_units + a.splits - 1) / a.splits;
  const int u_begin = split_id * units_per_split;
  const int u_end = min(total_units, u_begin + units_per_split);
  const h16_t* xb = x_lds + (lh * 8 + lq) * PX + wc * (CT * 32) + lg * 16 + lp * 4;
  const h16_t* yb = y_lds + (lh * 8 + lq) * PY + wn * (NT * 32) + lg * 16 + lp * 4;

  // The unit's global loads are converted to bf16 at once (half the staging registers).  KS == 1 (64 accumulators) and
  // the one-kernel-row 3x3 form (48) fetch the NEXT unit while the current one is multiplied; with all nine taps in one
  // workgroup (144 accumulators) there are no registers for that (measured: 124 spills, or -6 %).
  constexpr bool PREF = (KS == 1) || (KR < KS);
  bf16x4 xr[NPL][NLD_X], yr[NPL][NLD_Y];
#define LOAD_UNIT(st_)                                                                               \
  {                                                                                                  \
    int q = (st_);                                                                                   \
    const int sx = q % a.strips_x; q /= a.strips_x;                                                  \
    const int sy = q % units_y;                                                                      \
    const int img = q / units_y;                                                                     \
    const int y0 = sy * SH, x0 = sx * SW;                                                            \
                                                                                                     \
_Pragma("unroll")                                                                                    \
    for (int p = 0; p < NLD_X; ++p) {                                                                \
      const int f = tid + p * 256;                                                                   \
      const int pix = f / (BC / 4), c4 = f % (BC / 4);                                               \
      f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                                \
      if (pix < HP) {                                                                                \
        const int hy = pix / HW, hx = pix - hy * HW;                                                 \
        const int iy = y0 + hy - PAD + r0, ix = x0 + hx - PAD;                                       \
        const int c = c_blk + c4 * 4;                                                                \
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && c < a.x_cvalid)                            \
          v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)(img * a.H + iy) * a.W + ix) * a.x_cs + a.x_coff + c);\
      }                                                                                              \
      _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                                           \
        xr[pl][p] = __builtin_convertvector(v, bf16x4);                                              \
        if (pl + 1 < NPL) v = v - __builtin_convertvector(xr[pl][p], f32x4);                         \
      }                                                                                              \
    }                                                                                                \
_Pragma("unroll")                                                                                    \
    for (int p = 0; p < NLD_Y; ++p) {                                                                \
      const int f = tid + p * 256;                                                                   \
      const int pix = f / (BNW / 4), n4 = f % (BNW / 4);                                             \
      const int iy = y0 + (pix >> 5), ix = x0 + (pix & 31);                                          \
      const int n = n_blk + n4 * 4;                                                                  \
      f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                                \
      if (iy < a.H && ix < a.W && n < a.dy_cvalid) {                                                 \
        if (BMODE == HPRI_A_DIRECT) {                                                                \
          v = *reinterpret_cast<const f32x4*>(a.dy + ((size_t)(img * a.H + iy) * a.W + ix) * a.dy_cs + a.dy_coff + n);\
        } else {                                                                                     \
          const int tap = n / a.Cup, co = n - tap * a.Cup;                                           \
          const int yy = 2 * iy + (tap >> 1) + a.py0, xx = 2 * ix + (tap & 1) + a.px0;               \
          v = *reinterpret_cast<const f32x4*>(a.dy + ((size_t)(img * a.H2 + yy) * a.W2 + xx) * a.dy_cs + a.dy_coff + co);\
        }                                                                                            \
      }                                                                                              \
      _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                                           \
        yr[pl][p] = __builtin_convertvector(v, bf16x4);                                              \
        if (pl + 1 < NPL) v = v - __builtin_convertvector(yr[pl][p], f32x4);                         \
      }                                                                                              \
    }                                                                                                \
  }
  if (PREF && u_begin < u_end) LOAD_UNIT(u_begin)
  for (int st = u_begin; st < u_end; ++st) {
    if (!PREF) LOAD_UNIT(st)
    __syncthreads();   // previous unit's LDS reads are finished
#pragma unroll
    for (int p = 0; p < NLD_X; ++p) {
      const int f = tid + p * 256;
      const int pix = f / (BC / 4), c4 = f % (BC / 4);
      if (pix < HP) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<bf16x4*>(x_lds + pl * HP * PX + pix * PX + c4 * 4) = xr[pl][p];
      }
    }
#pragma unroll
    for (int p = 0; p < NLD_Y; ++p) {
      const int f = tid + p * 256;
      const int pix = f / (BNW / 4), n4 = f % (BNW / 4);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<bf16x4*>(y_lds + pl * NPIX * PY + pix * PY + n4 * 4) = yr[pl][p];
    }
    __syncthreads();
    if (PREF && st + 1 < u_end) LOAD_UNIT(st + 1)

    // NPIX/16 k16-steps: unit pixels 16*ks .. 16*ks+15 (row ks>>1, columns 16*(ks&1) ..)
#pragma unroll
    for (int ks = 0; ks < NPIX / 16; ++ks) {
      const int py = ks >> 1, pxo = 16 * (ks & 1);
      bf16x8 af[NPL][NT];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < NT; ++j) af[pl][j] = tr_frag(yb + pl * NPIX * PY + (ks * 16) * PY + j * 32, 4 * PY);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int dy = t / KS, dx = t - dy * KS;
        bf16x8 bfr[NPL][CT];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int i = 0; i < CT; ++i) bfr[pl][i] = tr_frag(xb + pl * HP * PX + ((py + dy) * HW + pxo + dx) * PX + i * 32, 4 * PX);
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 6 - NTERM; q < 6; ++q)
              acc[t][i][j] = HPRI_MFMA_32X32X16(af[TA[q]][j], bfr[TB[q]][i], acc[t][i][j], 0, 0, 0);
      }
    }
  }

#undef LOAD_UNIT
  // partial slab ws[split][t][n][c]: MFMA rows = n (A operand = dY), cols = c (B operand = X)
  float* slab = a.ws + (size_t)split_id * (KS * KS) * a.Cr * a.Nr;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int c = c_blk + wc * (CT * 32) + i * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n_blk + wn * (NT * 32) + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          slab[((size_t)(r0 * KS + t) * a.Nr + n) * a.Cr + c] = acc[t][i][j][r];
        }
      }
}

// Fixed-order reduction of the partial slabs into the parameter gradient.
//   block = 32 c-lanes x 32 split-slices (1024 threads); one block per (n, 32-channel tile); every thread keeps the
//   T taps of its (n, c) in registers, the slices are combined through LDS in slice order (deterministic), and
//   the block's 32*T results -- contiguous in OIHW -- are written coalesced.
// dst modes: 0 = conv weight OIHW dW[n][c][t];  1 = convT weight dW[c][co][tap] with n = tap*Cup + co (T == 1)
template <int T>
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits,
                                                            int Cr, int Nr, int Cin, int Cout, int mode, int Cup,
                                                            int accumulate) {
  __shared__ float red[32][32 * T + 1];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int n = blockIdx.y, c = blockIdx.x * 32 + cl;
  const size_t slab = (size_t)T * Cr * Nr;
  float acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.f;
  if (c < Cin) {
    const float* p = ws + (size_t)n * Cr + c;
    for (int k = sl; k < splits; k += 32) {
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] += p[(size_t)k * slab + (size_t)t * Nr * Cr];
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) red[sl][cl * T + t] = acc[t];
  __syncthreads();
  // 32*T outputs per block: thread o sums the 32 slices of output o = cl*T + t
  for (int o = threadIdx.x; o < 32 * T; o += 1024) {
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) s += red[k][o];
    const int cc = blockIdx.x * 32 + o / T, t = o % T;
    if (cc < Cin) {
      size_t off;
      if (mode == 0) off = ((size_t)n * Cin + cc) * T + t;
      else { const int tap = n / Cup, co = n - tap * Cup; off = ((size_t)cc * Cup + co) * 4 + tap; }
      dw[off] = accumulate ? dw[off] + s : s;
    }
  }
}

// 3x3 slabs ws[splits][9][Nr][Cr] -> dW[n][c][9] (OIHW), bandwidth-shaped (round 3).  The generic kernel above gives one
// 1024-thread block 32 channels of one output row and reads 128-byte pieces with splits/32 of its threads: 0.5 TB/s on the
// 37.7 MB of slabs every layer of a C2 step leaves (1.1 ms per step in the bf16 mode).  Here a block owns ONE output row n and a
// run of 1024 / S channels; its 256 threads are (256 / S channel quads) x (S slab slices): every thread streams float4s of its
// slices for all nine taps (all loads independent), the slices meet in LDS in slice order (fixed: deterministic), and the
// block writes its [channel][tap] run of dW contiguously.  S = 16 / 4 / 1 by the number of slabs.
template <int S, int T = 9>
__global__ __launch_bounds__(256) void wgrad_reduce9_wide_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits,
                                                                 int Cr, int Nr, int Cin, int Cout, int accumulate) {
  constexpr int CQ = 256 / S, CR = CQ * 4;               // channel quads / channels per block
  __shared__ float red[S][T][CR];
  const int cq = threadIdx.x % CQ, sl = threadIdx.x / CQ;
  const int n = blockIdx.y, c0 = blockIdx.x * CR, c = c0 + cq * 4;
  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (c < Cr) {
    const size_t slab = (size_t)T * Cr * Nr, row = (size_t)Nr * Cr;
    const float* p = ws + (size_t)n * Cr + c;
    if (T == 1) {
      // one tap (the 1x1 layers: wgrad_bf16v3.hip): four slabs of a slice in flight instead of nine taps of one
      int k = sl;
      for (; k + 3 * S < splits; k += 4 * S) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(k + u * S) * slab);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[0] += v[u];
      }
      for (; k < splits; k += S) acc[0] += *reinterpret_cast<const f32x4*>(p + (size_t)k * slab);
    } else {
      for (int k = sl; k < splits; k += S) {
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] += *reinterpret_cast<const f32x4*>(p + (size_t)k * slab + (size_t)t * row);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) *reinterpret_cast<f32x4*>(&red[sl][t][cq * 4]) = acc[t];
  __syncthreads();
  const int nout = min(CR, Cin - c0) * T;                // this block's run of dW: [channel][tap], contiguous
  float* o = dw + ((size_t)n * Cin + c0) * T;
  for (int i = threadIdx.x; i < nout; i += 256) {
    const int cl = i / T, t = i - cl * T;
    float v = red[0][t][cl];
#pragma unroll
    for (int q = 1; q < S; ++q) v += red[q][t][cl];
    o[i] = accumulate ? o[i] + v : v;
  }
}

template <int T>
static void launch_wgrad_reduce_wide(const float* ws, float* dw, int splits, int Cr, int Nr, int Cin, int Cout, int accumulate,
                                     hipStream_t stream) {
  if (splits >= 16)
    hipLaunchKernelGGL((wgrad_reduce9_wide_kernel<16, T>), dim3((unsigned)hpri_cdiv(Cin, 64), (unsigned)Cout), dim3(256), 0, stream, ws, dw,
                       splits, Cr, Nr, Cin, Cout, accumulate);
  else if (splits >= 3)
    hipLaunchKernelGGL((wgrad_reduce9_wide_kernel<4, T>), dim3((unsigned)hpri_cdiv(Cin, 256), (unsigned)Cout), dim3(256), 0, stream, ws, dw,
                       splits, Cr, Nr, Cin, Cout, accumulate);
  else
    hipLaunchKernelGGL((wgrad_reduce9_wide_kernel<1, T>), dim3((unsigned)hpri_cdiv(Cin, 1024), (unsigned)Cout), dim3(256), 0, stream, ws, dw,
                       splits, Cr, Nr, Cin, Cout, accumulate);
}
static void launch_wgrad_reduce9_wide(const float* ws, float* dw, int splits, int Cr, int Nr, int Cin, int Cout, int accumulate,
                                      hipStream_t stream) {
  launch_wgrad_reduce_wide<9>(ws, dw, splits, Cr, Nr, Cin, Cout, accumulate, stream);
}


// ConvTranspose2d(k2, s2) weights (dst mode 1, T == 1): dW[c][co][tap] from slab rows n = tap * Cup + co.  One workgroup owns
// 64 channels x 16 co x 4 taps: every thread sums its four (row, channel) pairs over the slabs in order (256-byte row segments),
// the 64 x 64 tile turns in LDS, and each channel's 16 co x 4 taps leave as 256 contiguous bytes (the generic kernel above wrote
// them as scattered 4-byte stores from 32-lane row segments, with most of its 32 slices idle at 4-16 slabs: 118 -> 9 us for
// up1; used for <= 16 slabs).
__global__ __launch_bounds__(1024) void wgrad_reduce_convt_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits,
                                                                  int Cr, int Nr, int Cin, int Cup, int accumulate) {
  __shared__ float tile[64][65];
  const int cl = threadIdx.x & 63, r = threadIdx.x >> 6;          // channel lane, row group (0..15)
  const int c0 = blockIdx.x * 64, co0 = blockIdx.y * 16;
  const size_t slab = (size_t)Cr * Nr;
  // tile rows of this thread: rr = r + 16 j (tap = j, co = co0 + r), i.e. slab rows n_j = j * Cup + co0 + r
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c0 + cl < Cin) {
    const float* p = ws + (size_t)(co0 + r) * Cr + c0 + cl;
    const size_t tapstep = (size_t)Cup * Cr;
    int k = 0;
    for (; k + 2 <= splits; k += 2) {               // 8 independent loads in flight; each (n, c) still sums its slabs in order
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = p[(size_t)k * slab + j * tapstep];
        v[4 + j] = p[(size_t)(k + 1) * slab + j * tapstep];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[j] += v[j]; acc[j] += v[4 + j]; }
    }
    for (; k < splits; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += p[(size_t)k * slab + j * tapstep];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) tile[r + 16 * j][cl] = acc[j];
  __syncthreads();
  const int c = threadIdx.x >> 4, q = threadIdx.x & 15;            // output: channel c0 + c, co0 + q, taps 0..3 as one float4
  if (c0 + c < Cin) {
    float4 v = make_float4(tile[q][c], tile[16 + q][c], tile[32 + q][c], tile[48 + q][c]);
    float4* o = reinterpret_cast<float4*>(dw + ((size_t)(c0 + c) * Cup + co0 + q) * 4);
    if (accumulate) { const float4 w = *o; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
    *o = v;
  }
}

// XCD-aware grid (wgrad_block_ids) from this many (C, N) tiles on: at least two XCDs' worth of resident workgroups,
// and enough pixel strips that 8k splits still amortise the slab write.  Options "wgrad_xcd_min_tiles" (0 = never) and
// "wgrad_xcd_min_strips" (api.cpp).
static bool wgrad_xcd(int tiles, int total_strips) {
  const int v = hpri_option(1);
  return v > 0 && tiles >= v && total_strips >= hpri_option(2);
}

static inline void wgrad_cfg(int KS, int* bc, int* bn) {
  if (KS == 3) { *bc = 64; *bn = 64; } else { *bc = 128; *bn = 128; }
}

// Number of pixel splits for a given problem; the caller sizes the workspace as
// splits * KS*KS * Cr * Nr floats (hpri_wgrad_workspace).
extern "C" int hpri_wgrad_plan(int N, int H, int W, int Cin_pad, int Cout_pad, int KS,
                               int* splits, int* Cr, int* Nr) {
  int bc, bn; wgrad_cfg(KS, &bc, &bn);
  const int cblk = hpri_cdiv(Cin_pad, bc), nblk = hpri_cdiv(Cout_pad, bn);
  const int total = N * hpri_cdiv(H, 2) * hpri_cdiv(W, 32);
  *Cr = cblk * bc; *Nr = nblk * bn;
  if (wgrad_xcd(cblk * nblk, total)) {            // 8k splits, k such that the grid is close to whole rounds of 512
    const int tiles = cblk * nblk;
    int best = 1; double best_eff = 0.0;
    for (int k = 1; k <= 4; ++k) {
      const double rounds = 8.0 * k * tiles / 512.0, eff = rounds / (double)((long long)(rounds + 0.999999));
      if (eff > best_eff + 1e-9) { best_eff = eff; best = k; }
    }
    *splits = 8 * best;
    return HPRI_OK;
  }
  int s = hpri_cdiv(512, cblk * nblk);           // one round of 256 CUs x 2 resident workgroups
  if (s > total) s = total;
  if (s < 1) s = 1;
  const int per = hpri_cdiv(total, s);
  s = hpri_cdiv(total, per);
  *splits = s;
  return HPRI_OK;
}

extern "C" int hpri_conv_wgrad(const float* x, int x_cs, int x_coff, int x_cvalid,
                               const float* dy, int dy_cs, int dy_coff, int dy_cvalid,
                               float* ws, size_t ws_floats,
                               int N, int H, int W, int Cin_pad, int Cout_pad,
                               int KS, int bmode,
                               int H2, int W2, int py0, int px0, int Cup, hipStream_t stream) {
  HPRI_REQUIRE(x && dy && ws, "conv_wgrad: null pointer");
  HPRI_REQUIRE(KS == 1 || KS == 3, "conv_wgrad: kernel size must be 1 or 3");
  HPRI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0 && dy_cs % 4 == 0 && dy_coff % 4 == 0 && x_cvalid % 4 == 0 && dy_cvalid % 4 == 0,
               "conv_wgrad: channel strides/offsets/valid counts must be multiples of 4");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0 && Cin_pad > 0 && Cout_pad > 0, "conv_wgrad: empty problem");
  if (bmode == HPRI_A_S2D) {
    HPRI_REQUIRE(KS == 1 && Cup > 0 && Cup % 4 == 0, "conv_wgrad: S2D needs KS==1 and Cup % 4 == 0");
    HPRI_REQUIRE(py0 >= 0 && px0 >= 0 && 2 * H + py0 <= H2 && 2 * W + px0 <= W2, "conv_wgrad: patch grid exceeds the hi-res image");
  }
  WgradArgs a;
  a.x = x; a.x_cs = x_cs; a.x_coff = x_coff; a.x_cvalid = x_cvalid;
  a.dy = dy; a.dy_cs = dy_cs; a.dy_coff = dy_coff; a.dy_cvalid = dy_cvalid;
  a.ws = ws; a.N = N; a.H = H; a.W = W;
  a.strips_x = hpri_cdiv(W, 32); a.strips_y = hpri_cdiv(H, 2);
  a.total_strips = N * a.strips_x * a.strips_y;
  int splits, Cr, Nr;
  hpri_wgrad_plan(N, H, W, Cin_pad, Cout_pad, KS, &splits, &Cr, &Nr);
  a.strips_per_split = hpri_cdiv(a.total_strips, splits);
  a.Cr = Cr; a.Nr = Nr; a.H2 = H2; a.W2 = W2; a.py0 = py0; a.px0 = px0; a.Cup = Cup;
  const int T = KS * KS;
  if ((size_t)splits * T * Cr * Nr > ws_floats) return hpri_set_error(HPRI_ERR_WORKSPACE, "conv_wgrad: workspace too small");
  int bc, bn; wgrad_cfg(KS, &bc, &bn);
  dim3 grid((unsigned)splits, (unsigned)(Cr / bc), (unsigned)(Nr / bn));
  a.splits = splits; a.cblk = Cr / bc; a.xcd_tiles = 0;
  if (wgrad_xcd((Cr / bc) * (Nr / bn), a.total_strips)) {
    a.xcd_tiles = (Cr / bc) * (Nr / bn);
    grid = dim3((unsigned)(splits * a.xcd_tiles), 1u, 1u);
  }
  if (KS == 3) hipLaunchKernelGGL((conv_wgrad_kernel<3, 1, 1, HPRI_A_DIRECT>), grid, dim3(256), 0, stream, a);
  else if (bmode == HPRI_A_S2D) hipLaunchKernelGGL((conv_wgrad_kernel<1, 2, 2, HPRI_A_S2D>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<1, 2, 2, HPRI_A_DIRECT>), grid, dim3(256), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// bf16-operand variant of hpri_conv_wgrad (same modes, plan, workspace and reduce).
extern "C" int hpri_conv_wgrad_bf16(const float* x, int x_cs, int x_coff, int x_cvalid,
                                    const float* dy, int dy_cs, int dy_coff, int dy_cvalid,
                                    float* ws, size_t ws_floats,
                                    int N, int H, int W, int Cin_pad, int Cout_pad, int KS, int bmode,
                                    int H2, int W2, int py0, int px0, int Cup, int split, hipStream_t stream) {
  HPRI_REQUIRE(x && dy && ws, "conv_wgrad_bf16: null pointer");
  HPRI_REQUIRE(split >= 0 && split <= 2, "conv_wgrad_bf16: split must be 0 (bf16), 1 (bf16x3) or 2 (bf16x6)");
  HPRI_REQUIRE(KS == 1 || KS == 3, "conv_wgrad_bf16: kernel size must be 1 or 3");
  HPRI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0 && dy_cs % 4 == 0 && dy_coff % 4 == 0 && x_cvalid % 4 == 0 && dy_cvalid % 4 == 0,
               "conv_wgrad_bf16: channel strides/offsets/valid counts must be multiples of 4");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0 && Cin_pad > 0 && Cout_pad > 0, "conv_wgrad_bf16: empty problem");
  if (bmode == HPRI_A_S2D) {
    HPRI_REQUIRE(KS == 1 && Cup > 0 && Cup % 4 == 0, "conv_wgrad_bf16: S2D needs KS==1 and Cup % 4 == 0");
    HPRI_REQUIRE(py0 >= 0 && px0 >= 0 && 2 * H + py0 <= H2 && 2 * W + px0 <= W2, "conv_wgrad_bf16: patch grid exceeds the hi-res image");
  }
  WgradArgs a;
  a.x = x; a.x_cs = x_cs; a.x_coff = x_coff; a.x_cvalid = x_cvalid;
  a.dy = dy; a.dy_cs = dy_cs; a.dy_coff = dy_coff; a.dy_cvalid = dy_cvalid;
  a.ws = ws; a.N = N; a.H = H; a.W = W;
  a.strips_x = hpri_cdiv(W, 32); a.strips_y = hpri_cdiv(H, 2);
  a.total_strips = N * a.strips_x * a.strips_y;
  int splits, Cr, Nr;
  hpri_wgrad_plan(N, H, W, Cin_pad, Cout_pad, KS, &splits, &Cr, &Nr);
  a.strips_per_split = hpri_cdiv(a.total_strips, splits);
  a.Cr = Cr; a.Nr = Nr; a.H2 = H2; a.W2 = W2; a.py0 = py0; a.px0 = px0; a.Cup = Cup;
  const int T = KS * KS;
  if ((size_t)splits * T * Cr * Nr > ws_floats) return hpri_set_error(HPRI_ERR_WORKSPACE, "conv_wgrad_bf16: workspace too small");
  int bc, bn; wgrad_cfg(KS, &bc, &bn);
  dim3 grid((unsigned)splits, (unsigned)(Cr / bc), (unsigned)(Nr / bn));
  a.splits = splits; a.cblk = Cr / bc; a.xcd_tiles = 0;
  if (wgrad_xcd((Cr / bc) * (Nr / bn), a.total_strips)) {
    a.xcd_tiles = (Cr / bc) * (Nr / bn);
    grid = dim3((unsigned)(splits * a.xcd_tiles), 1u, 1u);
  }
  if (KS == 3 && split) {   // one kernel row per workgroup: the (C, N) tile index carries the row (z = nblk * 3 + row).
    // Only the split form: measured 29.2 -> 25.7 ms/step for bf16x3, but 18.7 -> 21.1 for plain bf16, whose nine-tap
    // workgroup already stages two strips per unit and has a third of the MFMA work per staged byte.
    if (a.xcd_tiles > 0) { a.xcd_tiles *= 3; grid = dim3((unsigned)(splits * a.xcd_tiles), 1u, 1u); }
    else grid.z *= 3;
  }
  if (split == 2) {
    if (KS == 3) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<3, 1, 1, HPRI_A_DIRECT, 2, 1>), grid, dim3(256), 0, stream, a);
    else if (bmode == HPRI_A_S2D) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<1, 2, 2, HPRI_A_S2D, 2, 1>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv_wgrad_bf16_kernel<1, 2, 2, HPRI_A_DIRECT, 2, 1>), grid, dim3(256), 0, stream, a);
  } else if (split) {
    if (KS == 3) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<3, 1, 1, HPRI_A_DIRECT, 1, 1>), grid, dim3(256), 0, stream, a);
    else if (bmode == HPRI_A_S2D) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<1, 2, 2, HPRI_A_S2D, 1, 1>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv_wgrad_bf16_kernel<1, 2, 2, HPRI_A_DIRECT, 1, 1>), grid, dim3(256), 0, stream, a);
  } else {
    if (KS == 3) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<3, 1, 1, HPRI_A_DIRECT, 0, 3>), grid, dim3(256), 0, stream, a);
    else if (bmode == HPRI_A_S2D) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<1, 2, 2, HPRI_A_S2D, 0, 1>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv_wgrad_bf16_kernel<1, 2, 2, HPRI_A_DIRECT, 0, 1>), grid, dim3(256), 0, stream, a);
  }
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// Fixed-order sum of the partial slabs written by hpri_conv_wgrad into the parameter-gradient tensor.
extern "C" int hpri_wgrad_reduce(const float* ws, float* dw, int N, int H, int W, int Cin, int Cin_pad, int Cout,
                                 int Cout_pad, int KS, int dst_mode, int Cup, int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(ws && dw && Cin > 0 && Cout > 0, "wgrad_reduce: bad arguments");
  if (dst_mode == 1) HPRI_REQUIRE(Cup > 0 && Cout == 4 * Cup, "wgrad_reduce: convT layout needs Cout == 4*Cup");
  int splits, Cr, Nr;
  hpri_wgrad_plan(N, H, W, Cin_pad, Cout_pad, KS, &splits, &Cr, &Nr);
  dim3 grid((unsigned)hpri_cdiv(Cin, 32), (unsigned)Cout);
  // few slabs x many outputs (up1: 4 x 2 M): the transposing kernel, one thread per four outputs; many slabs x few outputs
  // (up4: 256 x 32 K) keep the slice-parallel generic kernel
  if (KS == 1 && dst_mode == 1 && Cup % 16 == 0 && ((uintptr_t)dw & 15) == 0 && splits <= 16)
    hipLaunchKernelGGL(wgrad_reduce_convt_kernel, dim3((unsigned)hpri_cdiv(Cin, 64), (unsigned)(Cup / 16)), dim3(1024), 0, stream, ws, dw,
                       splits, Cr, Nr, Cin, Cup, accumulate);
  else if (KS == 3 && dst_mode == 0 && Cr % 4 == 0) launch_wgrad_reduce9_wide(ws, dw, splits, Cr, Nr, Cin, Cout, accumulate, stream);
  else if (KS == 3) hipLaunchKernelGGL((wgrad_reduce_kernel<9>), grid, dim3(1024), 0, stream, ws, dw, splits, Cr, Nr, Cin, Cout, dst_mode, Cup, accumulate);
  else if (KS == 1 && dst_mode == 0 && Cr % 4 == 0) launch_wgrad_reduce_wide<1>(ws, dw, splits, Cr, Nr, Cin, Cout, accumulate, stream);
  else hipLaunchKernelGGL((wgrad_reduce_kernel<1>), grid, dim3(1024), 0, stream, ws, dw, splits, Cr, Nr, Cin, Cout, dst_mode, Cup, accumulate);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// The same fixed-order reduction for callers that planned their own pixel splits (hpri_conv_wgrad_bf16v2): slabs
// ws[splits][KS*KS][Nr][Cr].
extern "C" int hpri_wgrad_reduce_ex(const float* ws, float* dw, int splits, int Cr, int Nr, int Cin, int Cout, int KS, int dst_mode,
                                    int Cup, int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(ws && dw && Cin > 0 && Cout > 0 && splits > 0, "wgrad_reduce_ex: bad arguments");
  HPRI_REQUIRE(KS == 1 || KS == 3, "wgrad_reduce_ex: kernel size must be 1 or 3");
  HPRI_REQUIRE(Cin <= Cr && Cout <= Nr, "wgrad_reduce_ex: slab smaller than the gradient");
  if (dst_mode == 1) HPRI_REQUIRE(Cup > 0 && Cout == 4 * Cup, "wgrad_reduce_ex: convT layout needs Cout == 4*Cup");
  dim3 grid((unsigned)hpri_cdiv(Cin, 32), (unsigned)Cout);
  if (KS == 3 && dst_mode == 0 && Cr % 4 == 0) launch_wgrad_reduce9_wide(ws, dw, splits, Cr, Nr, Cin, Cout, accumulate, stream);
  else if (KS == 3) hipLaunchKernelGGL((wgrad_reduce_kernel<9>), grid, dim3(1024), 0, stream, ws, dw, splits, Cr, Nr, Cin, Cout, dst_mode, Cup, accumulate);
  else if (KS == 1 && dst_mode == 0 && Cr % 4 == 0) launch_wgrad_reduce_wide<1>(ws, dw, splits, Cr, Nr, Cin, Cout, accumulate, stream);
  else hipLaunchKernelGGL((wgrad_reduce_kernel<1>), grid, dim3(1024), 0, stream, ws, dw, splits, Cr, Nr, Cin, Cout, dst_mode, Cup, accumulate);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}
