// Weight gradient of the 1x1 layers on bf16 activation PLANES (precision mode "bf16"; the autograd of nn.Linear / Conv2d(k=1):
// reference models.py:105-115,143 -- SpectralUNET's per-pixel MLP):   dW[n][c] = sum_pixels dY[p][n] * X[p][c]
//
// The round-1 kernel these layers used (conv_wgrad_bf16<1>) reads fp32 X and dY, rounds them in registers and ds_writes them:
// 160-180 TFLOP/s on config C3, where after gemm_bf16v3.hip it was 150 of the 290 ms of a step.  Here both operands are the bf16
// planes the forward / BatchNorm-backward kernels have already written, staged by LDS-DMA exactly as conv_wgrad_bf16v2.hip
// stages them ([pixel][64 channels] rows of 128 B, the two 64-byte halves of a row swapped when bit 1 of the pixel index is set;
// ds_read_b64_tr_b16 hands every lane 4 consecutive PIXELS of one channel: conflict-free for any four consecutive rows).
//
//   GEMM       rows n (output features) x columns c (input features), K = pixels; MFMA v_mfma_f32_32x32x16_bf16
//   workgroup  256 threads = 4 waves = 2 (n) x 2 (c); wave tile 128 n x 64 c = 4 x 2 MFMA tiles (128 accumulator VGPRs): six
//              fragment reads per eight MFMAs (a 64 x 64 wave tile needs one per MFMA, which is exactly the LDS bandwidth);
//              workgroup tile 256 n x 128 c, 72 KB of LDS, two workgroups per CU
//   stages     32 pixels x (256 + 128) channels = 24 KB, triple-buffered; one counted vmcnt + one barrier per stage of 16 MFMAs
//              per wave; the six DMA pieces of stage s+2 are issued between the MFMAs of stage s
//   grid       one workgroup per (pixel split, tile); workgroup id mod 8 is the XCD and an XCD owns whole splits, so the
//              tiles of a split run side by side on one XCD and share its X / dY rows through that XCD's L2
//   output     deterministic split-K: slab ws[split][n][c] per workgroup, summed in fixed order by hpri_wgrad_reduce_ex
#include "common.h"

typedef h16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef h16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16x4 __attribute__((address_space(3))) * w1_lds_bf16x4_ptr;

#define W1_STAGE_BYTES (24 * 1024)       // dY: 4 arrays of [32 px][64 ch] (16 KB), X: 2 arrays (8 KB)
#define W1_XB (16 * 1024)

struct Wg1Args {
  const h16_t* xp; int x_cs, x_coff, x_cvalid;       // plane 0 of the layer input; channels >= x_cvalid read as zero
  const h16_t* dyp; int dy_cs, dy_coff, dy_cvalid;   // plane 0 of the gradient w.r.t. the layer output
  float* ws;                                          // [splits][Nr][Cr]
  long long P;                                        // pixels (all images)
  int Cr, Nr, splits, tiles_c, tiles, stages_per_split, total_stages;
  // MODE 1 (ConvTranspose2d(2,2): dW[ci][co][tap] = sum_p X[p][ci] * dY[up(p, tap)][co]): X rows are the N*H*W low-resolution pixels,
  // row n = tap * cup + co of the slab gathers dY at pixel (py0 + 2y + tap/2, px0 + 2x + tap%2) of the [N, H2, W2] gradient planes
  int HW, W, H2, W2, py0, px0, cup;
};

// One fragment = two transposed reads (pixel rows L and L + 4).  Written as inline assembly: through the builtin
// (__builtin_amdgcn_ds_read_tr16_b64_*) hipcc sees an LDS read it cannot tell apart from the LDS-DMA writes in flight and puts
// "s_waitcnt vmcnt(0)" in front of every group of reads -- each DMA piece issued between the MFMAs was then waited for at once
// (a full memory round trip per piece, no load ever in flight under the MFMAs).  The counted vmcnt + barrier at the top of a stage
// is what orders DMA writes and reads here; the reads' own completion is waited for by W1_WAIT_FRAGS below.
template <int OFF>
__device__ __forceinline__ bf16x8 w1_tr_frag(unsigned lds_addr) {
  bf16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(lo), "=&v"(hi) : "v"(lds_addr), "n"(OFF), "n"(OFF + 512));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// all six fragments of a k16-step have arrived (the operands tie the wait to the registers it guards)
#define W1_WAIT_FRAGS(a_, b_)                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_[0]), "+v"(a_[1]), "+v"(a_[2]), "+v"(a_[3]), "+v"(b_[0]), "+v"(b_[1]))

template <int MODE>
__global__ __launch_bounds__(256, 2) void wgrad1x1_bf16v3_kernel(Wg1Args a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * W1_STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wc = wave & 1;

  const int per_xcd = gridDim.x >> 3;
  const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (item >= a.splits * a.tiles) return;
  const int split_id = item / a.tiles, tile = item - split_id * a.tiles;
  const int nb = tile / a.tiles_c, cb = tile - nb * a.tiles_c;
  const int n_blk = nb * 256, c_blk = cb * 128;
  const int s_begin = split_id * a.stages_per_split;
  const int s_end = min(a.total_stages, s_begin + a.stages_per_split);

  // ---- DMA roles: piece i = wave + 4 q (q = 0..5) of a stage; pieces 0..15 are the four dY arrays (4 pieces = 32 pixel rows each),
  //      16..23 the two X arrays.  Lane -> pixel row 8 (i & 3) + (lane >> 3), physical 16-byte slot lane & 7, which holds logical
  //      slot (lane & 7) ^ 4 * ((row >> 1) & 1) of that pixel's 64 channels ----
  unsigned off[6]; unsigned okbits = 0u;
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int i = wave + 4 * q;
    const bool isx = i >= 16;
    const int arr = isx ? (i - 16) >> 2 : i >> 2;
    const int r = 8 * (i & 3) + (lane >> 3);
    const int ls = (lane & 7) ^ (((r >> 1) & 1) << 2);
    const int ch = arr * 64 + ls * 8;
    const bool ok = isx ? (c_blk + ch < a.x_cvalid) : (n_blk + ch < a.dy_cvalid);
    off[q] = (MODE == 1 && !isx) ? (unsigned)((ch - arr * 64) * 2)         // channel inside the array; the row comes per stage
                                 : (unsigned)((isx ? r * a.x_cs : r * a.dy_cs) + ch) * 2u;
    okbits |= ok ? (1u << q) : 0u;                   // channels beyond the valid width: zero-filled (out-of-range offset)
  }
  const h16_t* xbase = a.xp + a.x_coff + c_blk;
  const h16_t* ybase = MODE == 1 ? a.dyp + a.dy_coff : a.dyp + a.dy_coff + n_blk;
  // MODE 1: the four dY arrays of a workgroup are 64-channel runs of (possibly different) taps; a wave's four dY pieces (q = 0..3)
  // are the SAME 8 pixel rows of the four arrays, so the gather position is computed once per stage and lane
  int tapoff[4] = {0, 0, 0, 0};                      // elements to add to the gathered row's offset for array q
  if (MODE == 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n0 = min(n_blk + q * 64, 4 * a.cup - 64);      // (arrays beyond the valid rows are masked by okbits)
      const int tap = n0 / a.cup, co0 = n0 - tap * a.cup;
      tapoff[q] = ((tap >> 1) * a.W2 + (tap & 1)) * a.dy_cs + co0;
    }
  }
  // (descriptors rebuilt per stage from the stage's first pixel row: wave-uniform, 64-bit, whatever the tensor size)
#define W1_ISSUE_PREP(s_)                                                                                              \
  const long long p0_ = (long long)(s_) * 32;                                                                          \
  const unsigned long long px_ = (unsigned long long)(uintptr_t)(xbase + p0_ * a.x_cs);                                \
  const unsigned long long py_ = (unsigned long long)(uintptr_t)(MODE == 1 ? ybase : ybase + p0_ * a.dy_cs);           \
  /* (readfirstlane returns a SIGNED int: without the casts the low word is sign-extended over the high one) */       \
  const unsigned pxl_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)px_), pxh_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(px_ >> 32)); \
  const unsigned pyl_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)py_), pyh_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(py_ >> 32)); \
  const hpri_rsrc_t rx_ = HPRI_MAKE_RSRC((((unsigned long long)pxh_ << 32) | pxl_), 0x7FFFFF00);                       \
  const hpri_rsrc_t ry_ = HPRI_MAKE_RSRC((((unsigned long long)pyh_ << 32) | pyl_), 0x7FFFFF00);                       \
  const long long left_ = a.P - p0_;                 /* pixel rows of this stage that exist */                         \
  unsigned grow_ = 0u;                               /* MODE 1: byte offset of this lane's gathered dY row (parity 0) */ \
  if (MODE == 1) {                                                                                                     \
    const long long pp_ = p0_ + 8 * wave + (lane >> 3);                                                                \
    const int pc_ = (int)(pp_ < a.P ? pp_ : a.P - 1);                                                                  \
    const int img_ = pc_ / a.HW, rem_ = pc_ - img_ * a.HW, yy_ = rem_ / a.W, xx_ = rem_ - yy_ * a.W;                   \
    grow_ = (unsigned)(((img_ * a.H2 + a.py0 + 2 * yy_) * a.W2 + a.px0 + 2 * xx_) * a.dy_cs) * 2u;                     \
  }                                                                                                                    \
  (void)rx_; (void)ry_; (void)left_; (void)pxl_; (void)pxh_; (void)pyl_; (void)pyh_; (void)grow_;
#define W1_ISSUE(q, bo_)                                                                                               \
  {                                                                                                                    \
    const int i_ = wave + 4 * (q);                                                                                     \
    const bool in_ = ((okbits >> (q)) & 1u) && (long long)(8 * (i_ & 3) + (lane >> 3)) < left_;                        \
    if (i_ >= 16) { HPRI_LDS_DMA16(rx_, smem + (bo_) + i_ * 1024, in_ ? off[q] : HPRI_DMA_OOB, 0); }                   \
    else if (MODE == 1) { HPRI_LDS_DMA16(ry_, smem + (bo_) + i_ * 1024, in_ ? grow_ + off[q] + (unsigned)(tapoff[(q) & 3] * 2) : HPRI_DMA_OOB, 0); } \
    else          { HPRI_LDS_DMA16(ry_, smem + (bo_) + i_ * 1024, in_ ? off[q] : HPRI_DMA_OOB, 0); }                   \
  }

  // ---- transposed-read lane roles (conv_wgrad_bf16v2.hip): 16-lane group = (k half lh, channel half lg); lane i of the group
  //      addresses pixel row lq = i >> 2, channel quad lp = i & 3 and receives channel i of its 16 ----
  const int lg = (lane >> 4) & 1, lh = lane >> 5, lq = (lane >> 2) & 3, lp = lane & 3;
  const int li = lane & 31;
  const int L = 8 * lh + lq;
  const int eL = (L >> 1) & 1;
  // fragment h (32 channels) of a [32 px][64 ch] array at k16-step kk: array + (16 kk + L) * 128 + ((h ^ eL) << 6) + lg * 32 + lp * 8
  const int f0 = L * 128 + ((0 ^ eL) << 6) + lg * 32 + lp * 8;
  const int f1 = L * 128 + ((1 ^ eL) << 6) + lg * 32 + lp * 8;

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;      // LDS byte address of the staging area
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (s_begin < s_end) {
    { W1_ISSUE_PREP(s_begin) W1_ISSUE(0, 0) W1_ISSUE(1, 0) W1_ISSUE(2, 0) W1_ISSUE(3, 0) W1_ISSUE(4, 0) W1_ISSUE(5, 0) }
    if (s_begin + 1 < s_end) {
      W1_ISSUE_PREP(s_begin + 1)
      W1_ISSUE(0, W1_STAGE_BYTES) W1_ISSUE(1, W1_STAGE_BYTES) W1_ISSUE(2, W1_STAGE_BYTES) W1_ISSUE(3, W1_STAGE_BYTES)
      W1_ISSUE(4, W1_STAGE_BYTES) W1_ISSUE(5, W1_STAGE_BYTES)
    }
  }
  int bo = 0;
  for (int s = s_begin; s < s_end; ++s) {
    // the wave's own six pieces of stage s have landed when at most the six of stage s+1 are in flight; behind the barrier every
    // wave has left stage s-1, whose buffer takes stage s+2.  ONE straight-line MFMA sequence per stage (see gemm_bf16v3.hip).
    if (s + 1 < s_end) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const bool more = s + 2 < s_end;
    const int bo2 = bo == 0 ? 2 * W1_STAGE_BYTES : bo - W1_STAGE_BYTES;
    W1_ISSUE_PREP(more ? s + 2 : s)
    const unsigned yb0 = lds0 + bo + wn * (2 * 4096) + f0, yb1 = lds0 + bo + wn * (2 * 4096) + f1;     // this wave's two dY arrays (128 n)
    const unsigned xb0 = lds0 + bo + W1_XB + wc * 4096 + f0, xb1 = lds0 + bo + W1_XB + wc * 4096 + f1; // this wave's X array (64 c)
    bf16x8 af0[4], bf0[2], af1[4], bf1[2];
#define W1_READ(a_, b_, KO_)                                                                                          \
  a_[0] = w1_tr_frag<(KO_)>(yb0); a_[1] = w1_tr_frag<(KO_)>(yb1);                                                     \
  a_[2] = w1_tr_frag<4096 + (KO_)>(yb0); a_[3] = w1_tr_frag<4096 + (KO_)>(yb1);                                       \
  b_[0] = w1_tr_frag<(KO_)>(xb0); b_[1] = w1_tr_frag<(KO_)>(xb1);
#define W1_MFMAS(a_, b_, M0_)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                       \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                 \
    acc[i][j] = HPRI_MFMA_32X32X16(a_[i], b_[j], acc[i][j], 0, 0, 0);                            \
    if (more) {                                                                                                       \
      const int m = (M0_) + i * 2 + j;                                                                                \
      if (m == 1) W1_ISSUE(0, bo2)                                                                                    \
      if (m == 3) W1_ISSUE(1, bo2)                                                                                    \
      if (m == 5) W1_ISSUE(2, bo2)                                                                                    \
      if (m == 7) W1_ISSUE(3, bo2)                                                                                    \
      if (m == 9) W1_ISSUE(4, bo2)                                                                                    \
      if (m == 11) W1_ISSUE(5, bo2)                                                                                   \
    }                                                                                                                 \
  }
    W1_READ(af0, bf0, 0)
    W1_WAIT_FRAGS(af0, bf0);
    W1_READ(af1, bf1, 2048)                     // the second k16-step's fragments arrive under the first one's MFMAs
    W1_MFMAS(af0, bf0, 0)
    W1_WAIT_FRAGS(af1, bf1);
    W1_MFMAS(af1, bf1, 8)
#undef W1_READ
#undef W1_MFMAS
    bo = bo == 2 * W1_STAGE_BYTES ? 0 : bo + W1_STAGE_BYTES;
  }
#undef W1_ISSUE
#undef W1_ISSUE_PREP

  // slab ws[split][n][c]: MFMA rows = n (A operand = dY), columns = c (B operand = X); 32 lanes = 128 contiguous bytes
  float* slab = a.ws + (size_t)split_id * a.Nr * a.Cr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c_blk + wc * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n_blk + wn * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        slab[(size_t)n * a.Cr + c] = acc[i][j][r];
      }
    }
}

static void wg1_geometry(long long P, int Cin_pad, int Cout_pad, Wg1Args* a) {
  a->Cr = hpri_cdiv(Cin_pad, 128) * 128; a->Nr = hpri_cdiv(Cout_pad, 256) * 256;
  a->tiles_c = a->Cr / 128; a->tiles = a->tiles_c * (a->Nr / 256);
  a->total_stages = (int)((P + 31) / 32);
  // two workgroups per CU: about three rounds of 2 x CUs (split, tile) items, a multiple of 8 splits (one XCD owns whole splits),
  // at least 64 stages (2048 pixels) per item
  const int slots = 2 * hpri_cu_count();
  int s = hpri_cdiv(3 * slots, a->tiles);
  s = hpri_cdiv(s, 8) * 8;
  while (s > 8 && a->total_stages / s < 64) s -= 8;
  if (s == 8 && a->total_stages / 8 < 16) s = 1;       // a small problem: one item per tile
  a->stages_per_split = hpri_cdiv(a->total_stages, s);
  a->splits = hpri_cdiv(a->total_stages, a->stages_per_split);
}

// Workspace of hpri_wgrad1x1_bf16v3: splits * Nr * Cr floats.
extern "C" int hpri_wgrad1x1_bf16v3_plan(long long P, int Cin_pad, int Cout_pad, int* splits, int* Cr, int* Nr) {
  HPRI_REQUIRE(P > 0 && Cin_pad > 0 && Cout_pad > 0 && splits && Cr && Nr, "wgrad1x1_bf16v3_plan: bad arguments");
  Wg1Args a;
  wg1_geometry(P, Cin_pad, Cout_pad, &a);
  *splits = a.splits; *Cr = a.Cr; *Nr = a.Nr;
  return HPRI_OK;
}

// Partial weight-gradient slabs of a 1x1 layer from bf16 planes (plane 0 of X and of dY: P pixel rows, channel strides / offsets /
// valid widths multiples of 8 elements, 16-byte aligned); finish with hpri_wgrad_reduce_ex(ws, dw, splits, Cr, Nr, Cin, Cout, 1, 0, 0,
// accumulate).
extern "C" int hpri_wgrad1x1_bf16v3(const void* x_planes, int x_cs, int x_coff, int x_cvalid, const void* dy_planes, int dy_cs,
                                    int dy_coff, int dy_cvalid, float* ws, size_t ws_floats, long long P, int Cin_pad, int Cout_pad,
                                    hipStream_t stream) {
  HPRI_REQUIRE(x_planes && dy_planes && ws, "wgrad1x1_bf16v3: null pointer");
  HPRI_REQUIRE(P > 0 && P < (1ll << 36) && Cin_pad > 0 && Cout_pad > 0, "wgrad1x1_bf16v3: empty problem");
  HPRI_REQUIRE(x_cs % 8 == 0 && x_coff % 8 == 0 && x_cvalid % 8 == 0 && dy_cs % 8 == 0 && dy_coff % 8 == 0 && dy_cvalid % 8 == 0,
               "wgrad1x1_bf16v3: channel strides / offsets / valid widths must be multiples of 8 (16-byte DMA granules)");
  HPRI_REQUIRE(x_coff + x_cvalid <= x_cs && dy_coff + dy_cvalid <= dy_cs, "wgrad1x1_bf16v3: valid channels exceed the channel stride");
  HPRI_REQUIRE(x_cs <= 16384 && dy_cs <= 16384, "wgrad1x1_bf16v3: channel stride too large");
  HPRI_REQUIRE(((uintptr_t)x_planes & 15) == 0 && ((uintptr_t)dy_planes & 15) == 0, "wgrad1x1_bf16v3: planes must be 16-byte aligned");
  Wg1Args a;
  wg1_geometry(P, Cin_pad, Cout_pad, &a);
  if ((size_t)a.splits * a.Cr * a.Nr > ws_floats) return hpri_set_error(HPRI_ERR_WORKSPACE, "wgrad1x1_bf16v3: workspace too small");
  a.xp = reinterpret_cast<const h16_t*>(x_planes); a.x_cs = x_cs; a.x_coff = x_coff; a.x_cvalid = x_cvalid;
  a.dyp = reinterpret_cast<const h16_t*>(dy_planes); a.dy_cs = dy_cs; a.dy_coff = dy_coff; a.dy_cvalid = dy_cvalid;
  a.ws = ws; a.P = P;
  a.HW = a.W = a.H2 = a.W2 = a.py0 = a.px0 = a.cup = 0;
  const long long items = (long long)a.splits * a.tiles;
  HPRI_REQUIRE(items < (1ll << 24), "wgrad1x1_bf16v3: too many work items");
  dim3 grid((unsigned)(hpri_cdiv((int)items, 8) * 8), 1u, 1u);
  hipLaunchKernelGGL(wgrad1x1_bf16v3_kernel<0>, grid, dim3(256), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// ConvTranspose2d(k = 2, s = 2) weight gradient from bf16 planes: x planes [N, H, W] (Cin channels), dy planes [N, H2, W2] (Cup channels
// from dy_coff on; Cup a multiple of 64); slab rows n = tap * Cup + co, columns ci; sizes from hpri_wgrad1x1_bf16v3_plan(N*H*W, Cin_pad,
// 4*Cup); finish with hpri_wgrad_reduce_ex(ws, dw, splits, Cr, Nr, Cin, 4*Cup, 1, 1, Cup, accumulate).
extern "C" int hpri_wgrad_convt_bf16v3(const void* x_planes, int x_cs, int x_coff, int x_cvalid, const void* dy_planes, int dy_cs, int dy_coff,
                                       float* ws, size_t ws_floats, int N, int H, int W, int Cin_pad, int Cup, int H2, int W2, int py0,
                                       int px0, hipStream_t stream) {
  HPRI_REQUIRE(x_planes && dy_planes && ws, "wgrad_convt_bf16v3: null pointer");
  HPRI_REQUIRE(N > 0 && H > 0 && W > 0 && Cin_pad > 0 && Cup > 0 && Cup % 64 == 0, "wgrad_convt_bf16v3: Cup must be a positive multiple of 64");
  HPRI_REQUIRE(x_cs % 8 == 0 && x_coff % 8 == 0 && x_cvalid % 8 == 0 && dy_cs % 8 == 0 && dy_coff % 8 == 0,
               "wgrad_convt_bf16v3: channel strides / offsets / valid widths must be multiples of 8 (16-byte DMA granules)");
  HPRI_REQUIRE(x_coff + x_cvalid <= x_cs && dy_coff + Cup <= dy_cs && x_cs <= 16384, "wgrad_convt_bf16v3: valid channels exceed the channel stride");
  HPRI_REQUIRE(((uintptr_t)x_planes & 15) == 0 && ((uintptr_t)dy_planes & 15) == 0, "wgrad_convt_bf16v3: planes must be 16-byte aligned");
  HPRI_REQUIRE(py0 >= 0 && px0 >= 0 && py0 + 2 * H <= H2 && px0 + 2 * W <= W2, "wgrad_convt_bf16v3: geometry out of range");
  HPRI_REQUIRE((long long)N * H2 * W2 * dy_cs * 2 < 0x7FFFFF00ll && (long long)N * H * W < (1ll << 31), "wgrad_convt_bf16v3: gradient planes exceed 2 GiB (32-bit DMA offsets)");
  Wg1Args a;
  const long long P = (long long)N * H * W;
  wg1_geometry(P, Cin_pad, 4 * Cup, &a);
  if ((size_t)a.splits * a.Cr * a.Nr > ws_floats) return hpri_set_error(HPRI_ERR_WORKSPACE, "wgrad_convt_bf16v3: workspace too small");
  a.xp = reinterpret_cast<const h16_t*>(x_planes); a.x_cs = x_cs; a.x_coff = x_coff; a.x_cvalid = x_cvalid;
  a.dyp = reinterpret_cast<const h16_t*>(dy_planes); a.dy_cs = dy_cs; a.dy_coff = dy_coff; a.dy_cvalid = 4 * Cup;
  a.ws = ws; a.P = P;
  a.HW = H * W; a.W = W; a.H2 = H2; a.W2 = W2; a.py0 = py0; a.px0 = px0; a.cup = Cup;
  const long long items = (long long)a.splits * a.tiles;
  HPRI_REQUIRE(items < (1ll << 24), "wgrad_convt_bf16v3: too many work items");
  dim3 grid((unsigned)(hpri_cdiv((int)items, 8) * 8), 1u, 1u);
  hipLaunchKernelGGL(wgrad1x1_bf16v3_kernel<1>, grid, dim3(256), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}
