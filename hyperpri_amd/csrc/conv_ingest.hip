// First-layer fused ingest (predict path of the 16-bit modes; reference models.py:169,215-216: the Conv3d(1, F, (D,3,3)) of CubeNET reads
// the caller's (N,1,D,H,W) fp32 cube, i.e. a 3x3 convolution over D channels of an NCHW tensor): the convolution stages the
// caller's cube ITSELF -- fp32 NC(D)HW -> 16-bit [pixel][channel] tiles in LDS -- instead of reading planes that a separate
// layout pass (nchw_to_nhwc_v4: 1.1 GB read + 0.6 GB written per batch-2 step of the benched shape) wrote first.  Eval-mode
// BatchNorm folded into the weights / bias by the caller (hpri_pack_weight_bf16_scaled, hpri_bn_fold), ReLU in the epilogue, result
// as 16-bit rows = the next convolution's planes.  No weight gradient reads the input here, which is why this is the predict path's.
//
//   workgroup  256 threads = 4 waves, 8 x 32 pixels x 64 output channels, 67 KB of LDS: two workgroups per CU; persistent over a
//              fixed item list, workgroup id mod 8 = XCD, each XCD walks one band of the image in raster order (conv_bf16v3.hip).
//   MFMA       v_mfma_f32_16x16x32 (bf16 / f16 by build), wave tile 64 px x 64 ch, weights = A operand, pixels = B operand, the
//              accumulation order of conv_bf16v3 (chunk, kernel row, tap): results are bit-identical to layout pass + conv_bf16v3.
//   input      per 32-channel chunk the halo (10 x 34 pixels) is loaded as fp32 dwords, wave w = channels 8 w .. 8 w + 7 of the chunk
//              (the channel is wave-uniform: it sits in the buffer instruction's scalar offset), lane = halo pixel (consecutive
//              lanes = consecutive columns of one channel plane: coalesced rows of 136 bytes); 48 loads per lane stay in flight over
//              two stages of MFMAs, are rounded to 16 bits and written as one ds_write_b128 per pixel and 8-channel slot into the
//              other halo buffer (same XOR swizzle as conv_bf16v3: conflict-free fragment reads at every tap).  Pixels outside the
//              image: offset beyond the descriptor's range, the hardware returns zeros.  Channels beyond C: clamped to C - 1 (their
//              packed weights are zero).
//   weights    one kernel row (3 taps x 64 ch x 32 k = 12 KB) per stage by LDS-DMA from the packed weights, double-buffered.
#include "common.h"

typedef h16_t igx8 __attribute__((ext_vector_type(8)));

// tiles of 256 pixels: 8 x 32, and for the columns left over by W mod 32 one column of 16 x 16 or 32 x 8 tiles (halo <= 340 pixels)
#define IG_HP_MAX 340
#define IG_A_BYTES (IG_HP_MAX * 64)           // 21760
#define IG_B_BYTES (12 * 1024)

struct IngestArgs {
  const float* x;                             // [N][C][H][W] fp32
  const h16_t* wp;                            // packed weights [chunk][tap][Cout_pad][32]
  const float* bias;
  h16_t* y; int y_cs, y_coff;                 // 16-bit rows: y + pixel * y_cs + y_coff + channel
  int N, C, H, W, Cout, Cout_pad, relu;
  int nseg, seg_twl[2], seg_xbeg[2], seg_ntx[2], seg_first[2];      // column bands of one tile width each (tile width 1 << twl)
  int tiles_img, ntiles, nb_count, per_xcd;
};

__global__ __launch_bounds__(256, 2) void conv_ingest_kernel(IngestArgs a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * IG_B_BYTES + 2 * IG_A_BYTES + 512];
  unsigned char* b_lds = smem;
  unsigned char* a_lds = smem + 2 * IG_B_BYTES;
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * IG_B_BYTES + 2 * IG_A_BYTES);      // [2 slots][64]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int xcd = blockIdx.x & 7, nloc = (int)(gridDim.x >> 3);
  const int items_all = a.ntiles * a.nb_count;
  const int nchunks = (a.C + 31) >> 5;
  const int S = nchunks * 3;
  const int HWs = a.H * a.W;
  const int tap_bytes = a.Cout_pad * 64;
  const hpri_rsrc_t rs_b = HPRI_MAKE_RSRC(a.wp, 0x7FFFFF00);
  (void)tap_bytes; (void)rs_b; (void)HWs; (void)S;

  // fragment addresses (conv_bf16v3.hip): pixels = B operand, lane (li, lq) reads k-slot lq of halo pixel hp = hp00 + rows / columns of
  // the M-tile and the tap; the slot swizzle depends on hp, so the address is formed per read (4 vector instructions beside 4 MFMAs)
  // from ONE register -- a table of the 36 addresses does not fit beside the staged input
  const int lq16 = lq << 4;
  const int bofs = li * 64 + ((lq ^ (((li >> 2) & 1) << 1)) << 4);

#define IG_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define IG_BARRIER()                                 \
  __builtin_amdgcn_sched_barrier(0);                 \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_s_barrier();                      \
  __builtin_amdgcn_sched_barrier(0)
#define IG_DMA_B(buf_, s_, q_, go_) \
  HPRI_LDS_DMA16(rs_b, b_lds + (buf_) * IG_B_BYTES + ((q_) * 4 + wave) * 1024, go_, ((s_) * 3 + (q_)) * tap_bytes)
// one third of a chunk's halo: rounds 2 q_ and 2 q_ + 1 (halo pixel round * 64 + lane) x the wave's 8 channels = 16 dwords per lane
// (vo0_ / vo1_: the two pixel offsets, or HPRI_DMA_OOB when the chunk does not exist -- the loads are issued UNCONDITIONALLY, an
// out-of-range one moves nothing: hipcc's own vmcnt bookkeeping takes the minimum over all paths, and a conditional load made it wait
// for everything in flight before every use of a staged register)
#define IG_LOAD_THIRD(st_, c_, vo0_, vo1_)                                                                            \
  _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                     \
    const int ch = min((c_) * 32 + wave * 8 + j, a.C - 1);                                                            \
    const int so = ch * HWs * 4;                                                                                      \
    (void)so;                                                                                                         \
    st_[0][j] = HPRI_BUFFER_LOAD_F32(rs_x, vo0_, so);                                                                 \
    st_[1][j] = HPRI_BUFFER_LOAD_F32(rs_x, vo1_, so);                                                                 \
  }
#define IG_WRITE_THIRD(st_, buf_, q_)                                                                                 \
  _Pragma("unroll") for (int rr = 0; rr < 2; ++rr) {                                                                  \
    const int hp = (2 * (q_) + rr) * 64 + lane;                                                                       \
    igx8 h;                                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) h[j] = (h16_t)st_[rr][j];                                           \
    if (hp < HP)                                                                                                      \
      *reinterpret_cast<igx8*>(a_lds + (buf_) * IG_A_BYTES + hp * 64 + ((wave ^ (((hp >> 2) & 1) << 1)) << 4)) = h;   \
  }
#define IG_READ_TAP(fa_, fb_, ab_, bb_, dy_, dx_)                                                                     \
  _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                                  \
    const int hp_ = hpo + (umt[mt] + (dy_) * HWd + (dx_));                                                            \
    fa_[mt] = *reinterpret_cast<const igx8*>((ab_) + (hp_ << 6) + (lq16 ^ ((hp_ & 4) << 3)));                         \
  }                                                                                                                   \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                                    \
      fb_[nt] = *reinterpret_cast<const igx8*>((bb_) + (dx_) * 4096 + nt * 1024);
#define IG_MFMA_TAP(fa_, fb_)                                                                                         \
  _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                                    \
      _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                                \
          acc[mt][nt] = HPRI_MFMA_16X16X32(fb_[nt], fa_[mt], acc[mt][nt], 0, 0, 0);
// One stage s = 3 c + dy = one kernel row of one chunk.  It issues, in this order, the three weight pieces of stage s + 1 and ONE THIRD
// of a later chunk's halo (third (s + 1) mod 3 of chunk (s + 1) / 3 + 1: 16 dwords per lane into the staging registers of its parity),
// and at its end rounds the third that the PREVIOUS stage issued into the other halo buffer -- every third has two stages of MFMAs to
// land, two thirds (32 registers) are in flight at any time, and chunk c + 1 is complete in LDS when stage (c, 2) ends.  The other halo
// buffer was last read in chunk c - 1; its first write is at the end of (c, 0), behind that stage's barrier.  Beyond the last chunk /
// stage the same instructions run with out-of-range offsets (nothing moves; the halo writes land in a buffer nobody reads again before
// the next item's barrier).  vmcnt counts in issue order: at the top of a stage the third issued one stage ago may stay in flight
// (16), the weights issued before it may not.
#define IG_STAGE(par_, dy_)                                                                                           \
  {                                                                                                                   \
    const int s_ = c * 3 + (dy_);                                                                                     \
    constexpr int bb_i = ((par_) * 3 + (dy_)) & 1;                                                                    \
    constexpr int sp_ = ((par_) + (dy_)) & 1;              /* parity of s: the staging registers this stage fills */ \
    constexpr int q_ = ((dy_) + 1) % 3;                                                                               \
    const unsigned char* ab_ = a_lds + (par_) * IG_A_BYTES;                                                           \
    const unsigned char* bb_ = b_lds + bb_i * IG_B_BYTES + bofs;                                                      \
    const bool ld_ = (dy_) == 2 ? more_c2 : more_c;                                                                   \
    const unsigned vo0_ = ld_ ? voff[2 * q_] : HPRI_DMA_OOB, vo1_ = ld_ ? voff[2 * q_ + 1] : HPRI_DMA_OOB;            \
    const unsigned go_ = s_ + 1 < S ? goff : HPRI_DMA_OOB;                                                            \
    (void)vo0_; (void)vo1_; (void)go_;                                                                                \
    IG_WAIT_VM(16);                                                                                                   \
    IG_BARRIER();                                                                                                     \
    IG_DMA_B(bb_i ^ 1, s_ + 1, 0, go_); IG_DMA_B(bb_i ^ 1, s_ + 1, 1, go_); IG_DMA_B(bb_i ^ 1, s_ + 1, 2, go_);       \
    if (sp_) { IG_LOAD_THIRD(st1, c + 1 + ((dy_) == 2), vo0_, vo1_) } else { IG_LOAD_THIRD(st0, c + 1 + ((dy_) == 2), vo0_, vo1_) } \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    igx8 fa0[4], fb0[4], fa1[4], fb1[4];                                                                              \
    int hpo = hp00;                                                                                                   \
    asm volatile("" : "+v"(hpo));                                                                                     \
    IG_READ_TAP(fa0, fb0, ab_, bb_, dy_, 0)                                                                           \
    IG_READ_TAP(fa1, fb1, ab_, bb_, dy_, 1)                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                                    \
    IG_MFMA_TAP(fa0, fb0)                                                                                             \
    IG_READ_TAP(fa0, fb0, ab_, bb_, dy_, 2)                                                                           \
    IG_MFMA_TAP(fa1, fb1)                                                                                             \
    IG_MFMA_TAP(fa0, fb0)                                                                                             \
    __builtin_amdgcn_s_setprio(0);                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    if (sp_) { IG_WRITE_THIRD(st0, (par_) ^ 1, dy_) } else { IG_WRITE_THIRD(st1, (par_) ^ 1, dy_) }                   \
  }

  int slot = 0;
  for (int k = (int)(blockIdx.x >> 3); k < a.per_xcd; k += nloc, slot ^= 1) {
    const int item = xcd * a.per_xcd + k;
    if (item >= items_all) break;
    const int bx = item / a.nb_count, nb = item - bx * a.nb_count;
    const int img = bx / a.tiles_img, tin = bx - img * a.tiles_img;
    const int seg = (a.nseg > 1 && tin >= a.seg_first[1]) ? 1 : 0;
    const int twl = a.seg_twl[seg], TW = 1 << twl, TH = 256 >> twl, HWd = TW + 2, HP = (TH + 2) * HWd;
    const int tt = tin - a.seg_first[seg];
    const int ty = tt / a.seg_ntx[seg], tx = tt - ty * a.seg_ntx[seg];
    const int y0 = ty * TH, x0 = a.seg_xbeg[seg] + tx * TW;
    const int xlim = min(a.W, a.seg_xbeg[seg] + a.seg_ntx[seg] * TW);
    const unsigned hw_inv = (65536u + (unsigned)HWd - 1u) / (unsigned)HWd;         // exact quotient for hp < 2048
    // fragment addressing: pixel p = (wave*4 + mt)*16 + li of the tile sits at halo pixel (p >> twl) * HWd + (p & (TW - 1)) for tap (0, 0):
    // a per-lane part (the same for every M-tile) plus a wave-uniform part per M-tile
    const int hp00 = twl == 3 ? (li >> 3) * HWd + (li & 7) : li;
    int umt[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int pb = (wave * 4 + mt) * 16;
      umt[mt] = (pb >> twl) * HWd + (pb & (TW - 1));
    }
    // per-lane byte offsets of the six halo pixels this lane loads (the same for every channel)
    unsigned voff[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int hp = r * 64 + lane;
      unsigned off = HPRI_DMA_OOB;
      if (hp < HP) {
        const int hy = (int)(((unsigned)hp * hw_inv) >> 16), hx = hp - hy * HWd;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) off = (unsigned)(iy * a.W + ix) * 4u;
      }
      voff[r] = off;
    }
    const unsigned long long pb = (unsigned long long)(uintptr_t)(a.x + (size_t)img * a.C * HWs);
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pb), phi = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32));
    const unsigned xbytes = __builtin_amdgcn_readfirstlane((unsigned)a.C * (unsigned)HWs * 4u);
    const hpri_rsrc_t rs_x = HPRI_MAKE_RSRC((((unsigned long long)phi << 32) | plo), xbytes);
    (void)plo; (void)phi; (void)xbytes; (void)rs_x;
    const int n = wave * 16 + (lane >> 2);
    const unsigned goff = (unsigned)((nb * 64 + n) * 32 + (((lane & 3) ^ (((n >> 2) & 1) << 1)) << 3)) * 2u;
    (void)goff;

    // every wave has left the previous item's main loop (and its last, empty, transfers have landed) before its buffers are written again
    IG_WAIT_VM(0);
    IG_BARRIER();
    if (tid < 64) {
      const int co = nb * 64 + tid;
      bias_lds[slot * 64 + tid] = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
    }
    float st0[2][8], st1[2][8];
    // chunk 0 whole (three thirds through the two staging sets), then the first weight stage and the first third of chunk 1
    IG_LOAD_THIRD(st0, 0, voff[0], voff[1])
    IG_LOAD_THIRD(st1, 0, voff[2], voff[3])
    IG_WRITE_THIRD(st0, 0, 0)
    IG_LOAD_THIRD(st0, 0, voff[4], voff[5])
    IG_WRITE_THIRD(st1, 0, 1)
    IG_WRITE_THIRD(st0, 0, 2)
    __builtin_amdgcn_sched_barrier(0);
    IG_DMA_B(0, 0, 0, goff); IG_DMA_B(0, 0, 1, goff); IG_DMA_B(0, 0, 2, goff);
    {                                                   // ("stage -1", odd parity: the first third of chunk 1)
      const unsigned vo0_ = nchunks > 1 ? voff[0] : HPRI_DMA_OOB, vo1_ = nchunks > 1 ? voff[1] : HPRI_DMA_OOB;
      (void)vo0_; (void)vo1_;
      IG_LOAD_THIRD(st1, 1, vo0_, vo1_)
    }
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < nchunks; ++c) {
      {
        const bool more_c = c + 1 < nchunks, more_c2 = c + 2 < nchunks;
        IG_STAGE(0, 0)
        IG_STAGE(0, 1)
        IG_STAGE(0, 2)
        if (!more_c) break;
      }
      ++c;
      {
        const bool more_c = c + 1 < nchunks, more_c2 = c + 2 < nchunks;
        IG_STAGE(1, 0)
        IG_STAGE(1, 1)
        IG_STAGE(1, 2)
      }
    }

    // ---- epilogue: acc[mt][nt][r] = pixel (wave*4 + mt)*16 + li of the tile, channel nb*64 + nt*16 + 4*lq + r ----
    f32x4 b4[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) b4[nt] = *reinterpret_cast<const f32x4*>(bias_lds + slot * 64 + nt * 16 + 4 * lq);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int p = (wave * 4 + mt) * 16 + li;
      const int iy = y0 + (p >> twl), ix = x0 + (p & (TW - 1));
      const bool ok = iy < a.H && ix < xlim;
      h16_t* row = a.y + ((size_t)(img * a.H + min(iy, a.H - 1)) * a.W + min(ix, a.W - 1)) * a.y_cs + a.y_coff + nb * 64 + 4 * lq;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        f32x4 v = acc[mt][nt] + b4[nt];
        if (a.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        bf16x4_t h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (h16_t)v[r];
        if (ok && nb * 64 + nt * 16 + 4 * lq + 4 <= a.Cout) *reinterpret_cast<bf16x4_t*>(row + nt * 16) = h;
      }
    }
  }
}

// 3x3 pad-1 convolution of the caller's NC(D)HW fp32 cube (N, C, H, W contiguous) with weights packed by
// hpri_pack_weight_bf16(_scaled) ([chunk][tap][Cout_pad][32]; channels beyond C zero), + bias, optional ReLU, result as 16-bit rows
// y + pixel * y_cs + y_coff + channel (Cout a multiple of 4; y 8-byte aligned, y_cs and y_coff multiples of 4).  The 16-bit type is the
// library's (bf16: libhyperpri_hip.so, IEEE half: libhyperpri_hip_f16.so).
extern "C" int hpri_conv3x3_ingest_h16(const float* x, const void* wp, const float* bias, void* y, int y_cs, int y_coff, int N, int C,
                                       int H, int W, int Cout, int Cout_pad, int relu, hipStream_t stream) {
  HPRI_REQUIRE(x && wp && y, "conv3x3_ingest_h16: null pointer");
  HPRI_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "conv3x3_ingest_h16: empty cube");
  HPRI_REQUIRE(Cout > 0 && Cout % 4 == 0 && Cout_pad % 64 == 0 && Cout <= Cout_pad, "conv3x3_ingest_h16: Cout must be a multiple of 4, Cout_pad a multiple of 64 >= Cout");
  HPRI_REQUIRE(((uintptr_t)x & 3) == 0 && ((uintptr_t)wp & 15) == 0 && ((uintptr_t)y & 7) == 0, "conv3x3_ingest_h16: misaligned pointer");
  HPRI_REQUIRE(y_cs % 4 == 0 && y_coff % 4 == 0 && y_coff + Cout <= y_cs, "conv3x3_ingest_h16: the output rows must be 8-byte aligned and hold Cout channels");
  HPRI_REQUIRE((long long)C * H * W * 4 < 0x7FFFFF00ll, "conv3x3_ingest_h16: one image of the cube exceeds 2 GiB (32-bit buffer offsets)");
  HPRI_REQUIRE((long long)((C + 31) / 32) * 9 * Cout_pad * 64 < 0x7FFFFF00ll, "conv3x3_ingest_h16: packed weights exceed 2 GiB");
  IngestArgs a;
  a.x = x; a.wp = reinterpret_cast<const h16_t*>(wp); a.bias = bias; a.y = reinterpret_cast<h16_t*>(y); a.y_cs = y_cs; a.y_coff = y_coff;
  a.N = N; a.C = C; a.H = H; a.W = W; a.Cout = Cout; a.Cout_pad = Cout_pad; a.relu = relu ? 1 : 0;
  // column bands: 32-wide tiles (8 rows), and what W mod 32 leaves over as one column of 8-wide (32 rows) or 16-wide (16 rows) tiles
  // -- fewer, fuller items than a ragged 32-wide column (2 x 238x608x968: 4598 items = 8.98 rounds of 512 workgroups instead of 9.2)
  const int rem = W % 32, n32 = rem > 16 ? (W + 31) / 32 : W / 32;
  a.nseg = 0;
  int first = 0;
  if (n32 > 0) { a.seg_twl[a.nseg] = 5; a.seg_xbeg[a.nseg] = 0; a.seg_ntx[a.nseg] = n32; a.seg_first[a.nseg] = first; first += n32 * ((H + 7) / 8); a.nseg++; }
  if (rem > 0 && rem <= 16) {
    const int twl = rem <= 8 ? 3 : 4, th = 256 >> twl;
    a.seg_twl[a.nseg] = twl; a.seg_xbeg[a.nseg] = n32 * 32; a.seg_ntx[a.nseg] = 1; a.seg_first[a.nseg] = first; first += (H + th - 1) / th; a.nseg++;
  }
  for (int k = a.nseg; k < 2; ++k) { a.seg_twl[k] = 5; a.seg_xbeg[k] = 0; a.seg_ntx[k] = 1; a.seg_first[k] = first; }
  a.tiles_img = first;
  const long long ntiles = (long long)N * a.tiles_img;
  a.nb_count = Cout_pad / 64;
  const long long items = ntiles * a.nb_count;
  HPRI_REQUIRE(items < (1ll << 28), "conv3x3_ingest_h16: too many work items");
  a.ntiles = (int)ntiles;
  a.per_xcd = (int)((items + 7) / 8);
  int nloc = 2 * hpri_cu_count() / 8;
  if (nloc < 1) nloc = 1;
  if (nloc > a.per_xcd) nloc = a.per_xcd;
  hipLaunchKernelGGL(conv_ingest_kernel, dim3((unsigned)(nloc * 8)), dim3(256), 0, stream, a);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}
