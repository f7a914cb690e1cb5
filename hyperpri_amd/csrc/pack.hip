// Weight re-layout kernels: PyTorch parameter layouts (OIHW conv, (Cin,Cout,2,2) transposed conv,
// (out,in) linear) -> the [chunk][tap][32 k][Cout_pad] panels conv_fwd.hip streams through LDS.
// The packed copies are derived caches; the nn.Parameter stays the source of truth.
#include "common.h"

// mode 0: conv forward      wp[t][k=c][n]        = W[n][c][t]                 (W: [Cout][Cin][T])
// mode 1: conv data-grad    wp[t][k=n][col=c]    = W[n][c][T-1-t]             (K = Cout, cols = Cin)
// mode 2: convT forward     wp[0][k=ci][n=tap*Cup+co] = Wt[ci][co][tap]       (Wt: [Cin][Cup][4])
// mode 3: convT data-grad   wp[0][k=tap*Cup+co][col=ci] = Wt[ci][co][tap]
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int mode,
                                   int K, int Ncols, int Ncols_pad, int T, int chunks, int Cup,
                                   int src_d0, int src_d1, const float* __restrict__ colscale) {
  // one thread per packed element; layout [chunk][t][kk(32)][Ncols_pad]
  const size_t total = (size_t)chunks * T * 32 * Ncols_pad;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(idx % Ncols_pad);
    size_t r = idx / Ncols_pad;
    const int kk = (int)(r % 32); r /= 32;
    const int t = (int)(r % T);
    const int chunk = (int)(r / T);
    const int k = chunk * 32 + kk;
    float v = 0.f;
    if (k < K && col < Ncols) {
      if (mode == 0) v = w[((size_t)col * src_d1 + k) * T + t];                  // W[n=col][c=k][t]
      else if (mode == 1) v = w[((size_t)k * src_d1 + col) * T + (T - 1 - t)];   // W[n=k][c=col][flip t]
      else if (mode == 2) { const int tap = col / Cup, co = col - tap * Cup; v = w[((size_t)k * Cup + co) * 4 + tap]; }
      else { const int tap = k / Cup, co = k - tap * Cup; v = w[((size_t)col * Cup + co) * 4 + tap]; }
      if (colscale != nullptr) v *= colscale[col];     // eval-mode BN folded into the conv: w' = w * gamma/sqrt(var+eps)
    }
    wp[idx] = v;
  }
}

extern "C" size_t hpri_packed_weight_floats(int K, int Ncols_pad, int T) {
  return (size_t)hpri_cdiv(K, 32) * T * 32 * Ncols_pad;
}

// K = GEMM reduction length per tap (before padding), Ncols = GEMM output columns.
extern "C" int hpri_pack_weight(const float* w, float* wp, int mode, int K, int Ncols, int Ncols_pad,
                                int T, int Cup, int src_d0, int src_d1, hipStream_t stream) {
  HPRI_REQUIRE(w && wp, "pack_weight: null pointer");
  HPRI_REQUIRE(mode >= 0 && mode <= 3 && K > 0 && Ncols > 0 && Ncols_pad >= Ncols && Ncols_pad % 64 == 0,
               "pack_weight: bad arguments");
  const int chunks = hpri_cdiv(K, 32);
  const size_t total = (size_t)chunks * T * 32 * Ncols_pad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, stream, w, wp, mode, K, Ncols, Ncols_pad, T,
                     chunks, Cup, src_d0, src_d1, (const float*)nullptr);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// Forward pack (mode 0) with a per-output-channel scale: the eval-mode BatchNorm of a conv->BN->ReLU stage folded into
// the conv (PLTrainer's predict/validate/test paths run the modules in eval mode, PLTrainer.py:142-162).
extern "C" int hpri_pack_weight_scaled(const float* w, float* wp, const float* colscale, int K, int Ncols, int Ncols_pad,
                                       int T, int src_d1, hipStream_t stream) {
  HPRI_REQUIRE(w && wp && colscale, "pack_weight_scaled: null pointer");
  HPRI_REQUIRE(K > 0 && Ncols > 0 && Ncols_pad >= Ncols && Ncols_pad % 64 == 0, "pack_weight_scaled: bad arguments");
  const int chunks = hpri_cdiv(K, 32);
  const size_t total = (size_t)chunks * T * 32 * Ncols_pad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, stream, w, wp, 0, K, Ncols, Ncols_pad, T, chunks, 0, 0,
                     src_d1, colscale);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// ---- bf16 panels for conv_fwd_bf16.hip: [chunk][tap][Ncols_pad][32 k] (k contiguous per output column, the order the
// MFMA B operand wants); same four modes as the fp32 pack; round-to-nearest-even.
// split = 1 (mode "bf16x3"): two planes per tap, [chunk][tap][plane][Ncols_pad][32]: hi = bf16(w), lo = bf16(w - hi);
// split = 2 (mode "bf16x6"): three planes hi, mid, lo (24 mantissa bits: the fp32 value exactly)
// gap_len > 0 (modes 0 and 1): the INPUT-channel axis of the layer carries gap_len structural-zero channels from gap_at on (the
// padded concat of the bf16 plane mode: [a | zeros to the next multiple of 32 | b]); K (mode 0) resp. Ncols (mode 1) count the
// padded axis, the weight tensor has the reference's unpadded width.
__global__ void pack_weight_bf16_kernel(const float* __restrict__ w, h16_t* __restrict__ wp, int mode, int K, int Ncols,
                                        int Ncols_pad, int T, int chunks, int src_d1, int Cup, int split,
                                        const float* __restrict__ colscale, int gap_at = 0, int gap_len = 0) {
  const size_t total = (size_t)chunks * T * Ncols_pad * 32;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int kk = (int)(idx & 31);
    size_t r = idx >> 5;
    int col = (int)(r % Ncols_pad); r /= Ncols_pad;
    const int t = (int)(r % T);
    const int chunk = (int)(r / T);
    int k = chunk * 32 + kk;
    float v = 0.f;
    bool in_gap = false;
    const bool live = k < K && col < Ncols;
    if (gap_len > 0) {
      int& ax = mode == 0 ? k : col;             // the input-channel axis
      in_gap = ax >= gap_at && ax < gap_at + gap_len;
      if (ax >= gap_at + gap_len) ax -= gap_len;
    }
    if (live && !in_gap) {
      if (mode == 0) v = w[((size_t)col * src_d1 + k) * T + t];
      else if (mode == 1) v = w[((size_t)k * src_d1 + col) * T + (T - 1 - t)];
      else if (mode == 2) { const int tap = col / Cup, co = col - tap * Cup; v = w[((size_t)k * Cup + co) * 4 + tap]; }
      else { const int tap = k / Cup, co = k - tap * Cup; v = w[((size_t)col * Cup + co) * 4 + tap]; }
      if (colscale != nullptr) v *= colscale[col];     // eval-mode BN folded into the conv (scaled in fp32, then split)
    }
    if (!split) {
      wp[idx] = (h16_t)v;
    } else {               // split + 1 planes per tap: hi, (mid,) lo -- each the bf16 rounding of what the previous ones left
      const int npl = split + 1;
      const size_t plane = (size_t)Ncols_pad * 32;
      const size_t o = ((size_t)(chunk * T + t) * npl) * plane + (size_t)col * 32 + kk;
      float rest = v;
      for (int pl = 0; pl < npl; ++pl) {
        const h16_t h = (h16_t)rest;
        wp[o + pl * plane] = h;
        rest -= (float)h;
      }
    }
  }
}

extern "C" int hpri_pack_weight_bf16(const float* w, void* wp, int mode, int K, int Ncols, int Ncols_pad, int T,
                                     int src_d1, int Cup, int split, hipStream_t stream) {
  HPRI_REQUIRE(w && wp, "pack_weight_bf16: null pointer");
  HPRI_REQUIRE(mode >= 0 && mode <= 3 && K > 0 && Ncols > 0 && Ncols_pad >= Ncols && Ncols_pad % 64 == 0,
               "pack_weight_bf16: bad arguments");
  HPRI_REQUIRE(split >= 0 && split <= 2, "pack_weight_bf16: split must be 0, 1 or 2");
  const int chunks = hpri_cdiv(K, 32);
  const size_t total = (size_t)chunks * T * Ncols_pad * 32;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(blocks), dim3(256), 0, stream, w, reinterpret_cast<h16_t*>(wp), mode, K,
                     Ncols, Ncols_pad, T, chunks, src_d1, Cup, split, (const float*)nullptr);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// hpri_pack_weight_bf16 for a layer whose input-channel axis is padded with gap_len zero channels at gap_at (modes 0 / 1 only):
// K (mode 0) resp. Ncols (mode 1) is the PADDED channel count, src_d1 the weight tensor's own (unpadded) input width.
extern "C" int hpri_pack_weight_bf16_gap(const float* w, void* wp, int mode, int K, int Ncols, int Ncols_pad, int T, int src_d1,
                                         int gap_at, int gap_len, hipStream_t stream) {
  HPRI_REQUIRE(w && wp, "pack_weight_bf16_gap: null pointer");
  HPRI_REQUIRE((mode == 0 || mode == 1) && K > 0 && Ncols > 0 && Ncols_pad >= Ncols && Ncols_pad % 64 == 0, "pack_weight_bf16_gap: bad arguments");
  HPRI_REQUIRE(gap_at >= 0 && gap_len >= 0 && gap_at + gap_len <= (mode == 0 ? K : Ncols) && (mode == 0 ? K : Ncols) - gap_len == src_d1,
               "pack_weight_bf16_gap: the padded axis minus the gap must be the weight's input width");
  const int chunks = hpri_cdiv(K, 32);
  const size_t total = (size_t)chunks * T * Ncols_pad * 32;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(blocks), dim3(256), 0, stream, w, reinterpret_cast<h16_t*>(wp), mode, K, Ncols, Ncols_pad,
                     T, chunks, src_d1, 0, 0, (const float*)nullptr, gap_at, gap_len);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// Forward pack (mode 0) in the bf16 plane layouts with a per-output-channel scale: eval-mode BN folded into the conv
// for the bf16 / bf16x3 / bf16x6 predict path (PLTrainer.py:530-532).
extern "C" int hpri_pack_weight_bf16_scaled(const float* w, void* wp, const float* colscale, int K, int Ncols, int Ncols_pad,
                                            int T, int src_d1, int split, hipStream_t stream) {
  HPRI_REQUIRE(w && wp && colscale, "pack_weight_bf16_scaled: null pointer");
  HPRI_REQUIRE(K > 0 && Ncols > 0 && Ncols_pad >= Ncols && Ncols_pad % 64 == 0, "pack_weight_bf16_scaled: bad arguments");
  HPRI_REQUIRE(split >= 0 && split <= 2, "pack_weight_bf16_scaled: split must be 0, 1 or 2");
  const int chunks = hpri_cdiv(K, 32);
  const size_t total = (size_t)chunks * T * Ncols_pad * 32;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(blocks), dim3(256), 0, stream, w, reinterpret_cast<h16_t*>(wp), 0, K,
                     Ncols, Ncols_pad, T, chunks, src_d1, 0, split, colscale);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// ---- content fingerprint of a parameter tensor (engine.py: HPRI_PACK_VERIFY / engine.verify_packs) ----------------------
// The packed panels are caches keyed on torch's version counter, which writes through `p.data` do not advance.  The verify
// mode of the cache compares this 64-bit fingerprint of the fp32 bits (sum of word * odd multiplier of its index: integer
// adds, so the result does not depend on the order the workgroups finish in) against the one taken when the pack was built.
__global__ void fingerprint_kernel(const unsigned* __restrict__ w, long long n, unsigned long long* __restrict__ out) {
  unsigned long long acc = 0ull;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    acc += (unsigned long long)w[i] * (2ull * (unsigned long long)i + 0x9E3779B97F4A7C15ull);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

extern "C" int hpri_fingerprint(const void* w, long long n_words, unsigned long long* out, hipStream_t stream) {
  HPRI_REQUIRE(w && out && n_words > 0, "fingerprint: null pointer or empty tensor");
  HPRI_REQUIRE(((uintptr_t)w & 3) == 0 && ((uintptr_t)out & 7) == 0, "fingerprint: misaligned pointer");
  if (hipMemsetAsync(out, 0, sizeof(unsigned long long), stream) != hipSuccess)
    return hpri_set_error(HPRI_ERR_LAUNCH, "fingerprint: hipMemsetAsync failed");
  long long blocks = (n_words + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fingerprint_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const unsigned*>(w),
                     n_words, out);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}
