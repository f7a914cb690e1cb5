// BatchNorm{1,2,3}d (+ fused ReLU) over NHWC fp32 activations: training-mode statistics from the conv
// epilogue's per-tile partials, normalise+ReLU, and the backward pass.  Reference semantics:
// torch.nn.BatchNorm2d/3d/1d as used at model_parts.py:23,26 and models.py:113,172,178 -- biased batch
// variance (eps 1e-5) for normalisation, running stats updated with momentum and the UNBIASED variance,
// running stats in eval mode.  "Groups" (G) are independent statistic sets: G = 1 for the conv nets,
// G = N images for SpectralUNET whose per-image loop (models.py:132) makes every image its own batch.
// All cross-workgroup reductions are two-stage and fixed-order (bitwise reproducible; no float atomics).
#include "common.h"

// -------------------------------------------------------------------------------------------------
// forward statistics: combine per-tile (mean, M2, count) partials -> mean, invstd, scale, shift
// -------------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float4* __restrict__ part, int tiles_per_group, int Cp, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ var_unbiased,
                                   float* __restrict__ scale, float* __restrict__ shift, float momentum,
                                   float* __restrict__ rm, float* __restrict__ rv, long long* __restrict__ nbt) {
  // block = 32 slices x 32 channels (1024 threads); grid = (ceil(C/32), G)
  __shared__ double s1[32][32], s2[32][32], sn[32][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, g = blockIdx.y;
  double a1 = 0.0, a2 = 0.0, an = 0.0;
  if (c < C) {
    const float4* p = part + (size_t)g * tiles_per_group * Cp + c;
    int t = sl;
    for (; t + 96 < tiles_per_group; t += 128) {      // 4 independent loads in flight (latency-bound otherwise)
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p[(size_t)(t + 32 * u) * Cp];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double m = v[u].x, n = v[u].z;
        a1 += n * m;
        a2 += (double)v[u].y + n * m * m;
        an += n;
      }
    }
    for (; t < tiles_per_group; t += 32) {
      const float4 v = p[(size_t)t * Cp];
      const double m = v.x, n = v.z;
      a1 += n * m;
      a2 += (double)v.y + n * m * m;
      an += n;
    }
  }
  s1[sl][cl] = a1; s2[sl][cl] = a2; sn[sl][cl] = an;
  __syncthreads();
  if (sl == 0 && c < C) {
    double t1 = 0.0, t2 = 0.0, tn = 0.0;
    for (int k = 0; k < 32; ++k) { t1 += s1[k][cl]; t2 += s2[k][cl]; tn += sn[k][cl]; }
    const double mu = t1 / tn;
    double var = t2 / tn - mu * mu;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    const int o = g * C + c;
    mean[o] = (float)mu;
    invstd[o] = (float)is;
    var_unbiased[o] = (float)(tn > 1.0 ? var * tn / (tn - 1.0) : var);
    const float sc = gamma[c] * (float)is;
    scale[o] = sc;
    shift[o] = beta[c] - (float)mu * sc;
    if (rm != nullptr) {   // one group only: the running-stat update rides along (G > 1 needs the groups in order)
      rm[c] = (1.f - momentum) * rm[c] + momentum * mean[o];
      rv[c] = (1.f - momentum) * rv[c] + momentum * var_unbiased[o];
      if (c == 0 && nbt != nullptr) *nbt += 1;
    }
  }
}

// The same for MANY tiles per group (a 608 x 968 layer: 9272 tiles x 64 channels = 9.5 MB of partials, which the kernel
// above read with two workgroups: 80 us).  block = 256 tile slices x 4 channels, grid = (ceil(C/4), G): 16+ workgroups, 64-byte
// segments; slices meet by a fixed shuffle tree inside a wave, the 16 waves in order through LDS (deterministic).
__global__ __launch_bounds__(1024) void bn_finalize_wide_kernel(
    const float4* __restrict__ part, int tiles_per_group, int Cp, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float* __restrict__ mean, float* __restrict__ invstd,
    float* __restrict__ var_unbiased, float* __restrict__ scale, float* __restrict__ shift, float momentum,
    float* __restrict__ rm, float* __restrict__ rv, long long* __restrict__ nbt) {
  __shared__ double sw[16][4][3];
  const int cl = threadIdx.x & 3, sl = threadIdx.x >> 2;          // channel of the block, tile slice (0..255)
  const int c = blockIdx.x * 4 + cl, g = blockIdx.y;
  double a1 = 0.0, a2 = 0.0, an = 0.0;
  if (c < C) {
    const float4* p = part + (size_t)g * tiles_per_group * Cp + c;
    int t = sl;
    for (; t + 768 < tiles_per_group; t += 1024) {    // 4 independent loads in flight
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p[(size_t)(t + 256 * u) * Cp];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double m = v[u].x, n = v[u].z;
        a1 += n * m;
        a2 += (double)v[u].y + n * m * m;
        an += n;
      }
    }
    for (; t < tiles_per_group; t += 256) {
      const float4 v = p[(size_t)t * Cp];
      const double m = v.x, n = v.z;
      a1 += n * m;
      a2 += (double)v.y + n * m * m;
      an += n;
    }
  }
#pragma unroll
  for (int d = 4; d < 64; d <<= 1) {                  // the 16 slices of a wave that share a channel: lanes cl, cl+4, ...
    a1 += __shfl_xor(a1, d);
    a2 += __shfl_xor(a2, d);
    an += __shfl_xor(an, d);
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 4) { sw[wv][cl][0] = a1; sw[wv][cl][1] = a2; sw[wv][cl][2] = an; }
  __syncthreads();
  if (threadIdx.x < 4 && c < C) {
    double t1 = 0.0, t2 = 0.0, tn = 0.0;
    for (int k = 0; k < 16; ++k) { t1 += sw[k][cl][0]; t2 += sw[k][cl][1]; tn += sw[k][cl][2]; }
    const double mu = t1 / tn;
    double var = t2 / tn - mu * mu;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    const int o = g * C + c;
    mean[o] = (float)mu;
    invstd[o] = (float)is;
    var_unbiased[o] = (float)(tn > 1.0 ? var * tn / (tn - 1.0) : var);
    const float sc = gamma[c] * (float)is;
    scale[o] = sc;
    shift[o] = beta[c] - (float)mu * sc;
    if (rm != nullptr) {
      rm[c] = (1.f - momentum) * rm[c] + momentum * mean[o];
      rv[c] = (1.f - momentum) * rv[c] + momentum * var_unbiased[o];
      if (c == 0 && nbt != nullptr) *nbt += 1;
    }
  }
}

// running = (1-m)*running + m*stat, applied for g = 0..G-1 in order (SpectralUNET advances N times per call)
__global__ void bn_update_running_kernel(const float* __restrict__ mean, const float* __restrict__ var_unbiased,
                                         int G, int C, float momentum, float* __restrict__ rm, float* __restrict__ rv,
                                         long long* __restrict__ nbt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float m = rm[c], v = rv[c];
    for (int g = 0; g < G; ++g) {
      m = (1.f - momentum) * m + momentum * mean[g * C + c];
      v = (1.f - momentum) * v + momentum * var_unbiased[g * C + c];
    }
    rm[c] = m; rv[c] = v;
  }
  if (c == 0 && nbt != nullptr) *nbt += G;
}

__global__ void bn_eval_prepare_kernel(const float* __restrict__ rm, const float* __restrict__ rv,
                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int C,
                                       float* __restrict__ mean, float* __restrict__ invstd,
                                       float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float is = 1.f / sqrtf(rv[c] + eps);
    mean[c] = rm[c]; invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc; shift[c] = beta[c] - rm[c] * sc;
  }
}

// eval-mode folding: scale[c] = gamma/sqrt(rv+eps); fbias[c] = (conv_bias[c] - rm[c]) * scale[c] + beta[c]
__global__ void bn_fold_kernel(const float* __restrict__ rm, const float* __restrict__ rv, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ conv_bias, float eps, int C,
                               float* __restrict__ scale, float* __restrict__ fbias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    fbias[c] = ((conv_bias ? conv_bias[c] : 0.f) - rm[c]) * sc + beta[c];
  }
}

// -------------------------------------------------------------------------------------------------
// y = relu(x*scale + shift)  (float4 over channels; pad channels C..Cw are written as zeros)
// -------------------------------------------------------------------------------------------------
// grid = (pixel blocks, ceil(Cw4/CQ), G); block = 256 = ROWS x CQ channel quads; per-channel constants live in
// registers for the whole pixel loop (no integer division, no per-element parameter loads)

// The pre-BatchNorm tensor x (a convolution's output) is fp32, or -- bf16 precision mode with HPRI_YR_BF16, conv_bf16v3.hip --
// stored as bf16: XB selects the element type of `x`; the arithmetic is fp32 either way.
template <bool XB>
__device__ __forceinline__ float4 bn_load_x4(const float* __restrict__ x, size_t idx) {
  if (XB) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(reinterpret_cast<const h16_t*>(x) + idx);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  return *reinterpret_cast<const float4*>(x + idx);
}

template <bool XB>
__global__ void bn_apply_relu_kernel(const float* __restrict__ x, int x_cs, int x_coff, float* __restrict__ y,
                                     int y_cs, int y_coff, const float* __restrict__ scale,
                                     const float* __restrict__ shift, int pix_per_group, int C, int Cw, int CQ,
                                     int relu, PlaneOut pl) {
  const int rows = 256 / CQ;
  const int cq = threadIdx.x % CQ, pr = threadIdx.x / CQ;
  const int c = (blockIdx.y * CQ + cq) * 4;
  const bool in_f = c < Cw, in_p = pl.p != nullptr && c < pl.cw;
  if (!in_f && !in_p) return;
  const int g = blockIdx.z;
  float sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool ok = c + j < C;
    sc[j] = ok ? scale[g * C + c + j] : 0.f;
    sh[j] = ok ? shift[g * C + c + j] : 0.f;
  }
  const int per = (pix_per_group + gridDim.x - 1) / gridDim.x;
  const int q0 = blockIdx.x * per;
  const int q1 = min(pix_per_group, q0 + per);
  const size_t base = (size_t)g * pix_per_group;
  for (int q = q0 + pr; q < q1; q += rows) {
    const size_t p = base + q;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in_f) {
      const float4 v = bn_load_x4<XB>(x, p * x_cs + x_coff + c);
      o.x = v.x * sc[0] + sh[0]; o.y = v.y * sc[1] + sh[1]; o.z = v.z * sc[2] + sh[2]; o.w = v.w * sc[3] + sh[3];
      if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      if (y != nullptr) *reinterpret_cast<float4*>(y + p * y_cs + y_coff + c) = o;
    }
    if (in_p) plane_store4(pl, p, c, o.x, o.y, o.z, o.w);
  }
}

// -------------------------------------------------------------------------------------------------
// backward, stage 1: per-channel partial sums of g = dy * [y > 0] and g * xhat over pixel ranges
// -------------------------------------------------------------------------------------------------
// grid = (nblk, ceil(C4/CQ), G); block = 256 = ROWS x CQ; partial layout [G][nblk][2][Cq4*4]
template <int MODE, bool XB = false, bool DB = false>  // 0: BN+ReLU backward sums (s1 = sum g, s2 = sum g*xhat); 1: plain column sum of dy; DB: dy stored as bf16
__global__ void col_reduce_kernel(const float* __restrict__ dy, int dy_cs, int dy_coff, const float* __restrict__ x,
                                  int x_cs, int x_coff, const float* __restrict__ mean,
                                  const float* __restrict__ invstd, const float* __restrict__ scale,
                                  const float* __restrict__ shift, long long pix_per_group, int C, int CQ,
                                  int relu, float* __restrict__ part, int Cpart) {
  __shared__ float4 red[2][256];
  const int rows = 256 / CQ;
  const int cq = threadIdx.x % CQ, pr = threadIdx.x / CQ;
  const int c = (blockIdx.y * CQ + cq) * 4;
  const int g = blockIdx.z;
  const long long per = (pix_per_group + gridDim.x - 1) / gridDim.x;
  const long long p0 = g * pix_per_group + (long long)blockIdx.x * per;
  long long p1 = p0 + per;
  const long long pend = (long long)(g + 1) * pix_per_group;
  if (p1 > pend) p1 = pend;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    float mu[4], is[4], sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = (c + j < C) && MODE == 0;
      mu[j] = ok ? mean[g * C + c + j] : 0.f;
      is[j] = ok ? invstd[g * C + c + j] : 0.f;
      sc[j] = ok ? scale[g * C + c + j] : 0.f;
      sh[j] = ok ? shift[g * C + c + j] : 0.f;
    }
    // four pixels per thread and iteration, all eight loads issued before the first use: with one pixel per iteration a
    // thread had two 16-byte loads in flight and the pass ran at 56 % of the HBM rate of its two tensor sweeps (round 2: 1.57 ms
    // per bf16-mode step against 0.88); the summation order per thread is unchanged (pixels in ascending order)
    long long p = p0 + pr;
    for (; p + 3 * rows < p1; p += 4 * rows) {
      float4 dv[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        dv[u] = bn_load_x4<DB>(dy, (p + u * rows) * dy_cs + dy_coff + c);
        if (MODE == 0) xv[u] = bn_load_x4<XB>(x, (p + u * rows) * x_cs + x_coff + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
        if (MODE == 0) {
          const float xx[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float gj = (!relu || (xx[j] * sc[j] + sh[j] > 0.f)) ? d[j] : 0.f;
            s1[j] += gj;
            s2[j] += gj * ((xx[j] - mu[j]) * is[j]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) s1[j] += d[j];
        }
      }
    }
    for (; p < p1; p += rows) {
      const float4 dv = bn_load_x4<DB>(dy, p * dy_cs + dy_coff + c);
      const float d[4] = {dv.x, dv.y, dv.z, dv.w};
      if (MODE == 0) {
        const float4 xv = bn_load_x4<XB>(x, p * x_cs + x_coff + c);
        const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float gj = (!relu || (xx[j] * sc[j] + sh[j] > 0.f)) ? d[j] : 0.f;
          s1[j] += gj;
          s2[j] += gj * ((xx[j] - mu[j]) * is[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) s1[j] += d[j];
      }
    }
  }
  red[0][threadIdx.x] = make_float4(s1[0], s1[1], s1[2], s1[3]);
  red[1][threadIdx.x] = make_float4(s2[0], s2[1], s2[2], s2[3]);
  __syncthreads();
  if (pr == 0 && c < Cpart) {
    float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
    for (int r = 0; r < rows; ++r) {
      const float4 a = red[0][r * CQ + cq], b = red[1][r * CQ + cq];
      t1.x += a.x; t1.y += a.y; t1.z += a.z; t1.w += a.w;
      t2.x += b.x; t2.y += b.y; t2.z += b.z; t2.w += b.w;
    }
    float* o = part + ((size_t)(g * gridDim.x + blockIdx.x) * 2) * Cpart + c;
    *reinterpret_cast<float4*>(o) = t1;
    *reinterpret_cast<float4*>(o + Cpart) = t2;
  }
}

// stage 2: sums[g][k][c] = sum over blocks (double accumulation, fixed order); k in {0,1}
__global__ void col_finalize_kernel(const float* __restrict__ part, int nblk, int Cpart, int C, float* __restrict__ sums,
                                    float* __restrict__ out1, float* __restrict__ out2, int accumulate,
                                    float* __restrict__ zero_out = nullptr) {
  __shared__ double s[2][32][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, g = blockIdx.y;
  double a1 = 0.0, a2 = 0.0;
  if (c < C) {
    const float* p = part + (size_t)g * nblk * 2 * Cpart + c;
    int b = sl;
    for (; b + 96 < nblk; b += 128) {
      float u1[4], u2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { u1[u] = p[(size_t)(b + 32 * u) * 2 * Cpart]; u2[u] = p[(size_t)(b + 32 * u) * 2 * Cpart + Cpart]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a1 += u1[u]; a2 += u2[u]; }
    }
    for (; b < nblk; b += 32) { a1 += p[(size_t)b * 2 * Cpart]; a2 += p[(size_t)b * 2 * Cpart + Cpart]; }
  }
  s[0][sl][cl] = a1; s[1][sl][cl] = a2;
  __syncthreads();
  if (sl == 0 && c < C) {
    double t1 = 0.0, t2 = 0.0;
    for (int k = 0; k < 32; ++k) { t1 += s[0][k][cl]; t2 += s[1][k][cl]; }
    sums[((size_t)g * 2) * C + c] = (float)t1;
    sums[((size_t)g * 2 + 1) * C + c] = (float)t2;
    // one group only: the parameter gradients ride along (out1 (+)= first sum, out2 (+)= second sum)
    if (out1 != nullptr) out1[c] = accumulate ? out1[c] + (float)t1 : (float)t1;
    if (out2 != nullptr) out2[c] = accumulate ? out2[c] + (float)t2 : (float)t2;
    if (zero_out != nullptr) zero_out[c] = 0.f;      // (g == 0 only: the caller passes it for one group)
  }
}

// Many partial rows (one per tile of a convolution launch that took the reduction in its epilogue: 4.6 k rows for a full-size
// layer) are first folded to `S` rows by S x C/32 workgroups -- col_finalize_kernel alone walks them with C/32 workgroups.
// folded[s][k][c] = sum over the rows of slice s (double accumulation, fixed order); grid (ceil(C/32), S), 1024 threads
__global__ void col_fold_kernel(const float* __restrict__ part, int nblk, int Cpart, int C, float* __restrict__ folded) {
  __shared__ double s[2][32][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(b0 + per, nblk);
  double a1 = 0.0, a2 = 0.0;
  if (c < C) {
    const float* p = part + c;
    int b = b0 + sl;
    for (; b + 96 < b1; b += 128) {
      float u1[4], u2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { u1[u] = p[(size_t)(b + 32 * u) * 2 * Cpart]; u2[u] = p[(size_t)(b + 32 * u) * 2 * Cpart + Cpart]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a1 += u1[u]; a2 += u2[u]; }
    }
    for (; b < b1; b += 32) { a1 += p[(size_t)b * 2 * Cpart]; a2 += p[(size_t)b * 2 * Cpart + Cpart]; }
  }
  s[0][sl][cl] = a1; s[1][sl][cl] = a2;
  __syncthreads();
  if (sl == 0 && c < C) {
    double t1 = 0.0, t2 = 0.0;
    for (int k = 0; k < 32; ++k) { t1 += s[0][k][cl]; t2 += s[1][k][cl]; }
    folded[((size_t)blockIdx.y * 2) * Cpart + c] = (float)t1;
    folded[((size_t)blockIdx.y * 2 + 1) * Cpart + c] = (float)t2;
  }
}

// dgamma[c] (+)= sum_g s2[g][c], dbeta[c] (+)= sum_g s1[g][c]
__global__ void bn_param_grad_kernel(const float* __restrict__ sums, int G, int C, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int g = 0; g < G; ++g) { b += sums[((size_t)g * 2) * C + c]; a += sums[((size_t)g * 2 + 1) * C + c]; }
  if (dgamma != nullptr) dgamma[c] = accumulate ? dgamma[c] + a : a;
  dbeta[c] = accumulate ? dbeta[c] + b : b;
}

// dx = scale * (g - s1/Np - xhat * s2/Np)   (training) ;  dx = scale * g  (eval: use_batch_stats = 0)
// same 2-D mapping as bn_apply_relu_kernel; also emits per-block column sums of dx (the conv-bias gradient)
template <bool XB, bool DB = false>
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, int dy_cs, int dy_coff, const float* __restrict__ x,
                                    int x_cs, int x_coff, float* __restrict__ dx, int dx_cs, int dx_coff,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ sums, int pix_per_group, int C, int Cw, int CQ,
                                    int relu, int use_batch_stats, float* __restrict__ dxpart, int Cpart, PlaneOut pl) {
  __shared__ float4 red[256];
  const int rows = 256 / CQ;
  const int cq = threadIdx.x % CQ, pr = threadIdx.x / CQ;
  const int c = (blockIdx.y * CQ + cq) * 4;
  const int g = blockIdx.z;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool in_p = pl.p != nullptr && c < pl.cw;
  if (c >= Cw && in_p) {           // plane pad channels beyond the fp32 width: zeros
    const int per = (pix_per_group + gridDim.x - 1) / gridDim.x;
    const int q0 = blockIdx.x * per, q1 = min(pix_per_group, q0 + per);
    for (int q = q0 + pr; q < q1; q += rows) plane_store4(pl, (size_t)g * pix_per_group + q, c, 0.f, 0.f, 0.f, 0.f);
  }
  if (c < Cw) {
    const float inv_np = 1.f / (float)pix_per_group;
    float sc[4], sh[4], mu[4], is[4], k1[4], k2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = c + j < C;
      const int k = g * C + c + j;
      sc[j] = ok ? scale[k] : 0.f;
      sh[j] = ok ? shift[k] : 0.f;
      mu[j] = ok ? mean[k] : 0.f;
      is[j] = ok ? invstd[k] : 0.f;
      k1[j] = (ok && use_batch_stats) ? sums[((size_t)g * 2) * C + c + j] * inv_np : 0.f;
      k2[j] = (ok && use_batch_stats) ? sums[((size_t)g * 2 + 1) * C + c + j] * inv_np : 0.f;
    }
    const int per = (pix_per_group + gridDim.x - 1) / gridDim.x;
    const int q0 = blockIdx.x * per;
    const int q1 = min(pix_per_group, q0 + per);
    const size_t base = (size_t)g * pix_per_group;
    for (int q = q0 + pr; q < q1; q += rows) {
      const size_t p = base + q;
      const float4 dv = bn_load_x4<DB>(dy, p * dy_cs + dy_coff + c);
      const float4 xv = bn_load_x4<XB>(x, p * x_cs + x_coff + c);
      const float d[4] = {dv.x, dv.y, dv.z, dv.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gj = (!relu || (xx[j] * sc[j] + sh[j] > 0.f)) ? d[j] : 0.f;
        const float xh = (xx[j] - mu[j]) * is[j];
        o[j] = (c + j < C) ? sc[j] * (gj - k1[j] - xh * k2[j]) : 0.f;   // pad channels stay exactly zero
        acc[j] += o[j];
      }
      if (dx != nullptr) *reinterpret_cast<float4*>(dx + p * dx_cs + dx_coff + c) = make_float4(o[0], o[1], o[2], o[3]);
      if (in_p) plane_store4(pl, p, c, o[0], o[1], o[2], o[3]);
    }
  }
  if (dxpart != nullptr) {
    red[threadIdx.x] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (pr == 0 && c < Cpart) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int r = 0; r < rows; ++r) { const float4 a = red[r * CQ + cq]; t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w; }
      float* o = dxpart + ((size_t)(g * gridDim.x + blockIdx.x) * 2) * Cpart + c;
      *reinterpret_cast<float4*>(o) = t;
      *reinterpret_cast<float4*>(o + Cpart) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// ------------------------------------------- C ABI ---------------------------------------------
extern "C" int hpri_bn_apply_relu_pl(const float* x, int x_cs, int x_coff, float* y, int y_cs, int y_coff,
                                     const float* scale, const float* shift, long long P, long long pix_per_group,
                                     int C, int Cw, int relu, void* planes, long long plane_stride, int pl_cs, int pl_coff,
                                     int pl_cw, int npl, hipStream_t stream);
extern "C" int hpri_bn_relu_bwd_pl(const float* dy, int dy_cs, int dy_coff, const float* x, int x_cs, int x_coff,
                                   float* dx, int dx_cs, int dx_coff, const float* mean, const float* invstd,
                                   const float* scale, const float* shift, float* dgamma, float* dbeta,
                                   int accumulate_param_grads, float* dbias, int accumulate_dbias, float* workspace,
                                   size_t ws_floats, long long P, long long pix_per_group, int C, int Cw, int relu,
                                   int use_batch_stats, void* planes, long long plane_stride, int pl_cs, int pl_coff,
                                   int pl_cw, int npl, hipStream_t stream);
// channel quads per workgroup row (a power of two).  Up to 64 (256 channels) a workgroup holds 4+ pixel rows; tensors wider than
// 1024 channels (SpectralUNET-1650) take a whole 1024-channel run of ONE pixel: 4-KB contiguous pieces instead of 1-KB pieces 6.6 KB
// apart (option "bn_wide_cq", api.cpp; measured: its three BatchNorm kernels 2-4 % faster, a C3 step 135.9 -> 133.6 ms; the 512 / 1024
// channel layers of the U-Nets measured neutral and keep the narrow form)
static inline int pick_cq(int c4) {
  const int cap = (c4 > 256 && hpri_option(4) != 0) ? 256 : 64;
  int q = 1;
  while (q < c4 && q < cap) q <<= 1;
  return q;
}
static inline int ew_blocks(long long total) {
  long long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int hpri_bn_finalize(const float* partials, int tiles_per_group, int G, int Cp, int C,
                                const float* gamma, const float* beta, float eps, float momentum,
                                float* mean, float* invstd, float* var_unbiased, float* scale, float* shift,
                                float* running_mean, float* running_var, long long* num_batches_tracked,
                                hipStream_t stream) {
  HPRI_REQUIRE(partials && gamma && beta && mean && invstd && var_unbiased && scale && shift, "bn_finalize: null pointer");
  HPRI_REQUIRE(tiles_per_group > 0 && G > 0 && C > 0 && Cp >= C, "bn_finalize: bad sizes");
  const bool running = running_mean != nullptr && running_var != nullptr;
  const bool fused = running && G == 1;
  if (tiles_per_group >= 512)                       // few channels x many tiles: spread the tiles over more workgroups
    hipLaunchKernelGGL(bn_finalize_wide_kernel, dim3(hpri_cdiv(C, 4), G), dim3(1024), 0, stream,
                       reinterpret_cast<const float4*>(partials), tiles_per_group, Cp, C, gamma, beta, eps, mean, invstd,
                       var_unbiased, scale, shift, momentum, fused ? running_mean : nullptr, fused ? running_var : nullptr,
                       fused ? num_batches_tracked : nullptr);
  else
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(hpri_cdiv(C, 32), G), dim3(1024), 0, stream,
                       reinterpret_cast<const float4*>(partials), tiles_per_group, Cp, C, gamma, beta, eps, mean, invstd,
                       var_unbiased, scale, shift, momentum, fused ? running_mean : nullptr, fused ? running_var : nullptr,
                       fused ? num_batches_tracked : nullptr);
  HPRI_CHECK_LAUNCH();
  if (running && !fused) {
    hipLaunchKernelGGL(bn_update_running_kernel, dim3(hpri_cdiv(C, 256)), dim3(256), 0, stream, mean, var_unbiased, G, C,
                       momentum, running_mean, running_var, num_batches_tracked);
    HPRI_CHECK_LAUNCH();
  }
  return HPRI_OK;
}

extern "C" int hpri_bn_eval_prepare(const float* running_mean, const float* running_var, const float* gamma,
                                    const float* beta, float eps, int C, float* mean, float* invstd, float* scale,
                                    float* shift, hipStream_t stream) {
  HPRI_REQUIRE(running_mean && running_var && gamma && beta && mean && invstd && scale && shift, "bn_eval_prepare: null pointer");
  hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(hpri_cdiv(C, 256)), dim3(256), 0, stream, running_mean, running_var,
                     gamma, beta, eps, C, mean, invstd, scale, shift);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_bn_fold(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                            const float* conv_bias, float eps, int C, float* scale, float* fbias, hipStream_t stream) {
  HPRI_REQUIRE(running_mean && running_var && gamma && beta && scale && fbias && C > 0, "bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(hpri_cdiv(C, 256)), dim3(256), 0, stream, running_mean, running_var, gamma, beta,
                     conv_bias, eps, C, scale, fbias);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_bn_apply_relu(const float* x, int x_cs, int x_coff, float* y, int y_cs, int y_coff,
                                  const float* scale, const float* shift, long long P, long long pix_per_group,
                                  int C, int Cw, int relu, hipStream_t stream) {
  return hpri_bn_apply_relu_pl(x, x_cs, x_coff, y, y_cs, y_coff, scale, shift, P, pix_per_group, C, Cw, relu, nullptr, 0, 0, 0, 0,
                               0, stream);
}

// the same pass, also writing the result as bf16 planes (planes == nullptr: fp32 only); see PlaneOut
static int bn_apply_relu_impl(const float* x, bool x16, int x_cs, int x_coff, float* y, int y_cs, int y_coff,
                              const float* scale, const float* shift, long long P, long long pix_per_group,
                              int C, int Cw, int relu, void* planes, long long plane_stride, int pl_cs, int pl_coff,
                              int pl_cw, int npl, hipStream_t stream) {
  // y == nullptr with planes given: the activation is wanted as bf16 planes only (the inner tensor of a DoubleConv in the plane
  // mode: the next convolution and its weight gradient read nothing else)
  HPRI_REQUIRE(x && (y || planes) && scale && shift, "bn_apply_relu: null pointer");
  PlaneOut po;
  { const int rc_ = hpri_plane_out(&po, planes, plane_stride, pl_cs, pl_coff, pl_cw, npl, C); if (rc_ != HPRI_OK) return rc_; }
  HPRI_REQUIRE(Cw % 4 == 0 && Cw >= C && x_cs % 4 == 0 && y_cs % 4 == 0 && x_coff % 4 == 0 && y_coff % 4 == 0 &&
                   Cw + x_coff <= x_cs && Cw + y_coff <= y_cs,
               "bn_apply_relu: channel layout must be float4-aligned and fit the strides");
  HPRI_REQUIRE(P > 0 && pix_per_group > 0 && P % pix_per_group == 0, "bn_apply_relu: bad pixel counts");
  const int G = (int)(P / pix_per_group);
  const int c4 = (Cw > po.cw ? Cw : po.cw) >> 2, cq = pick_cq(c4), rows = 256 / cq, ycols = hpri_cdiv(c4, cq);
  long long nbx = 4096 / ((long long)ycols * G);
  const long long maxb = (pix_per_group + rows * 4 - 1) / (rows * 4);
  if (nbx > maxb) nbx = maxb;
  if (nbx < 1) nbx = 1;
  if (x16)
    hipLaunchKernelGGL(bn_apply_relu_kernel<true>, dim3((unsigned)nbx, ycols, G), dim3(256), 0, stream, x, x_cs, x_coff, y, y_cs,
                       y_coff, scale, shift, (int)pix_per_group, C, Cw, cq, relu, po);
  else
    hipLaunchKernelGGL(bn_apply_relu_kernel<false>, dim3((unsigned)nbx, ycols, G), dim3(256), 0, stream, x, x_cs, x_coff, y, y_cs,
                       y_coff, scale, shift, (int)pix_per_group, C, Cw, cq, relu, po);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_bn_apply_relu_pl(const float* x, int x_cs, int x_coff, float* y, int y_cs, int y_coff,
                                     const float* scale, const float* shift, long long P, long long pix_per_group,
                                     int C, int Cw, int relu, void* planes, long long plane_stride, int pl_cs, int pl_coff,
                                     int pl_cw, int npl, hipStream_t stream) {
  return bn_apply_relu_impl(x, false, x_cs, x_coff, y, y_cs, y_coff, scale, shift, P, pix_per_group, C, Cw, relu, planes, plane_stride,
                            pl_cs, pl_coff, pl_cw, npl, stream);
}

// the same with the pre-BN tensor stored as bf16 (x16: element (p, c) at x16[p * x_cs + x_coff + c]; hpri_conv_bf16v3 with bit 2 of
// `accumulate` writes it)
extern "C" int hpri_bn_apply_relu_x16(const void* x16, int x_cs, int x_coff, float* y, int y_cs, int y_coff,
                                      const float* scale, const float* shift, long long P, long long pix_per_group,
                                      int C, int Cw, int relu, void* planes, long long plane_stride, int pl_cs, int pl_coff,
                                      int pl_cw, int npl, hipStream_t stream) {
  HPRI_REQUIRE(((uintptr_t)x16 & 7) == 0, "bn_apply_relu_x16: the bf16 tensor must be 8-byte aligned");
  return bn_apply_relu_impl(reinterpret_cast<const float*>(x16), true, x_cs, x_coff, y, y_cs, y_coff, scale, shift, P, pix_per_group, C,
                            Cw, relu, planes, plane_stride, pl_cs, pl_coff, pl_cw, npl, stream);
}

extern "C" int hpri_col_reduce_plan(long long pix_per_group, int G, int C, int* nblk, int* Cpart) {
  const int c4 = hpri_cdiv(C, 4), cq = pick_cq(c4), rows = 256 / cq;
  const int ycols = hpri_cdiv(c4, cq);
  long long nb = 1024 / ((long long)ycols * G);
  const long long maxb = (pix_per_group + rows * 8 - 1) / (rows * 8);
  if (nb > maxb) nb = maxb;
  if (nb < 1) nb = 1;
  *nblk = (int)nb;
  *Cpart = ycols * cq * 4;
  return HPRI_OK;
}

// BN(+ReLU) backward: reduce -> finalize -> parameter grads -> dx (+ column sums of dx = gradient of the bias of
// the conv in front, optional).  workspace floats = 2 * (G*nblk*2*Cpart + G*2*C), see hpri_col_reduce_plan.
extern "C" int hpri_bn_relu_bwd(const float* dy, int dy_cs, int dy_coff, const float* x, int x_cs, int x_coff,
                                float* dx, int dx_cs, int dx_coff, const float* mean, const float* invstd,
                                const float* scale, const float* shift, float* dgamma, float* dbeta,
                                int accumulate_param_grads, float* dbias, int accumulate_dbias, float* workspace,
                                size_t ws_floats, long long P, long long pix_per_group, int C, int Cw, int relu,
                                int use_batch_stats, hipStream_t stream) {
  return hpri_bn_relu_bwd_pl(dy, dy_cs, dy_coff, x, x_cs, x_coff, dx, dx_cs, dx_coff, mean, invstd, scale, shift, dgamma, dbeta,
                             accumulate_param_grads, dbias, accumulate_dbias, workspace, ws_floats, P, pix_per_group, C, Cw, relu,
                             use_batch_stats, nullptr, 0, 0, 0, 0, 0, stream);
}

// the same, with dx also written as bf16 planes for the data-gradient / weight-gradient kernels of the bf16 modes
static int bn_relu_bwd_impl(const float* ext_part, int ext_nblk, int ext_cpart,
                            const float* dy, int dy_cs, int dy_coff, const float* x, bool x16, int x_cs, int x_coff,
                            bool dy16,
                            float* dx, int dx_cs, int dx_coff, const float* mean, const float* invstd,
                            const float* scale, const float* shift, float* dgamma, float* dbeta,
                            int accumulate_param_grads, float* dbias, int accumulate_dbias, float* workspace,
                            size_t ws_floats, long long P, long long pix_per_group, int C, int Cw, int relu,
                            int use_batch_stats, void* planes, long long plane_stride, int pl_cs, int pl_coff,
                            int pl_cw, int npl, hipStream_t stream) {
  // dx == nullptr with planes given: the gradient is wanted as bf16 planes only (both consumers, the data-gradient and the
  // weight-gradient kernel of the plane mode, read nothing else): one fp32 tensor write less
  HPRI_REQUIRE(dy && x && (dx || planes) && mean && invstd && scale && shift && workspace, "bn_relu_bwd: null pointer");
  PlaneOut po;
  { const int rc_ = hpri_plane_out(&po, planes, plane_stride, pl_cs, pl_coff, pl_cw, npl, C); if (rc_ != HPRI_OK) return rc_; }
  HPRI_REQUIRE(Cw % 4 == 0 && Cw >= C && dy_cs % 4 == 0 && x_cs % 4 == 0 && dx_cs % 4 == 0 && dy_coff % 4 == 0 &&
                   x_coff % 4 == 0 && dx_coff % 4 == 0, "bn_relu_bwd: channel layout must be float4-aligned");
  HPRI_REQUIRE(P > 0 && pix_per_group > 0 && P % pix_per_group == 0 && pix_per_group < (1ll << 31), "bn_relu_bwd: bad pixel counts");
  const int G = (int)(P / pix_per_group);
  int nblk, Cpart;
  hpri_col_reduce_plan(pix_per_group, G, C, &nblk, &Cpart);
  const size_t half = (size_t)G * nblk * 2 * Cpart + (size_t)G * 2 * C;
  if (2 * half > ws_floats) return hpri_set_error(HPRI_ERR_WORKSPACE, "bn_relu_bwd: workspace too small");
  float* part = workspace;
  float* sums = workspace + (size_t)G * nblk * 2 * Cpart;
  float* dxpart = workspace + half;
  float* dxsums = dxpart + (size_t)G * nblk * 2 * Cpart;
  const int c4 = hpri_cdiv(C, 4), cq = pick_cq(c4), ycols = hpri_cdiv(c4, cq);
  // the two reduction sweeps, unless the kernel that produced dy left the partial sums itself (hpri_conv_wino4_bnred)
  const float* fin_part = part; int fin_nblk = nblk, fin_cpart = Cpart;
  if (ext_part != nullptr) {
    HPRI_REQUIRE(G == 1 && ext_nblk > 0 && ext_cpart >= C, "bn_relu_bwd_fused: partial sums need one group and cover the channels");
    fin_part = ext_part; fin_nblk = ext_nblk; fin_cpart = ext_cpart;
    // (the workspace's own partial rows are unused on this path: they hold the folded rows)
    long long S = ((long long)nblk * 2 * Cpart) / (2ll * ext_cpart);
    if (S > 64) S = 64;
    if (S >= 4 && ext_nblk >= 16 * S) {
      hipLaunchKernelGGL(col_fold_kernel, dim3(hpri_cdiv(C, 32), (unsigned)S), dim3(1024), 0, stream, ext_part, ext_nblk, ext_cpart, C, part);
      HPRI_CHECK_LAUNCH();
      fin_part = part; fin_nblk = (int)S;
    }
  } else {
    if (x16 && dy16)
      hipLaunchKernelGGL((col_reduce_kernel<0, true, true>), dim3(nblk, ycols, G), dim3(256), 0, stream, dy, dy_cs, dy_coff, x,
                         x_cs, x_coff, mean, invstd, scale, shift, pix_per_group, C, cq, relu, part, Cpart);
    else if (x16)
      hipLaunchKernelGGL((col_reduce_kernel<0, true>), dim3(nblk, ycols, G), dim3(256), 0, stream, dy, dy_cs, dy_coff, x,
                         x_cs, x_coff, mean, invstd, scale, shift, pix_per_group, C, cq, relu, part, Cpart);
    else
      hipLaunchKernelGGL((col_reduce_kernel<0, false>), dim3(nblk, ycols, G), dim3(256), 0, stream, dy, dy_cs, dy_coff, x,
                         x_cs, x_coff, mean, invstd, scale, shift, pix_per_group, C, cq, relu, part, Cpart);
    HPRI_CHECK_LAUNCH();
  }
  const bool pg = dgamma != nullptr && dbeta != nullptr;
  // The bias of the convolution in front of a TRAINING-mode BatchNorm has an exactly zero gradient: sum_p dx = scale * (sum g -
  // Np * mean(g) - mean(g xhat) * sum xhat) and sum xhat = 0.  The reference holds rounding noise there (~1e-9 of the other
  // gradients; the fixtures compare it against zero); round 1-2 reproduced that noise with a column sum inside the apply kernel
  // and a second finalize launch per layer.  Now: exact zeros, written by the finalize launch that exists anyway (and nothing at
  // all when the caller accumulates).  Eval-mode statistics (use_batch_stats = 0) keep the computed sum: it is not zero there.
  const bool dbias_zero = dbias != nullptr && use_batch_stats;
  float* zero_out = (dbias_zero && !accumulate_dbias && G == 1) ? dbias : nullptr;
  hipLaunchKernelGGL(col_finalize_kernel, dim3(hpri_cdiv(C, 32), G), dim3(1024), 0, stream, fin_part, fin_nblk, fin_cpart, C, sums,
                     (pg && G == 1) ? dbeta : nullptr, (pg && G == 1) ? dgamma : nullptr, accumulate_param_grads, zero_out);
  HPRI_CHECK_LAUNCH();
  if (dbias_zero && !accumulate_dbias && G > 1) {
    if (hipMemsetAsync(dbias, 0, (size_t)C * sizeof(float), stream) != hipSuccess) return hpri_set_error(HPRI_ERR_LAUNCH, "bn_relu_bwd: memset failed");
  }
  if (dbias_zero) dbias = nullptr;          // nothing left to compute for it
  if (pg && G > 1) {
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3(hpri_cdiv(C, 256)), dim3(256), 0, stream, sums, G, C, dgamma, dbeta,
                       accumulate_param_grads);
    HPRI_CHECK_LAUNCH();
  }
  // the apply kernel uses the same (pixel blocks x channel columns x groups) grid as the reduce, so its dx column
  // partials have the reduce's layout; Cw may add one more channel column than C (zero pads)
  const int ycols_w = hpri_cdiv((Cw > po.cw ? Cw : po.cw) >> 2, cq);
  if (x16 && dy16)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true>), dim3(nblk, ycols_w, G), dim3(256), 0, stream, dy, dy_cs, dy_coff, x, x_cs,
                       x_coff, dx, dx_cs, dx_coff, mean, invstd, scale, shift, sums, (int)pix_per_group, C, Cw, cq, relu,
                       use_batch_stats, dbias != nullptr ? dxpart : nullptr, Cpart, po);
  else if (x16)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(nblk, ycols_w, G), dim3(256), 0, stream, dy, dy_cs, dy_coff, x, x_cs,
                       x_coff, dx, dx_cs, dx_coff, mean, invstd, scale, shift, sums, (int)pix_per_group, C, Cw, cq, relu,
                       use_batch_stats, dbias != nullptr ? dxpart : nullptr, Cpart, po);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(nblk, ycols_w, G), dim3(256), 0, stream, dy, dy_cs, dy_coff, x, x_cs,
                       x_coff, dx, dx_cs, dx_coff, mean, invstd, scale, shift, sums, (int)pix_per_group, C, Cw, cq, relu,
                       use_batch_stats, dbias != nullptr ? dxpart : nullptr, Cpart, po);
  HPRI_CHECK_LAUNCH();
  if (dbias != nullptr) {
    hipLaunchKernelGGL(col_finalize_kernel, dim3(hpri_cdiv(C, 32), G), dim3(1024), 0, stream, dxpart, nblk, Cpart, C, dxsums,
                       G == 1 ? dbias : nullptr, (float*)nullptr, accumulate_dbias);
    HPRI_CHECK_LAUNCH();
    if (G > 1) {   // dxsums[g][0][c] = per-group column sums; "dbeta" path of the param-grad kernel adds the groups
      hipLaunchKernelGGL(bn_param_grad_kernel, dim3(hpri_cdiv(C, 256)), dim3(256), 0, stream, dxsums, G, C, nullptr, dbias,
                         accumulate_dbias);
      HPRI_CHECK_LAUNCH();
    }
  }
  return HPRI_OK;
}

extern "C" int hpri_bn_relu_bwd_pl(const float* dy, int dy_cs, int dy_coff, const float* x, int x_cs, int x_coff,
                                   float* dx, int dx_cs, int dx_coff, const float* mean, const float* invstd,
                                   const float* scale, const float* shift, float* dgamma, float* dbeta,
                                   int accumulate_param_grads, float* dbias, int accumulate_dbias, float* workspace,
                                   size_t ws_floats, long long P, long long pix_per_group, int C, int Cw, int relu,
                                   int use_batch_stats, void* planes, long long plane_stride, int pl_cs, int pl_coff,
                                   int pl_cw, int npl, hipStream_t stream) {
  return bn_relu_bwd_impl(nullptr, 0, 0, dy, dy_cs, dy_coff, x, false, x_cs, x_coff, false, dx, dx_cs, dx_coff, mean, invstd, scale, shift, dgamma, dbeta,
                          accumulate_param_grads, dbias, accumulate_dbias, workspace, ws_floats, P, pix_per_group, C, Cw, relu,
                          use_batch_stats, planes, plane_stride, pl_cs, pl_coff, pl_cw, npl, stream);
}

// the same with the reduction already done: `partials`[part_blocks][2][part_cpart] = per-block sums of g*[y>0] and g*[y>0]*xhat
// left by the data-gradient kernel that wrote dy (hpri_conv_wino4_bnred); one group only.  Workspace as hpri_bn_relu_bwd.
extern "C" int hpri_bn_relu_bwd_fused(const float* partials, int part_blocks, int part_cpart, const float* dy, int dy_cs, int dy_coff,
                                      const float* x, int x_cs, int x_coff, float* dx, int dx_cs, int dx_coff, const float* mean,
                                      const float* invstd, const float* scale, const float* shift, float* dgamma, float* dbeta,
                                      int accumulate_param_grads, float* dbias, int accumulate_dbias, float* workspace,
                                      size_t ws_floats, long long P, long long pix_per_group, int C, int Cw, int relu,
                                      int use_batch_stats, void* planes, long long plane_stride, int pl_cs, int pl_coff,
                                      int pl_cw, int npl, hipStream_t stream) {
  HPRI_REQUIRE(partials != nullptr, "bn_relu_bwd_fused: null partial sums");
  return bn_relu_bwd_impl(partials, part_blocks, part_cpart, dy, dy_cs, dy_coff, x, false, x_cs, x_coff, false, dx, dx_cs, dx_coff, mean, invstd,
                          scale, shift, dgamma, dbeta, accumulate_param_grads, dbias, accumulate_dbias, workspace, ws_floats, P,
                          pix_per_group, C, Cw, relu, use_batch_stats, planes, plane_stride, pl_cs, pl_coff, pl_cw, npl, stream);
}

#ifdef HPRI_DIAG_KERNELS   // counterpart of hpri_conv_bf16v3_bnred (diagnostics build only)
// hpri_bn_relu_bwd_fused with the pre-BN tensor stored as bf16 (partial sums from hpri_conv_bf16v3_bnred)
extern "C" int hpri_bn_relu_bwd_fused_x16(const float* partials, int part_blocks, int part_cpart, const float* dy, int dy_cs, int dy_coff,
                                          const void* x16, int x_cs, int x_coff, float* dx, int dx_cs, int dx_coff, const float* mean,
                                          const float* invstd, const float* scale, const float* shift, float* dgamma, float* dbeta,
                                          int accumulate_param_grads, float* dbias, int accumulate_dbias, float* workspace,
                                          size_t ws_floats, long long P, long long pix_per_group, int C, int Cw, int relu,
                                          int use_batch_stats, void* planes, long long plane_stride, int pl_cs, int pl_coff,
                                          int pl_cw, int npl, hipStream_t stream) {
  HPRI_REQUIRE(partials != nullptr, "bn_relu_bwd_fused_x16: null partial sums");
  HPRI_REQUIRE(((uintptr_t)x16 & 7) == 0, "bn_relu_bwd_fused_x16: the bf16 tensor must be 8-byte aligned");
  return bn_relu_bwd_impl(partials, part_blocks, part_cpart, dy, dy_cs, dy_coff, reinterpret_cast<const float*>(x16), true, x_cs, x_coff, false, dx,
                          dx_cs, dx_coff, mean, invstd, scale, shift, dgamma, dbeta, accumulate_param_grads, dbias, accumulate_dbias,
                          workspace, ws_floats, P, pix_per_group, C, Cw, relu, use_batch_stats, planes, plane_stride, pl_cs, pl_coff,
                          pl_cw, npl, stream);
}
#endif   // HPRI_DIAG_KERNELS

// the same with the pre-BN tensor stored as bf16 (see hpri_bn_apply_relu_x16)
extern "C" int hpri_bn_relu_bwd_x16(const float* dy, int dy_cs, int dy_coff, const void* x16, int x_cs, int x_coff,
                                    float* dx, int dx_cs, int dx_coff, const float* mean, const float* invstd,
                                    const float* scale, const float* shift, float* dgamma, float* dbeta,
                                    int accumulate_param_grads, float* dbias, int accumulate_dbias, float* workspace,
                                    size_t ws_floats, long long P, long long pix_per_group, int C, int Cw, int relu,
                                    int use_batch_stats, void* planes, long long plane_stride, int pl_cs, int pl_coff,
                                    int pl_cw, int npl, hipStream_t stream) {
  HPRI_REQUIRE(((uintptr_t)x16 & 7) == 0, "bn_relu_bwd_x16: the bf16 tensor must be 8-byte aligned");
  return bn_relu_bwd_impl(nullptr, 0, 0, dy, dy_cs, dy_coff, reinterpret_cast<const float*>(x16), true, x_cs, x_coff, false, dx, dx_cs, dx_coff, mean,
                          invstd, scale, shift, dgamma, dbeta, accumulate_param_grads, dbias, accumulate_dbias, workspace, ws_floats,
                          P, pix_per_group, C, Cw, relu, use_batch_stats, planes, plane_stride, pl_cs, pl_coff, pl_cw, npl, stream);
}

// the same with the incoming gradient dy ALSO stored as bf16 (dy_cs / dy_coff in elements): the gradient of the inner tensor of a
// DoubleConv in the bf16 mode, written by its only producer (hpri_conv_bf16v3 with the bf16 output bit) and read only here
extern "C" int hpri_bn_relu_bwd_x16_dy16(const void* dy16, int dy_cs, int dy_coff, const void* x16, int x_cs, int x_coff,
                                         float* dx, int dx_cs, int dx_coff, const float* mean, const float* invstd,
                                         const float* scale, const float* shift, float* dgamma, float* dbeta,
                                         int accumulate_param_grads, float* dbias, int accumulate_dbias, float* workspace,
                                         size_t ws_floats, long long P, long long pix_per_group, int C, int Cw, int relu,
                                         int use_batch_stats, void* planes, long long plane_stride, int pl_cs, int pl_coff,
                                         int pl_cw, int npl, hipStream_t stream) {
  HPRI_REQUIRE(((uintptr_t)x16 & 7) == 0 && ((uintptr_t)dy16 & 7) == 0, "bn_relu_bwd_x16_dy16: the bf16 tensors must be 8-byte aligned");
  return bn_relu_bwd_impl(nullptr, 0, 0, reinterpret_cast<const float*>(dy16), dy_cs, dy_coff, reinterpret_cast<const float*>(x16), true, x_cs,
                          x_coff, true, dx, dx_cs, dx_coff, mean, invstd, scale, shift, dgamma, dbeta, accumulate_param_grads, dbias,
                          accumulate_dbias, workspace, ws_floats, P, pix_per_group, C, Cw, relu, use_batch_stats, planes, plane_stride,
                          pl_cs, pl_coff, pl_cw, npl, stream);
}

// out[c] (+)= sum over all P pixels of src[p][coff + c]   (conv / linear bias gradients)
extern "C" int hpri_col_sum(const float* src, int cs, int coff, float* out, int accumulate, float* workspace,
                            size_t ws_floats, long long P, int C, hipStream_t stream) {
  HPRI_REQUIRE(src && out && workspace, "col_sum: null pointer");
  HPRI_REQUIRE(cs % 4 == 0 && coff % 4 == 0 && P > 0 && C > 0, "col_sum: bad layout");
  int nblk, Cpart;
  hpri_col_reduce_plan(P, 1, C, &nblk, &Cpart);
  const size_t need = (size_t)nblk * 2 * Cpart + 2 * (size_t)C;
  if (need > ws_floats) return hpri_set_error(HPRI_ERR_WORKSPACE, "col_sum: workspace too small");
  float* part = workspace;
  float* sums = workspace + (size_t)nblk * 2 * Cpart;
  const int c4 = hpri_cdiv(C, 4), cq = pick_cq(c4);
  // note: reading float4 at channel c..c+3 needs c+3 < cs - coff; callers keep cs a multiple of 4 >= C
  hipLaunchKernelGGL((col_reduce_kernel<1>), dim3(nblk, hpri_cdiv(c4, cq), 1), dim3(256), 0, stream, src, cs, coff, nullptr, 0,
                     0, nullptr, nullptr, nullptr, nullptr, P, C, cq, 0, part, Cpart);
  HPRI_CHECK_LAUNCH();
  // sums[0][c] holds the column sums; the finalize kernel writes (or accumulates) them into `out` on the way
  hipLaunchKernelGGL(col_finalize_kernel, dim3(hpri_cdiv(C, 32), 1), dim3(1024), 0, stream, part, nblk, Cpart, C, sums, out,
                     (float*)nullptr, accumulate);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// ---- column sums from per-tile statistics records --------------------------------------------------------------------
// out[c] (+)= sum over tiles of mean * count of channel c0 + c, from the (mean, M2, count, 0) records a convolution epilogue
// leaves (hpri_conv_wino4 / hpri_conv_bf16v3 / hpri_conv_fwd with `stats`).  Used for the bias gradient of a
// ConvTranspose2d (model_parts.py:63-64): it is the column sum of the gradient of the upsampled half of the concat, which
// the data-gradient kernel of the decoder's first convolution has just written -- its epilogue records replace a
// dedicated pass over that tensor (hpri_col_sum: 0.65 ms per C2 step in the bf16 mode).  One block per 4 channels, 64 tile
// slices per block, double accumulation in a fixed order (deterministic).
__global__ __launch_bounds__(256) void colsum_from_stats_kernel(const float4* __restrict__ stats, int tiles, int Cpad, int c0, int C,
                                                                float* __restrict__ out, int accumulate) {
  __shared__ double red[64][4];
  const int cl = threadIdx.x & 3, sl = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  double a = 0.0;
  if (c < C) {
    // eight records in flight per thread: with one, the loop was a chain of dependent L2 / HBM round trips (184 us per call
    // beside a weight gradient on the second stream; profiles/r03_bf16_mode_bench_kernel_stats.csv)
    int t = sl;
    for (; t + 7 * 64 < tiles; t += 8 * 64) {
      float4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = stats[(size_t)(t + u * 64) * Cpad + c0 + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += (double)r[u].x * (double)r[u].z;
    }
    for (; t < tiles; t += 64) {
      const float4 r = stats[(size_t)t * Cpad + c0 + c];
      a += (double)r.x * (double)r.z;
    }
  }
  red[sl][cl] = a;
  __syncthreads();
  if (sl == 0 && c < C) {
    double tsum = 0.0;
    for (int k = 0; k < 64; ++k) tsum += red[k][cl];
    out[c] = accumulate ? out[c] + (float)tsum : (float)tsum;
  }
}

extern "C" int hpri_colsum_from_stats(const float* stats, int tiles, int Cpad, int c0, int C, float* out, int accumulate,
                                      hipStream_t stream) {
  HPRI_REQUIRE(stats && out && tiles > 0 && C > 0 && c0 >= 0 && c0 + C <= Cpad, "colsum_from_stats: bad arguments");
  HPRI_REQUIRE(((uintptr_t)stats & 15) == 0, "colsum_from_stats: records must be 16-byte aligned");
  hipLaunchKernelGGL(colsum_from_stats_kernel, dim3((unsigned)hpri_cdiv(C, 4)), dim3(256), 0, stream,
                     reinterpret_cast<const float4*>(stats), tiles, Cpad, c0, C, out, accumulate);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}
