// Bandwidth-bound companions of the conv kernels, all on NHWC fp32 with float4 channel vectors:
// layout changes at the module boundary (the reference hands NC(D)HW tensors, dataset.py:267-271),
// MaxPool2d(2) forward/backward (model_parts.py:40), the skip-concat copy and zero padding of
// Up.forward (model_parts.py:77-87), the 1x1 OutConv / final Linear (model_parts.py:96; models.py:103),
// and the counter-based synthetic generator used by bench.py.
#include "common.h"

static inline int ew_blocks(long long total) {
  long long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

// ---------------------------------- NCHW <-> NHWC ----------------------------------------------
// src [N][C][P] -> dst [N][P][cs] at channel offset coff; channels C..Cw are zero-filled.
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, long long P,
                                    int cs, int coff, int Cw) {
  __shared__ float tile[32][65];
  const int n = blockIdx.z;
  const long long p0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  for (int r = ty; r < 32; r += 4) {
    const int c = c0 + r;
    const long long p = p0 + tx;
    tile[r][tx] = (c < C && p < P) ? src[((long long)n * C + c) * P + p] : 0.f;
  }
  __syncthreads();
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;   // 32 x 8
  for (int r = py; r < 64; r += 8) {
    const long long p = p0 + r;
    const int c = c0 + cx;
    if (p < P && c < Cw) dst[((long long)n * P + p) * cs + coff + c] = tile[cx][r];
  }
}

// The same for 16-byte aligned problems (P % 4 == 0, 16-byte aligned bases, cs/coff/Cw multiples of 4), which is what
// the 560 MB cubes are: a 64 channel x 128 pixel tile, float4 reads along the pixels (512-byte runs per channel row),
// float4 writes along the channels (256-byte runs per pixel).
__global__ __launch_bounds__(256) void nchw_to_nhwc_v4_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                              long long P, int cs, int coff, int Cw, PlaneOut pl) {
  __shared__ float tile[128][65];                    // [pixel][channel], pitch 65: conflict-free scalar transposition
  const int n = blockIdx.z;
  const long long p0 = (long long)blockIdx.x * 128;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 pixel quads x 8 channel rows
#pragma unroll
  for (int r = ty; r < 64; r += 8) {
    const int c = c0 + r;
    const long long p = p0 + tx * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < C && p < P) v = *reinterpret_cast<const f32x4*>(src + ((long long)n * C + c) * P + p);
    const int rc = (r + (tx >> 3)) & 63;             // columns rotated by pixel>>5: the 32 lanes hit 32 different banks
    tile[tx * 4 + 0][rc] = v[0]; tile[tx * 4 + 1][rc] = v[1]; tile[tx * 4 + 2][rc] = v[2]; tile[tx * 4 + 3][rc] = v[3];
  }
  __syncthreads();
  const int cq = threadIdx.x & 15, py = threadIdx.x >> 4;   // 16 channel quads x 16 pixels
#pragma unroll
  for (int r = py; r < 128; r += 16) {
    const long long p = p0 + r;
    const int c = c0 + cq * 4;
    if (p < P && (c < Cw || c < pl.cw)) {
      const int ro = r >> 5;
      f32x4 v = {tile[r][(cq * 4 + 0 + ro) & 63], tile[r][(cq * 4 + 1 + ro) & 63], tile[r][(cq * 4 + 2 + ro) & 63],
                 tile[r][(cq * 4 + 3 + ro) & 63]};           // channels >= C were staged as zeros
      if (dst != nullptr && c < Cw) *reinterpret_cast<f32x4*>(dst + ((long long)n * P + p) * cs + coff + c) = v;
      if (c < pl.cw) plane_store4(pl, (size_t)((long long)n * P + p), c, v[0], v[1], v[2], v[3]);
    }
  }
}

// src [N][P][cs]+coff -> dst [N][C][P]
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, long long P,
                                    int cs, int coff, int accumulate) {
  __shared__ float tile[64][33];
  const int n = blockIdx.z;
  const long long p0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 32;
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
  for (int r = py; r < 64; r += 8) {
    const long long p = p0 + r;
    const int c = c0 + cx;
    tile[r][cx] = (p < P && c < C) ? src[((long long)n * P + p) * cs + coff + c] : 0.f;
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 32; r += 4) {
    const int c = c0 + r;
    const long long p = p0 + tx;
    if (c < C && p < P) {
      float* o = dst + ((long long)n * C + c) * P + p;
      *o = accumulate ? *o + tile[tx][r] : tile[tx][r];
    }
  }
}

// ---------------------------------- MaxPool2d(2) ------------------------------------------------
// XB / DXB: the tensor is stored as bf16 rows (plane 0 of a plane buffer; strides and offsets in elements): bf16 mode, round 4 --
// the skip tensors of the U-Nets and their gradients have plane readers only
template <bool B16>
__device__ __forceinline__ float4 ew_load4(const float* __restrict__ p) {
  if (B16) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  return *reinterpret_cast<const float4*>(p);
}
template <bool B16>
__device__ __forceinline__ void ew_store4(float* __restrict__ p, float4 v) {
  if (B16) {
    bf16x4_t h;
    h[0] = (h16_t)v.x; h[1] = (h16_t)v.y; h[2] = (h16_t)v.z; h[3] = (h16_t)v.w;
    *reinterpret_cast<bf16x4_t*>(p) = h;
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}
// element pointer into a tensor of either storage type (the float* carries bf16 elements when B16)
template <bool B16>
__device__ __forceinline__ const float* ew_at(const float* base, long long idx) {
  return B16 ? reinterpret_cast<const float*>(reinterpret_cast<const h16_t*>(base) + idx) : base + idx;
}
template <bool B16>
__device__ __forceinline__ float* ew_at(float* base, long long idx) {
  return B16 ? reinterpret_cast<float*>(reinterpret_cast<h16_t*>(base) + idx) : base + idx;
}

template <bool XB>
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, int x_cs, int x_coff, float* __restrict__ y, int y_cs,
                                    int y_coff, int N, int H, int W, int OH, int OW, int C4v, int C4, PlaneOut pl) {
  // C4v = channel quads with fp32 data, C4 >= C4v = quads covered (the extra ones only zero-fill plane pad channels)
  const long long total = (long long)N * OH * OW * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long r = i / C4;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const int n = (int)(r / OH);
    const size_t opix = (size_t)(((long long)n * OH + oy) * OW + ox);
    if (c >= C4v * 4) { plane_store4(pl, opix, c, 0.f, 0.f, 0.f, 0.f); continue; }
    const long long b = (((long long)n * H + 2 * oy) * W + 2 * ox) * x_cs + x_coff + c;
    const float4 v00 = ew_load4<XB>(ew_at<XB>(x, b));
    const float4 v01 = ew_load4<XB>(ew_at<XB>(x, b + x_cs));
    const float4 v10 = ew_load4<XB>(ew_at<XB>(x, b + (long long)W * x_cs));
    const float4 v11 = ew_load4<XB>(ew_at<XB>(x, b + (long long)W * x_cs + x_cs));
    float4 m;
    m.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x));
    m.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
    m.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z));
    m.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
    if (y != nullptr) *reinterpret_cast<float4*>(y + opix * y_cs + y_coff + c) = m;
    if (pl.p != nullptr) plane_store4(pl, opix, c, m.x, m.y, m.z, m.w);
  }
}

// dx[iy][ix] (+)= dy[iy/2][ix/2] if (iy,ix) is the FIRST maximum of its window in scan order
// (ATen's max_pool2d keeps the first element that compares greater), else 0; rows/cols dropped by the
// floor get 0.
__device__ __forceinline__ int first_argmax4(float a, float b, float c, float d) {
  int k = 0; float m = a;
  if (b > m) { m = b; k = 1; }
  if (c > m) { m = c; k = 2; }
  if (d > m) { k = 3; }
  return k;
}

// One thread per 2x2 WINDOW and channel quad (round 1-3: one per input pixel, each reading all four values of its window -- four
// times the load instructions, three 64-bit divisions per element): the window's four values and the pooled gradient are read once,
// the four results written (or accumulated) from the same thread.  Windows of the last row / column of an odd-sized map have no
// pooled value: their pixels get 0.  32-bit index arithmetic (host: N * ceil(H/2) * ceil(W/2) * C4 < 2^31).
template <bool XB, bool DXB>
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, int x_cs, int x_coff, const float* __restrict__ dy,
                                    int dy_cs, int dy_coff, float* __restrict__ dx, int dx_cs, int dx_coff, int N, int H,
                                    int W, int OH, int OW, int C4, int accumulate) {
  const int WH = (H + 1) >> 1, WW = (W + 1) >> 1;
  const unsigned total = (unsigned)N * WH * WW * C4;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned c = (i % (unsigned)C4) * 4u;
    unsigned r = i / (unsigned)C4;
    const int wx = (int)(r % (unsigned)WW); r /= (unsigned)WW;
    const int wy = (int)(r % (unsigned)WH);
    const int n = (int)(r / (unsigned)WH);
    const int iy = 2 * wy, ix = 2 * wx;
    const bool row1 = iy + 1 < H, col1 = ix + 1 < W, pooled = wy < OH && wx < OW;      // (pooled implies row1 && col1)
    const long long pix = ((long long)n * H + iy) * W + ix;
    float4 o[4] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f),
                   make_float4(0.f, 0.f, 0.f, 0.f)};
    if (pooled) {
      const long long b = pix * x_cs + x_coff + c;
      const float4 v00 = ew_load4<XB>(ew_at<XB>(x, b));
      const float4 v01 = ew_load4<XB>(ew_at<XB>(x, b + x_cs));
      const float4 v10 = ew_load4<XB>(ew_at<XB>(x, b + (long long)W * x_cs));
      const float4 v11 = ew_load4<XB>(ew_at<XB>(x, b + (long long)W * x_cs + x_cs));
      const float4 g = *reinterpret_cast<const float4*>(dy + (((long long)n * OH + wy) * OW + wx) * dy_cs + dy_coff + c);
      const int kx = first_argmax4(v00.x, v01.x, v10.x, v11.x), ky = first_argmax4(v00.y, v01.y, v10.y, v11.y);
      const int kz = first_argmax4(v00.z, v01.z, v10.z, v11.z), kw = first_argmax4(v00.w, v01.w, v10.w, v11.w);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        o[m].x = kx == m ? g.x : 0.f; o[m].y = ky == m ? g.y : 0.f; o[m].z = kz == m ? g.z : 0.f; o[m].w = kw == m ? g.w : 0.f;
      }
    }
    const long long p = pix * dx_cs + dx_coff + c;
    float* pm[4] = {ew_at<DXB>(dx, p), ew_at<DXB>(dx, p + dx_cs), ew_at<DXB>(dx, p + (long long)W * dx_cs),
                    ew_at<DXB>(dx, p + (long long)W * dx_cs + dx_cs)};
    const bool ok[4] = {true, col1, row1, row1 && col1};
    if (accumulate) {
      float4 old[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) old[m] = ok[m] ? ew_load4<DXB>(pm[m]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int m = 0; m < 4; ++m) { o[m].x += old[m].x; o[m].y += old[m].y; o[m].z += old[m].z; o[m].w += old[m].w; }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (ok[m]) ew_store4<DXB>(pm[m], o[m]);
  }
}

// ---------------------------------- slice copy / pad fill / add ---------------------------------
__global__ void copy_slice_kernel(const float* __restrict__ s, int s_cs, int s_coff, float* __restrict__ d, int d_cs,
                                  int d_coff, long long P, int C4, int accumulate) {
  const long long total = P * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / C4;
    const int c = (int)(i - p * C4) * 4;
    float4 v = *reinterpret_cast<const float4*>(s + p * s_cs + s_coff + c);
    float* o = d + p * d_cs + d_coff + c;
    if (accumulate) {
      const float4 w = *reinterpret_cast<const float4*>(o);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    *reinterpret_cast<float4*>(o) = v;
  }
}

// zero channels [coff, coff+4*C4) of every pixel outside the rectangle [y0,y1) x [x0,x1)
__global__ void fill_pad_kernel(float* __restrict__ d, int cs, int coff, int N, int H, int W, int C4, int y0, int y1,
                                int x0, int x1) {
  const long long total = (long long)N * H * W * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long r = i / C4;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    if (iy >= y0 && iy < y1 && ix >= x0 && ix < x1) continue;
    *reinterpret_cast<float4*>(d + (i / C4) * cs + coff + c) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// F.pad with any mix of positive (zero ring) and negative (crop) widths: d[n][y][x] = s[n][y-oy][x-ox] where that lies
// inside the Hs x Ws source, 0 elsewhere (model_parts.py:77-80 when the skip is SMALLER than the upsampled tensor).
__global__ void shift_copy_kernel(const float* __restrict__ s, int s_cs, int s_coff, int Hs, int Ws, float* __restrict__ d,
                                  int d_cs, int d_coff, int N, int Hd, int Wd, int oy, int ox, int C4, int accumulate) {
  const long long total = (long long)N * Hd * Wd * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long r = i / C4;
    const int ix = (int)(r % Wd); r /= Wd;
    const int iy = (int)(r % Hd);
    const int n = (int)(r / Hd);
    const int sy = iy - oy, sx = ix - ox;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws)
      v = *reinterpret_cast<const float4*>(s + (((long long)n * Hs + sy) * Ws + sx) * s_cs + s_coff + c);
    float* o = d + (i / C4) * d_cs + d_coff + c;
    if (accumulate) {
      const float4 w = *reinterpret_cast<const float4*>(o);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    *reinterpret_cast<float4*>(o) = v;
  }
}

__global__ void fill_kernel(float* __restrict__ d, long long n, float v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) d[i] = v;
}

typedef h16_t bf16x8 __attribute__((ext_vector_type(8)));
// ---------------------------------- 1x1 output conv ---------------------------------------------
// logits[n][k][p] = sum_c x[n][p][c] * w[k][c] + b[k]   (NHWC in, NCHW out).  One 16-lane group per pixel.
// BCE-with-logits pieces for the fused head (the loss inside the last layer's kernels; the stand-alone forms
// live in step.hip and compute the same expressions)
__device__ __forceinline__ float oc_bce_elem(float x, float y) { return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float oc_bce_grad(float xi, float yi) {           // (sigmoid(x) - y), as step.hip's bce_bwd_kernel
  const float e = expf(-fabsf(xi)), r = 1.f / (1.f + e);
  const float sp = xi >= 0.f ? r : e * r, sn = xi >= 0.f ? e * r : r;
  return (1.f - yi) * sp - yi * sn;
}

// BCE = true: also the per-block fp64 partial sum of BCEWithLogits(y, target) (nn.BCEWithLogitsLoss of PLTrainer.py:86 on the
// logits this kernel has just produced): one pass over the logits less, the loss finishes with step.hip's finalize kernel
// XB: the source is bf16 NHWC rows (plane 0 of an activation's plane buffer; strides and offsets in elements) instead of fp32
template <bool XB>
__device__ __forceinline__ float4 oc_load4(const float* __restrict__ x, size_t idx) {
  if (XB) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(reinterpret_cast<const h16_t*>(x) + idx);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  return *reinterpret_cast<const float4*>(x + idx);
}

template <bool BCE, bool XB>
__global__ __launch_bounds__(256) void outconv_fwd_kernel(const float* __restrict__ x, int x_cs, int x_coff, const float* __restrict__ w,
                                   const float* __restrict__ b, float* __restrict__ y, int N, long long P, int C, int K,
                                   const float* __restrict__ target, double* __restrict__ partial) {
  __shared__ double bred[256];
  double bsum = 0.0;
  const int gl = threadIdx.x & 15;
  const long long grp = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const long long ngrp = ((long long)gridDim.x * blockDim.x) >> 4;
  const int C4 = (C + 3) >> 2;
  if (!XB && K == 1 && (C & 3) == 0 && C4 <= 64) {
    // one class, up to 256 channels (every head of the reference: model_parts.py:96 with n_classes = 1): the lane's weight quads live
    // in registers, four pixels are in flight per 16-lane group, no per-pixel division.  Same products in the same order as the
    // generic loop below (bit-identical logits): that loop paid four dependent scalar weight loads, each behind a condition, and a
    // 64-bit division per pixel -- 2.0 TB/s on the 301 MB of a full-resolution 64-channel map.
    float4 wq[4];
    int qi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = gl + 16 * j;
      qi[j] = min(q, C4 - 1) * 4;                                    // (lanes beyond the channels read a valid quad and weigh it by zero)
      wq[j] = q < C4 ? *reinterpret_cast<const float4*>(w + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int nj = (C4 + 15) >> 4;                                     // quads per lane (uniform)
    const float b0 = b ? b[0] : 0.f;
    const long long NP = (long long)N * P;
    for (long long pg0 = grp; pg0 < NP; pg0 += 4 * ngrp) {
      float4 v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long pg = min(pg0 + u * ngrp, NP - 1);
        const float* xp = x + pg * x_cs + x_coff;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < nj) v[u][j] = *reinterpret_cast<const float4*>(xp + qi[j]);
      }
      // the butterfly leaves every lane of the group with the pixel's sum: lane u (< 4) keeps pixel u, so the bias, the store and
      // the loss element run once for the four pixels instead of four times behind a one-lane-in-sixteen condition (the exp / log1p
      // of the loss made this kernel VALU-bound: 2.1 TB/s)
      float mine = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < nj) { s += v[u][j].x * wq[j].x; s += v[u][j].y * wq[j].y; s += v[u][j].z * wq[j].z; s += v[u][j].w * wq[j].w; }
        s += __shfl_xor(s, 8, 16); s += __shfl_xor(s, 4, 16); s += __shfl_xor(s, 2, 16); s += __shfl_xor(s, 1, 16);
        if (gl == u) mine = s;
      }
      const long long pgm = pg0 + gl * ngrp;
      if (gl < 4 && pgm < NP) {
        const float o = mine + b0;
        y[pgm] = o;
        if (BCE) bsum += (double)oc_bce_elem(o, target[pgm]);
      }
    }
  } else
  for (long long pg = grp; pg < (long long)N * P; pg += ngrp) {
    for (int k = 0; k < K; ++k) {
      float s = 0.f;
      for (int q = gl; q < C4; q += 16) {
        const float4 v = oc_load4<XB>(x, (size_t)pg * x_cs + x_coff + q * 4);
        const float* wk = w + (long long)k * C + q * 4;
        s += v.x * wk[0];
        if (q * 4 + 1 < C) s += v.y * wk[1];
        if (q * 4 + 2 < C) s += v.z * wk[2];
        if (q * 4 + 3 < C) s += v.w * wk[3];
      }
      s += __shfl_xor(s, 8, 16); s += __shfl_xor(s, 4, 16); s += __shfl_xor(s, 2, 16); s += __shfl_xor(s, 1, 16);
      if (gl == 0) {
        const long long n = pg / P, p = pg - n * P;
        const float v = s + (b ? b[k] : 0.f);
        y[(n * K + k) * P + p] = v;
        if (BCE) bsum += (double)oc_bce_elem(v, target[(n * K + k) * P + p]);
      }
    }
  }
  if (BCE) {
    bred[threadIdx.x] = bsum;
    __syncthreads();
    for (int wd = 128; wd > 0; wd >>= 1) {
      if (threadIdx.x < wd) bred[threadIdx.x] += bred[threadIdx.x + wd];
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = bred[0];
  }
}

// One class over MANY channels (SpectralUNET's Linear(2F, 1), models.py:103: 3300 channels) or over bf16 rows: the weight row sits
// in LDS (zero beyond C), a 16-lane group walks its pixel in 16-byte pieces per lane (4 fp32 / 8 bf16 channels) with four pixels in
// flight.  fp32 sources: lane gl adds the quads gl, gl + 16, ... in that order, then the butterfly -- the sums of the kernel above.
// (That kernel's generic loop on 3300 channels: a 64-bit division per pixel and four conditional scalar weight loads per quad,
// 1.1 TB/s.)
template <bool BCE, bool XB>
__global__ __launch_bounds__(256) void outconv_fwd_wide_kernel(const float* __restrict__ x, int x_cs, int x_coff, const float* __restrict__ w,
                                                               const float* __restrict__ b, float* __restrict__ y, long long NP, int C,
                                                               const float* __restrict__ target, double* __restrict__ partial) {
  extern __shared__ float wl[];                       // C rounded up to a whole round of the 16 lanes
  __shared__ double bred[256];
  constexpr int VEC = XB ? 8 : 4;
  const int Cr = ((C + 16 * VEC - 1) / (16 * VEC)) * (16 * VEC);
  for (int i = threadIdx.x; i < Cr; i += 256) wl[i] = i < C ? w[i] : 0.f;
  __syncthreads();
  const int Cv = ((C + VEC - 1) / VEC) * VEC;         // channels the source holds in whole pieces (its pad channels are zeros)
  double bsum = 0.0;
  const int gl = threadIdx.x & 15;
  const long long grp = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const long long ngrp = ((long long)gridDim.x * blockDim.x) >> 4;
  const float b0 = b ? b[0] : 0.f;
  for (long long pg0 = grp; pg0 < NP; pg0 += 4 * ngrp) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    size_t base[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) base[u] = (size_t)min(pg0 + u * ngrp, NP - 1) * x_cs + x_coff;
    for (int c = gl * VEC; c < Cv; c += 16 * VEC) {
      float4 v[4][VEC / 4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (XB) {
          const bf16x8 t = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const h16_t*>(x) + base[u] + c);
          v[u][0] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
          v[u][VEC / 4 - 1] = make_float4((float)t[4], (float)t[5], (float)t[6], (float)t[7]);
        } else {
          v[u][0] = *reinterpret_cast<const float4*>(x + base[u] + c);
        }
      }
#pragma unroll
      for (int h = 0; h < VEC / 4; ++h) {
        const float4 wv = *reinterpret_cast<const float4*>(wl + c + 4 * h);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s[u] += v[u][h].x * wv.x; s[u] += v[u][h].y * wv.y; s[u] += v[u][h].z * wv.z; s[u] += v[u][h].w * wv.w;
        }
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float t = s[u];
      t += __shfl_xor(t, 8, 16); t += __shfl_xor(t, 4, 16); t += __shfl_xor(t, 2, 16); t += __shfl_xor(t, 1, 16);
      if (gl == u) mine = t;
    }
    const long long pgm = pg0 + gl * ngrp;
    if (gl < 4 && pgm < NP) {
      const float o = mine + b0;
      y[pgm] = o;
      if (BCE) bsum += (double)oc_bce_elem(o, target[pgm]);
    }
  }
  if (BCE) {
    bred[threadIdx.x] = bsum;
    __syncthreads();
    for (int wd = 128; wd > 0; wd >>= 1) {
      if (threadIdx.x < wd) bred[threadIdx.x] += bred[threadIdx.x + wd];
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = bred[0];
  }
}

// dx[n][p][c] = sum_k dy[n][k][p] * w[k][c]  (written, or accumulated into dx)
// BCE = true: dy is not a gradient tensor but the LOGITS; the gradient of the mean BCE-with-logits loss is formed on the fly,
// g = (sigmoid(logit) - target) * gscale[0] / (N*K*P)
template <bool BCE, bool DXB = false>
__global__ void outconv_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                        int dx_cs, int dx_coff, int N, long long P, int C, int Cw, int K, int accumulate,
                                        const float* __restrict__ target, const float* __restrict__ gscale, int CQ, float lscale) {
  // (lscale: the loss scale of the half-precision mode, hpri_set_loss_scale -- 1 otherwise; a power of two: exact)
  const float gs = BCE ? (gscale ? gscale[0] : 1.f) / (float)((double)N * K * P) * lscale : 1.f;
  const int C4 = Cw >> 2;
  const long long total = (long long)N * P * C4;
  if (K == 1) {
    // one class (grid.y = blocks of CQ channel quads, CQ a power of two <= 64): a thread keeps ONE quad of the weight row and walks
    // pixels, four in flight (the generic loop below: a 64-bit division and four conditional scalar weight loads per element)
    const int qd = threadIdx.x & (CQ - 1), rows = blockDim.x / CQ;
    const int c = (blockIdx.y * CQ + qd) * 4;
    const bool live = c < Cw;
    float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c + 3 < C) wv = *reinterpret_cast<const float4*>(w + c);
    else { if (c < C) wv.x = w[c]; if (c + 1 < C) wv.y = w[c + 1]; if (c + 2 < C) wv.z = w[c + 2]; }
    const long long NP = (long long)N * P, step = (long long)gridDim.x * rows;
    for (long long pg0 = (long long)blockIdx.x * rows + threadIdx.x / CQ; pg0 < NP; pg0 += 4 * step) {
      float g[4];
      float4 old[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long pg = min(pg0 + u * step, NP - 1);
        g[u] = dy[pg];
        if (BCE && CQ < 4) g[u] = oc_bce_grad(g[u], target[pg]) * gs;
        old[u] = (accumulate && live) ? ew_load4<DXB>(ew_at<DXB>(dx, pg * dx_cs + dx_coff + c)) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (BCE && CQ >= 4) {
        // the CQ lanes of a pixel would each evaluate the same exp and division: lane u of the group does it for pixel u and hands
        // the result round (the four pixels of a thread belong to its whole group: same threadIdx.x / CQ)
        float lg = 0.f, lt = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (qd == u) { lg = g[u]; lt = target[min(pg0 + u * step, NP - 1)]; }
        const float mine = oc_bce_grad(lg, lt) * gs;
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = __shfl(mine, (threadIdx.x & 63 & ~(CQ - 1)) + u, 64);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long pg = pg0 + u * step;
        if (pg < NP && live) {
          // (0 + g*w, then + old: the generic loop's order)
          float o0 = 0.f + g[u] * wv.x, o1 = 0.f + g[u] * wv.y, o2 = 0.f + g[u] * wv.z, o3 = 0.f + g[u] * wv.w;
          if (accumulate) { o0 += old[u].x; o1 += old[u].y; o2 += old[u].z; o3 += old[u].w; }
          ew_store4<DXB>(ew_at<DXB>(dx, pg * dx_cs + dx_coff + c), make_float4(o0, o1, o2, o3));
        }
      }
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long pg = i / C4;
    const int c = (int)(i - pg * C4) * 4;
    const long long n = pg / P, p = pg - n * P;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
      float g = dy[(n * K + k) * P + p];
      if (BCE) g = oc_bce_grad(g, target[(n * K + k) * P + p]) * gs;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < C) o[j] += g * w[(long long)k * C + c + j];
    }
    float* q = dx + pg * dx_cs + dx_coff + c;
    if (accumulate) {
      const float4 old = *reinterpret_cast<const float4*>(q);
      o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
    }
    *reinterpret_cast<float4*>(q) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// partial[blk][k][c] = sum over the block's pixels of dy[k][p] * x[p][c]; partial_b[blk][k] = sum dy[k][p]
// grid = (nblk, ceil(C4/CQ), K); block = 256 = ROWS x CQ
template <bool BCE, bool XB>
__global__ void outconv_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x, int x_cs, int x_coff,
                                          int N, long long P, int C, int K, int CQ, float* __restrict__ part, int Cpart,
                                          const float* __restrict__ target, const float* __restrict__ gscale, float lscale) {
  const float gs = BCE ? (gscale ? gscale[0] : 1.f) / (float)((double)N * K * P) * lscale : 1.f;
  __shared__ float4 red[256];
  __shared__ float redb[256];
  const int rows = 256 / CQ;
  const int cq = threadIdx.x % CQ, pr = threadIdx.x / CQ;
  const int c = (blockIdx.y * CQ + cq) * 4;
  const int k = blockIdx.z;
  const long long NP = (long long)N * P;
  const long long per = (NP + gridDim.x - 1) / gridDim.x;
  const long long p0 = (long long)blockIdx.x * per;
  const long long p1 = (p0 + per < NP) ? p0 + per : NP;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  float sb = 0.f;
  long long pgs = p0 + pr;
  if (K == 1) {
    // one class: dy is indexed by the pixel itself; four pixels in flight, added in ascending order (the sums of the loop below)
    for (; pgs + 3 * rows < p1; pgs += 4 * rows) {
      float g[4];
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long pg = pgs + u * rows;
        g[u] = dy[pg];
        if (BCE && (CQ < 4 || CQ > 64)) g[u] = oc_bce_grad(g[u], target[pg]) * gs;
        v[u] = c < C ? oc_load4<XB>(x, (size_t)pg * x_cs + x_coff + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (BCE && CQ >= 4 && CQ <= 64) {      // one exp + division per pixel instead of one per lane (outconv_bwd_data_kernel's exchange)
        float lg = 0.f, lt = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (cq == u) { lg = g[u]; lt = target[pgs + u * rows]; }
        const float mine = oc_bce_grad(lg, lt) * gs;
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = __shfl(mine, (threadIdx.x & 63 & ~(CQ - 1)) + u, 64);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sb += g[u];
        if (c < C) { s[0] += g[u] * v[u].x; s[1] += g[u] * v[u].y; s[2] += g[u] * v[u].z; s[3] += g[u] * v[u].w; }
      }
    }
  }
  for (long long pg = pgs; pg < p1; pg += rows) {
    const long long n = pg / P, p = pg - n * P;
    float g = dy[(n * K + k) * P + p];
    if (BCE) g = oc_bce_grad(g, target[(n * K + k) * P + p]) * gs;
    sb += g;
    if (c < C) {
      const float4 v = oc_load4<XB>(x, (size_t)pg * x_cs + x_coff + c);
      s[0] += g * v.x; s[1] += g * v.y; s[2] += g * v.z; s[3] += g * v.w;
    }
  }
  red[threadIdx.x] = make_float4(s[0], s[1], s[2], s[3]);
  redb[threadIdx.x] = sb;
  __syncthreads();
  if (pr == 0) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    float tb = 0.f;
    for (int r = 0; r < rows; ++r) {
      const float4 a = red[r * CQ + cq];
      t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
      tb += redb[r * CQ + cq];
    }
    // layout [blk][K][2][Cpart]: row 0 = weight partials, row 1 (element 0 only) = bias partial
    float* o = part + (((size_t)blockIdx.x * K + k) * 2) * Cpart;
    *reinterpret_cast<float4*>(o + c) = t;
    if (blockIdx.y == 0 && cq == 0) o[Cpart] = tb;
  }
}

// one 256-thread block per output element: dw[k][c] (c < C) or db[k] (c == C); fixed-order tree in LDS
__global__ void outconv_bwd_weight_finalize_kernel(const float* __restrict__ part, int nblk, int K, int Cpart, int C,
                                                   float* __restrict__ dw, float* __restrict__ db, int accumulate) {
  __shared__ double red[256];
  const int idx = blockIdx.x;
  const int k = idx / (C + 1), c = idx - k * (C + 1);
  double s = 0.0;
  const size_t off = (c < C) ? (size_t)c : (size_t)Cpart;      // bias partial lives at row 1, element 0
  for (int b = threadIdx.x; b < nblk; b += 256) s += part[(((size_t)b * K + k) * 2) * Cpart + off];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (c < C) { float* o = dw + (size_t)k * C + c; *o = accumulate ? *o + (float)red[0] : (float)red[0]; }
    else if (db != nullptr) db[k] = accumulate ? db[k] + (float)red[0] : (float)red[0];
  }
}

// ---------------------------------- synthetic generator -----------------------------------------
__device__ __forceinline__ float synth_u(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed * 0x9E3779B97F4A7C15ull + idx;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// mode 0: u ; mode 1: (u > thr) ? 1 : 0 ; mode 2: (2u - 1) * scale
__global__ void synth_kernel(float* __restrict__ d, long long n, unsigned long long seed, int mode, float thr, float scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float u = synth_u(seed, (unsigned long long)i);
    d[i] = mode == 0 ? u : (mode == 1 ? (u > thr ? 1.f : 0.f) : (2.f * u - 1.f) * scale);
  }
}

// ------------------------------------------- C ABI ---------------------------------------------
extern "C" int hpri_nchw_to_nhwc_pl(const float* src, float* dst, int N, int C, long long P, int cs, int coff, int Cw,
                                    void* planes, long long plane_stride, int pl_cs, int pl_coff, int pl_cw, int npl,
                                    hipStream_t stream);
extern "C" int hpri_maxpool2_fwd_pl(const float* x, int x_cs, int x_coff, float* y, int y_cs, int y_coff, int N, int H,
                                    int W, int C, void* planes, long long plane_stride, int pl_cs, int pl_coff, int pl_cw,
                                    int npl, hipStream_t stream);
extern "C" int hpri_nchw_to_nhwc(const float* src, float* dst, int N, int C, long long P, int cs, int coff, int Cw,
                                 hipStream_t stream) {
  return hpri_nchw_to_nhwc_pl(src, dst, N, C, P, cs, coff, Cw, nullptr, 0, 0, 0, 0, 0, stream);
}

// the same, also writing bf16 planes (16-byte aligned problems only: P % 4 == 0 etc.; returns HPRI_ERR_UNSUPPORTED otherwise
// when planes are requested, so the caller can fall back to hpri_to_planes)
extern "C" int hpri_nchw_to_nhwc_pl(const float* src, float* dst, int N, int C, long long P, int cs, int coff, int Cw,
                                    void* planes, long long plane_stride, int pl_cs, int pl_coff, int pl_cw, int npl,
                                    hipStream_t stream) {
  // dst == NULL with planes given: bf16 planes only (the plane-mode first convolution and its weight gradient read nothing else)
  HPRI_REQUIRE(src && (dst || planes) && N > 0 && C > 0 && P > 0 && Cw >= C && Cw + coff <= cs, "nchw_to_nhwc: bad arguments");
  PlaneOut po;
  { const int rc_ = hpri_plane_out(&po, planes, plane_stride, pl_cs, pl_coff, pl_cw, npl, C); if (rc_ != HPRI_OK) return rc_; }
  if (P % 4 == 0 && cs % 4 == 0 && coff % 4 == 0 && Cw % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
    dim3 grid4((unsigned)hpri_cdiv64(P, 128), (unsigned)hpri_cdiv(Cw > po.cw ? Cw : po.cw, 64), (unsigned)N);
    hipLaunchKernelGGL(nchw_to_nhwc_v4_kernel, grid4, dim3(256), 0, stream, src, dst, C, P, cs, coff, Cw, po);
    HPRI_CHECK_LAUNCH();
    return HPRI_OK;
  }
  if (planes != nullptr || dst == nullptr) return hpri_set_error(HPRI_ERR_UNSUPPORTED, "nchw_to_nhwc_pl: plane output needs the 16-byte aligned form");
  dim3 grid((unsigned)hpri_cdiv64(P, 64), (unsigned)hpri_cdiv(Cw, 32), (unsigned)N);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, stream, src, dst, C, P, cs, coff, Cw);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_nhwc_to_nchw(const float* src, float* dst, int N, int C, long long P, int cs, int coff,
                                 int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(src && dst && N > 0 && C > 0 && P > 0 && C + coff <= cs, "nhwc_to_nchw: bad arguments");
  dim3 grid((unsigned)hpri_cdiv64(P, 64), (unsigned)hpri_cdiv(C, 32), (unsigned)N);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, stream, src, dst, C, P, cs, coff, accumulate);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

#define HPRI_REQ_V4(cs, coff) HPRI_REQUIRE((cs) % 4 == 0 && (coff) % 4 == 0, "channel stride/offset must be multiples of 4")

extern "C" int hpri_maxpool2_fwd(const float* x, int x_cs, int x_coff, float* y, int y_cs, int y_coff, int N, int H,
                                 int W, int C, hipStream_t stream) {
  return hpri_maxpool2_fwd_pl(x, x_cs, x_coff, y, y_cs, y_coff, N, H, W, C, nullptr, 0, 0, 0, 0, 0, stream);
}

// MaxPool2d(2), also writing the pooled map as bf16 planes (planes == nullptr: fp32 only)
template <bool XB>
static int maxpool2_fwd_impl(const float* x, int x_cs, int x_coff, float* y, int y_cs, int y_coff, int N, int H,
                             int W, int C, void* planes, long long plane_stride, int pl_cs, int pl_coff, int pl_cw,
                             int npl, hipStream_t stream) {
  HPRI_REQUIRE(x && (y || planes) && N > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, "maxpool2_fwd: bad arguments");
  HPRI_REQ_V4(x_cs, x_coff);
  if (y != nullptr) { HPRI_REQ_V4(y_cs, y_coff); }
  PlaneOut po;
  { const int rc_ = hpri_plane_out(&po, planes, plane_stride, pl_cs, pl_coff, pl_cw, npl, C); if (rc_ != HPRI_OK) return rc_; }
  const int OH = H / 2, OW = W / 2;
  const int c4 = (C > po.cw ? C : po.cw) / 4;
  hipLaunchKernelGGL(maxpool2_fwd_kernel<XB>, dim3(ew_blocks((long long)N * OH * OW * c4)), dim3(256), 0, stream, x, x_cs,
                     x_coff, y, y_cs, y_coff, N, H, W, OH, OW, C / 4, c4, po);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_maxpool2_fwd_pl(const float* x, int x_cs, int x_coff, float* y, int y_cs, int y_coff, int N, int H,
                                    int W, int C, void* planes, long long plane_stride, int pl_cs, int pl_coff, int pl_cw,
                                    int npl, hipStream_t stream) {
  HPRI_REQUIRE(y != nullptr, "maxpool2_fwd: bad arguments");
  return maxpool2_fwd_impl<false>(x, x_cs, x_coff, y, y_cs, y_coff, N, H, W, C, planes, plane_stride, pl_cs, pl_coff, pl_cw, npl, stream);
}

// bf16 mode: MaxPool2d(2) over bf16 rows (plane 0 of the skip tensor's planes: rounding is monotonic, so the pooled bf16 values are
// the ones the fp32 form would have written as planes); y (fp32) is optional.
extern "C" int hpri_maxpool2_fwd_x16(const void* x16, int x_cs, int x_coff, float* y, int y_cs, int y_coff, int N, int H,
                                     int W, int C, void* planes, long long plane_stride, int pl_cs, int pl_coff, int pl_cw,
                                     int npl, hipStream_t stream) {
  return maxpool2_fwd_impl<true>(reinterpret_cast<const float*>(x16), x_cs, x_coff, y, y_cs, y_coff, N, H, W, C, planes, plane_stride,
                                 pl_cs, pl_coff, pl_cw, npl, stream);
}

extern "C" int hpri_maxpool2_bwd(const float* x, int x_cs, int x_coff, const float* dy, int dy_cs, int dy_coff, float* dx,
                                 int dx_cs, int dx_coff, int N, int H, int W, int C, int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(x && dy && dx && N > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, "maxpool2_bwd: bad arguments");
  HPRI_REQ_V4(x_cs, x_coff); HPRI_REQ_V4(dy_cs, dy_coff); HPRI_REQ_V4(dx_cs, dx_coff);
  const long long windows = (long long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  HPRI_REQUIRE(windows < (1ll << 31), "maxpool2_bwd: more than 2^31 window quads");
  hipLaunchKernelGGL((maxpool2_bwd_kernel<false, false>), dim3(ew_blocks(windows)), dim3(256), 0, stream, x, x_cs,
                     x_coff, dy, dy_cs, dy_coff, dx, dx_cs, dx_coff, N, H, W, H / 2, W / 2, C / 4, accumulate);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// bf16 mode: the pooled tensor's input is read as bf16 rows (x_bf16; the first maximum among the ROUNDED values takes the gradient),
// and / or the input gradient is stored as bf16 rows (dx_bf16: written, or read - added - rounded).  dy is fp32.
extern "C" int hpri_maxpool2_bwd_x16(const void* x, int x_bf16, int x_cs, int x_coff, const float* dy, int dy_cs, int dy_coff, void* dx,
                                     int dx_bf16, int dx_cs, int dx_coff, int N, int H, int W, int C, int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(x && dy && dx && N > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, "maxpool2_bwd_x16: bad arguments");
  HPRI_REQ_V4(x_cs, x_coff); HPRI_REQ_V4(dy_cs, dy_coff); HPRI_REQ_V4(dx_cs, dx_coff);
  const long long windows = (long long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  HPRI_REQUIRE(windows < (1ll << 31), "maxpool2_bwd_x16: more than 2^31 window quads");
  const float* xf = reinterpret_cast<const float*>(x);
  float* dxf = reinterpret_cast<float*>(dx);
#define HPRI_MP_BWD(XB_, DXB_)                                                                                              \
  hipLaunchKernelGGL((maxpool2_bwd_kernel<XB_, DXB_>), dim3(ew_blocks(windows)), dim3(256), 0, stream, xf, x_cs, x_coff, dy, \
                     dy_cs, dy_coff, dxf, dx_cs, dx_coff, N, H, W, H / 2, W / 2, C / 4, accumulate)
  if (x_bf16 && dx_bf16) HPRI_MP_BWD(true, true);
  else if (x_bf16) HPRI_MP_BWD(true, false);
  else if (dx_bf16) HPRI_MP_BWD(false, true);
  else HPRI_MP_BWD(false, false);
#undef HPRI_MP_BWD
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_copy_slice(const float* src, int s_cs, int s_coff, float* dst, int d_cs, int d_coff, long long P,
                               int C, int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(src && dst && P > 0 && C > 0 && C % 4 == 0, "copy_slice: bad arguments");
  HPRI_REQ_V4(s_cs, s_coff); HPRI_REQ_V4(d_cs, d_coff);
  hipLaunchKernelGGL(copy_slice_kernel, dim3(ew_blocks(P * (C / 4))), dim3(256), 0, stream, src, s_cs, s_coff, dst, d_cs,
                     d_coff, P, C / 4, accumulate);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_shift_copy(const float* src, int s_cs, int s_coff, int Hs, int Ws, float* dst, int d_cs, int d_coff, int N,
                               int Hd, int Wd, int oy, int ox, int C, int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(src && dst && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0 && C % 4 == 0, "shift_copy: bad arguments");
  HPRI_REQ_V4(s_cs, s_coff); HPRI_REQ_V4(d_cs, d_coff);
  hipLaunchKernelGGL(shift_copy_kernel, dim3(ew_blocks((long long)N * Hd * Wd * (C / 4))), dim3(256), 0, stream, src, s_cs,
                     s_coff, Hs, Ws, dst, d_cs, d_coff, N, Hd, Wd, oy, ox, C / 4, accumulate);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_fill_pad(float* dst, int cs, int coff, int N, int H, int W, int C, int y0, int y1, int x0, int x1,
                             hipStream_t stream) {
  HPRI_REQUIRE(dst && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "fill_pad: bad arguments");
  HPRI_REQ_V4(cs, coff);
  hipLaunchKernelGGL(fill_pad_kernel, dim3(ew_blocks((long long)N * H * W * (C / 4))), dim3(256), 0, stream, dst, cs, coff, N,
                     H, W, C / 4, y0, y1, x0, x1);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_fill(float* dst, long long n, float value, hipStream_t stream) {
  HPRI_REQUIRE(dst && n > 0, "fill: bad arguments");
  hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, dst, n, value);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// forward of the head over fp32 rows (XB = false) or bf16 rows (XB = true); target != nullptr: with the loss partials
template <bool XB>
static int outconv_fwd_impl(const float* x, int x_cs, int x_coff, const float* w, const float* b, float* y, const float* target,
                            double* partial, size_t partial_doubles, int N, long long P, int C, int K, hipStream_t stream) {
  constexpr int VEC = XB ? 8 : 4;
  HPRI_REQUIRE(x && w && y && N > 0 && P > 0 && C > 0 && K > 0, "outconv_fwd: bad arguments");
  HPRI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0, "outconv_fwd: channel stride / offset must be multiples of 4");
  HPRI_REQUIRE(((C + 3) / 4) * 4 + x_coff <= x_cs, "outconv_fwd: channel stride too small for 4-channel reads");
  const int nb = ew_blocks((long long)N * P * 16);
  if (target != nullptr && (size_t)nb > partial_doubles) return hpri_set_error(HPRI_ERR_WORKSPACE, "outconv_fwd_bce: partial buffer too small");
  const int Cr = ((C + 16 * VEC - 1) / (16 * VEC)) * (16 * VEC);
  const bool wide = K == 1 && (XB || C > 256 || (C & 3) != 0) && Cr <= 12288 && x_cs % VEC == 0 && x_coff % VEC == 0 &&
                    ((C + VEC - 1) / VEC) * VEC + x_coff <= x_cs;
  if (wide) {
    if (target != nullptr)
      hipLaunchKernelGGL((outconv_fwd_wide_kernel<true, XB>), dim3(nb), dim3(256), Cr * sizeof(float), stream, x, x_cs, x_coff, w, b, y,
                         (long long)N * P, C, target, partial);
    else
      hipLaunchKernelGGL((outconv_fwd_wide_kernel<false, XB>), dim3(nb), dim3(256), Cr * sizeof(float), stream, x, x_cs, x_coff, w, b, y,
                         (long long)N * P, C, (const float*)nullptr, (double*)nullptr);
  } else if (target != nullptr) {
    hipLaunchKernelGGL((outconv_fwd_kernel<true, XB>), dim3(nb), dim3(256), 0, stream, x, x_cs, x_coff, w, b, y, N, P, C, K, target, partial);
  } else {
    hipLaunchKernelGGL((outconv_fwd_kernel<false, XB>), dim3(nb), dim3(256), 0, stream, x, x_cs, x_coff, w, b, y, N, P, C, K,
                       (const float*)nullptr, (double*)nullptr);
  }
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_outconv_fwd(const float* x, int x_cs, int x_coff, const float* w, const float* b, float* y, int N,
                                long long P, int C, int K, hipStream_t stream) {
  return outconv_fwd_impl<false>(x, x_cs, x_coff, w, b, y, nullptr, nullptr, 0, N, P, C, K, stream);
}

// The same head with nn.BCEWithLogitsLoss() (mean) of its logits against `target` (N, K, P) computed on the way: `partial` receives
// one fp64 partial sum per block (hpri_outconv_fwd_bce_blocks of them); hpri_bce_finish turns them into the loss.
extern "C" size_t hpri_outconv_fwd_bce_blocks(int N, long long P) { return (size_t)ew_blocks((long long)N * P * 16); }
extern "C" int hpri_outconv_fwd_bce(const float* x, int x_cs, int x_coff, const float* w, const float* b, float* y, const float* target,
                                    double* partial, size_t partial_doubles, int N, long long P, int C, int K, hipStream_t stream) {
  HPRI_REQUIRE(target && partial, "outconv_fwd_bce: bad arguments");
  return outconv_fwd_impl<false>(x, x_cs, x_coff, w, b, y, target, partial, partial_doubles, N, P, C, K, stream);
}

// ... over bf16 rows: the input is plane 0 of an activation's plane buffer (bf16 NHWC, x_cs / x_coff in elements, pad channels zero):
// the head of the bf16 mode reads what the last BatchNorm pass wrote for it, and no fp32 copy of that tensor exists.
// target == nullptr: logits only.
extern "C" int hpri_outconv_fwd_x16(const void* x16, int x_cs, int x_coff, const float* w, const float* b, float* y, const float* target,
                                    double* partial, size_t partial_doubles, int N, long long P, int C, int K, hipStream_t stream) {
  HPRI_REQUIRE(target == nullptr || partial != nullptr, "outconv_fwd_x16: bad arguments");
  return outconv_fwd_impl<true>(reinterpret_cast<const float*>(x16), x_cs, x_coff, w, b, y, target, partial, partial_doubles, N, P, C, K, stream);
}

static inline int pick_cq(int c4) { int q = 1; while (q < c4 && q < 64) q <<= 1; return q; }

extern "C" int hpri_outconv_bwd_plan(int N, long long P, int C, int K, int* nblk, int* Cpart) {
  const int c4 = hpri_cdiv(C, 4), cq = pick_cq(c4), rows = 256 / cq, ycols = hpri_cdiv(c4, cq);
  long long nb = 1024 / ((long long)ycols * K);
  const long long maxb = ((long long)N * P + rows * 8 - 1) / (rows * 8);
  if (nb > maxb) nb = maxb;
  if (nb < 1) nb = 1;
  *nblk = (int)nb; *Cpart = ycols * cq * 4;
  return HPRI_OK;
}

// dx (optional), dw, db of the 1x1 output conv.  workspace: nblk*K*2*Cpart floats (hpri_outconv_bwd_plan)
template <bool XB>
static int outconv_bwd_impl(const float* dy, const float* target, const float* gscale, const float* x, int x_cs, int x_coff,
                            const float* w, float* dx, int dx_cs, int dx_coff, int dx_cw, int dx_accumulate, float* dw, float* db,
                            int accumulate_param_grads, float* workspace, size_t ws_floats, int N, long long P, int C,
                            int K, hipStream_t stream, int dx_bf16 = 0) {
  const bool bce = target != nullptr;
  HPRI_REQUIRE(dy && x && w && dw && workspace && N > 0 && P > 0 && C > 0 && K > 0, "outconv_bwd: bad arguments");
  HPRI_REQ_V4(x_cs, x_coff);
  if (dx != nullptr) {
    HPRI_REQ_V4(dx_cs, dx_coff);
    HPRI_REQUIRE(dx_cw % 4 == 0 && dx_cw >= C && dx_cw + dx_coff <= dx_cs, "outconv_bwd: dx channel layout");
    dim3 grid(ew_blocks((long long)N * P * (dx_cw / 4)));
    int dq = 1;
    if (K == 1) {                          // (nbx, blocks of dq channel quads)
      dq = pick_cq(dx_cw / 4);
      const int ycols = hpri_cdiv(dx_cw / 4, dq), rows = 256 / dq;
      long long nbx = hpri_cdiv((long long)N * P, (long long)rows * 4);
      const long long cap = 8192 / ycols > 0 ? 8192 / ycols : 1;
      if (nbx > cap) nbx = cap;
      grid = dim3((unsigned)nbx, ycols);
    }
    HPRI_REQUIRE(!dx_bf16 || K == 1, "outconv_bwd: a bf16 input gradient is built for one class");
    if (dx_bf16 && bce)
      hipLaunchKernelGGL((outconv_bwd_data_kernel<true, true>), grid, dim3(256), 0, stream, dy, w,
                         dx, dx_cs, dx_coff, N, P, C, dx_cw, K, dx_accumulate, target, gscale, dq, hpri_loss_scale());
    else if (dx_bf16)
      hipLaunchKernelGGL((outconv_bwd_data_kernel<false, true>), grid, dim3(256), 0, stream, dy, w,
                         dx, dx_cs, dx_coff, N, P, C, dx_cw, K, dx_accumulate, (const float*)nullptr, (const float*)nullptr, dq, hpri_loss_scale());
    else if (bce)
      hipLaunchKernelGGL(outconv_bwd_data_kernel<true>, grid, dim3(256), 0, stream, dy, w,
                         dx, dx_cs, dx_coff, N, P, C, dx_cw, K, dx_accumulate, target, gscale, dq, hpri_loss_scale());
    else
      hipLaunchKernelGGL(outconv_bwd_data_kernel<false>, grid, dim3(256), 0, stream, dy, w,
                         dx, dx_cs, dx_coff, N, P, C, dx_cw, K, dx_accumulate, (const float*)nullptr, (const float*)nullptr, dq, hpri_loss_scale());
    HPRI_CHECK_LAUNCH();
  }
  int nblk, Cpart;
  hpri_outconv_bwd_plan(N, P, C, K, &nblk, &Cpart);
  if ((size_t)nblk * K * 2 * Cpart > ws_floats) return hpri_set_error(HPRI_ERR_WORKSPACE, "outconv_bwd: workspace too small");
  const int c4 = hpri_cdiv(C, 4), cq = pick_cq(c4);
  if (bce)
    hipLaunchKernelGGL((outconv_bwd_weight_kernel<true, XB>), dim3(nblk, hpri_cdiv(c4, cq), K), dim3(256), 0, stream, dy, x, x_cs, x_coff,
                       N, P, C, K, cq, workspace, Cpart, target, gscale, hpri_loss_scale());
  else
    hipLaunchKernelGGL((outconv_bwd_weight_kernel<false, XB>), dim3(nblk, hpri_cdiv(c4, cq), K), dim3(256), 0, stream, dy, x, x_cs, x_coff,
                       N, P, C, K, cq, workspace, Cpart, (const float*)nullptr, (const float*)nullptr, hpri_loss_scale());
  HPRI_CHECK_LAUNCH();
  hipLaunchKernelGGL(outconv_bwd_weight_finalize_kernel, dim3(K * (C + 1)), dim3(256), 0, stream, workspace,
                     nblk, K, Cpart, C, dw, db, accumulate_param_grads);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_outconv_bwd(const float* dy, const float* x, int x_cs, int x_coff, const float* w, float* dx,
                                int dx_cs, int dx_coff, int dx_cw, int dx_accumulate, float* dw, float* db,
                                int accumulate_param_grads, float* workspace, size_t ws_floats, int N, long long P, int C,
                                int K, hipStream_t stream) {
  return outconv_bwd_impl<false>(dy, nullptr, nullptr, x, x_cs, x_coff, w, dx, dx_cs, dx_coff, dx_cw, dx_accumulate, dw, db,
                          accumulate_param_grads, workspace, ws_floats, N, P, C, K, stream);
}

// Backward of the fused head + loss: the gradient of mean BCE-with-logits w.r.t. the logits, (sigmoid(logits) - target) *
// gscale[0] / (N*K*P) (gscale: device scalar, the gradient arriving at the loss; nullptr = 1), is formed inside the two kernels
// instead of being written out by a separate pass and read twice.
extern "C" int hpri_outconv_bwd_bce(const float* logits, const float* target, const float* gscale, const float* x, int x_cs, int x_coff,
                                    const float* w, float* dx, int dx_cs, int dx_coff, int dx_cw, int dx_accumulate, float* dw,
                                    float* db, int accumulate_param_grads, float* workspace, size_t ws_floats, int N, long long P,
                                    int C, int K, hipStream_t stream) {
  HPRI_REQUIRE(target != nullptr, "outconv_bwd_bce: null target");
  return outconv_bwd_impl<false>(logits, target, gscale, x, x_cs, x_coff, w, dx, dx_cs, dx_coff, dx_cw, dx_accumulate, dw, db,
                                 accumulate_param_grads, workspace, ws_floats, N, P, C, K, stream);
}

// Backward of the head over bf16 rows (hpri_outconv_fwd_x16): dw / db read x16; dx (optional) as hpri_outconv_bwd, or -- dx_bf16, one
// class -- as bf16 rows (strides in elements): its one reader, the last BatchNorm backward, reads bf16.
// target != nullptr: `dy` holds the LOGITS and the loss gradient is formed inside the kernels (hpri_outconv_bwd_bce).
extern "C" int hpri_outconv_bwd_x16(const float* dy, const float* target, const float* gscale, const void* x16, int x_cs, int x_coff,
                                    const float* w, void* dx, int dx_bf16, int dx_cs, int dx_coff, int dx_cw, int dx_accumulate, float* dw,
                                    float* db, int accumulate_param_grads, float* workspace, size_t ws_floats, int N, long long P,
                                    int C, int K, hipStream_t stream) {
  return outconv_bwd_impl<true>(dy, target, gscale, reinterpret_cast<const float*>(x16), x_cs, x_coff, w, reinterpret_cast<float*>(dx),
                                dx_cs, dx_coff, dx_cw, dx_accumulate, dw, db, accumulate_param_grads, workspace, ws_floats, N, P, C, K,
                                stream, dx_bf16);
}

extern "C" int hpri_synth_fill(float* dst, long long n, unsigned long long seed, int mode, float thr, float scale,
                               hipStream_t stream) {
  HPRI_REQUIRE(dst && n > 0 && mode >= 0 && mode <= 2, "synth_fill: bad arguments");
  hipLaunchKernelGGL(synth_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, dst, n, seed, mode, thr, scale);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// ---------------------------------- bilinear x2 upsample (align_corners=True) ---------------------
// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True), model_parts.py:57 / models.py:195.
// ATen semantics: src = dst * (in-1)/(out-1); i0 = (int)src; l1 = src - i0; i1 = i0 + (i0 < in-1).
// The result is written at pixel offset (py0, px0) of a (possibly larger, padded) destination image.
__device__ __forceinline__ void bil_coord(int o, int in, int out, int* i0, int* i1, float* l0, float* l1) {
  const float scale = (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float src = scale * (float)o;
  const int a = (int)src;
  *i0 = a; *i1 = a + ((a < in - 1) ? 1 : 0);
  *l1 = src - (float)a; *l0 = 1.f - *l1;
}

__global__ void upsample2x_fwd_kernel(const float* __restrict__ x, int x_cs, int x_coff, float* __restrict__ y, int y_cs,
                                      int y_coff, int N, int H, int W, int H2, int W2, int py0, int px0, int C4) {
  const int OH = 2 * H, OW = 2 * W;
  const long long total = (long long)N * OH * OW * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long r = i / C4;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const int n = (int)(r / OH);
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    bil_coord(oy, H, OH, &y0, &y1, &ly0, &ly1);
    bil_coord(ox, W, OW, &x0, &x1, &lx0, &lx1);
    const float* b = x + (long long)n * H * W * x_cs + x_coff + c;
    const float4 v00 = *reinterpret_cast<const float4*>(b + ((long long)y0 * W + x0) * x_cs);
    const float4 v01 = *reinterpret_cast<const float4*>(b + ((long long)y0 * W + x1) * x_cs);
    const float4 v10 = *reinterpret_cast<const float4*>(b + ((long long)y1 * W + x0) * x_cs);
    const float4 v11 = *reinterpret_cast<const float4*>(b + ((long long)y1 * W + x1) * x_cs);
    float4 o;
    o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
    o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
    o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
    o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
    *reinterpret_cast<float4*>(y + (((long long)n * H2 + oy + py0) * W2 + ox + px0) * y_cs + y_coff + c) = o;
  }
}

// gradient w.r.t. the low-res input as a GATHER (deterministic): input pixel (iy,ix) collects from the <= 4x4
// output pixels whose interpolation footprint contains it.
__global__ void upsample2x_bwd_kernel(const float* __restrict__ dy, int dy_cs, int dy_coff, float* __restrict__ dx,
                                      int dx_cs, int dx_coff, int N, int H, int W, int H2, int W2, int py0, int px0,
                                      int C4, int accumulate) {
  const int OH = 2 * H, OW = 2 * W;
  const long long total = (long long)N * H * W * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long long r = i / C4;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const int n = (int)(r / H);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int oy_lo = max(0, 2 * iy - 3), oy_hi = min(OH - 1, 2 * iy + 3);
    const int ox_lo = max(0, 2 * ix - 3), ox_hi = min(OW - 1, 2 * ix + 3);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float ly0, ly1;
      bil_coord(oy, H, OH, &y0, &y1, &ly0, &ly1);
      float wy = 0.f;
      if (y0 == iy) wy += ly0;
      if (y1 == iy) wy += ly1;
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float lx0, lx1;
        bil_coord(ox, W, OW, &x0, &x1, &lx0, &lx1);
        float wx = 0.f;
        if (x0 == ix) wx += lx0;
        if (x1 == ix) wx += lx1;
        if (wx == 0.f) continue;
        const float4 g = *reinterpret_cast<const float4*>(dy + (((long long)n * H2 + oy + py0) * W2 + ox + px0) * dy_cs + dy_coff + c);
        const float w = wy * wx;
        acc[0] += w * g.x; acc[1] += w * g.y; acc[2] += w * g.z; acc[3] += w * g.w;
      }
    }
    float* p = dx + (((long long)n * H + iy) * W + ix) * dx_cs + dx_coff + c;
    if (accumulate) {
      const float4 old = *reinterpret_cast<const float4*>(p);
      acc[0] += old.x; acc[1] += old.y; acc[2] += old.z; acc[3] += old.w;
    }
    *reinterpret_cast<float4*>(p) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// out = a * b (the reference's "attention", model_parts.py:84-85), float4 over channels; accumulate adds into out
__global__ void mul_kernel(const float* __restrict__ a, int a_cs, int a_coff, const float* __restrict__ b, int b_cs,
                           int b_coff, float* __restrict__ o, int o_cs, int o_coff, long long P, int C4, int accumulate) {
  const long long total = P * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / C4;
    const int c = (int)(i - p * C4) * 4;
    const float4 u = *reinterpret_cast<const float4*>(a + p * a_cs + a_coff + c);
    const float4 v = *reinterpret_cast<const float4*>(b + p * b_cs + b_coff + c);
    float4 w = make_float4(u.x * v.x, u.y * v.y, u.z * v.z, u.w * v.w);
    float* q = o + p * o_cs + o_coff + c;
    if (accumulate) { const float4 z = *reinterpret_cast<const float4*>(q); w.x += z.x; w.y += z.y; w.z += z.z; w.w += z.w; }
    *reinterpret_cast<float4*>(q) = w;
  }
}

extern "C" int hpri_upsample2x_fwd(const float* x, int x_cs, int x_coff, float* y, int y_cs, int y_coff, int N, int H,
                                   int W, int H2, int W2, int py0, int px0, int C, hipStream_t stream) {
  HPRI_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample2x_fwd: bad arguments");
  HPRI_REQ_V4(x_cs, x_coff); HPRI_REQ_V4(y_cs, y_coff);
  HPRI_REQUIRE(py0 >= 0 && px0 >= 0 && 2 * H + py0 <= H2 && 2 * W + px0 <= W2, "upsample2x_fwd: output exceeds the destination image");
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(ew_blocks((long long)N * 4 * H * W * (C / 4))), dim3(256), 0, stream, x, x_cs,
                     x_coff, y, y_cs, y_coff, N, H, W, H2, W2, py0, px0, C / 4);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_upsample2x_bwd(const float* dy, int dy_cs, int dy_coff, float* dx, int dx_cs, int dx_coff, int N,
                                   int H, int W, int H2, int W2, int py0, int px0, int C, int accumulate,
                                   hipStream_t stream) {
  HPRI_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample2x_bwd: bad arguments");
  HPRI_REQ_V4(dy_cs, dy_coff); HPRI_REQ_V4(dx_cs, dx_coff);
  HPRI_REQUIRE(py0 >= 0 && px0 >= 0 && 2 * H + py0 <= H2 && 2 * W + px0 <= W2, "upsample2x_bwd: region exceeds the source image");
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(ew_blocks((long long)N * H * W * (C / 4))), dim3(256), 0, stream, dy, dy_cs,
                     dy_coff, dx, dx_cs, dx_coff, N, H, W, H2, W2, py0, px0, C / 4, accumulate);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_mul(const float* a, int a_cs, int a_coff, const float* b, int b_cs, int b_coff, float* out, int o_cs,
                        int o_coff, long long P, int C, int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(a && b && out && P > 0 && C > 0 && C % 4 == 0, "mul: bad arguments");
  HPRI_REQ_V4(a_cs, a_coff); HPRI_REQ_V4(b_cs, b_coff); HPRI_REQ_V4(o_cs, o_coff);
  hipLaunchKernelGGL(mul_kernel, dim3(ew_blocks(P * (C / 4))), dim3(256), 0, stream, a, a_cs, a_coff, b, b_cs, b_coff, out, o_cs,
                     o_coff, P, C / 4, accumulate);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// element-granular slice copy for channel offsets / counts that are not multiples of 4 (SpectralUNET's
// F = 1650 concat, models.py:139-143); channels [C, Cz) of the destination are zero-filled
__global__ void copy_slice_any_kernel(const float* __restrict__ s, int s_cs, int s_coff, float* __restrict__ d, int d_cs,
                                      int d_coff, long long P, int C, int Cz, int accumulate) {
  const int per = Cz > C ? Cz : C;
  const long long total = P * per;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / per;
    const int c = (int)(i - p * per);
    float* o = d + p * d_cs + d_coff + c;
    if (c < C) {
      const float v = s[p * s_cs + s_coff + c];
      *o = accumulate ? *o + v : v;
    } else if (!accumulate) {
      *o = 0.f;
    }
  }
}

// the same for wide rows (SpectralUNET's 1650-channel halves): a block walks whole pixel rows, its threads the row's channels in
// pieces of V floats (V = 2 when every stride, offset and count is even: 1650 is) -- no division per element, 8-byte accesses.
// (The element loop above: 3.1 TB/s on the 2.8 GB halves of config C3.)
template <int V>
__global__ void copy_slice_rows_kernel(const float* __restrict__ s, int s_cs, int s_coff, float* __restrict__ d, int d_cs,
                                       int d_coff, long long P, int C, int Cz, int accumulate) {
  typedef float vec_t __attribute__((ext_vector_type(V)));
  const int per = Cz > C ? Cz : C;
  for (long long p = blockIdx.x; p < P; p += gridDim.x) {
    const float* sp = s + p * s_cs + s_coff;
    float* dp = d + p * d_cs + d_coff;
    for (int c = threadIdx.x * V; c < per; c += 256 * V) {
      if (c < C) {
        vec_t v = *reinterpret_cast<const vec_t*>(sp + c);
        if (accumulate) v += *reinterpret_cast<const vec_t*>(dp + c);
        *reinterpret_cast<vec_t*>(dp + c) = v;
      } else if (!accumulate) {
        vec_t z = {};
        *reinterpret_cast<vec_t*>(dp + c) = z;
      }
    }
  }
}

extern "C" int hpri_copy_slice_any(const float* src, int s_cs, int s_coff, float* dst, int d_cs, int d_coff, long long P,
                                   int C, int Cz, int accumulate, hipStream_t stream) {
  HPRI_REQUIRE(src && dst && P > 0 && C > 0 && C + s_coff <= s_cs && (Cz > C ? Cz : C) + d_coff <= d_cs, "copy_slice_any: bad arguments");
  const int per = Cz > C ? Cz : C;
  if (per >= 256) {
    const bool even = ((s_cs | s_coff | d_cs | d_coff | C | per) & 1) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 7) == 0;
    const unsigned nb = (unsigned)(P < 16384 ? P : 16384);
    if (even)
      hipLaunchKernelGGL(copy_slice_rows_kernel<2>, dim3(nb), dim3(256), 0, stream, src, s_cs, s_coff, dst, d_cs, d_coff, P, C, Cz, accumulate);
    else
      hipLaunchKernelGGL(copy_slice_rows_kernel<1>, dim3(nb), dim3(256), 0, stream, src, s_cs, s_coff, dst, d_cs, d_coff, P, C, Cz, accumulate);
    HPRI_CHECK_LAUNCH();
    return HPRI_OK;
  }
  hipLaunchKernelGGL(copy_slice_any_kernel, dim3(ew_blocks(P * (Cz > C ? Cz : C))), dim3(256), 0, stream, src, s_cs, s_coff, dst,
                     d_cs, d_coff, P, C, Cz, accumulate);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// ---- fp32 NHWC view -> bf16 planes (generic producer; fused producers write planes themselves) ---------------------------
// planes[p][pixel][cs16]: plane 0 = bf16(x), plane 1 = bf16(x - hi), plane 2 = bf16(x - hi - mid); channels [C, cw16) = 0
__global__ void to_planes_kernel(const float* __restrict__ x, int cs, int coff, h16_t* __restrict__ pl, long long plane,
                                 int cs16, int coff16, long long P, int C, int cw16, int npl) {
  const int q8 = cw16 >> 3;
  const long long total = P * q8;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long p = idx / q8;
    const int c = (int)(idx - p * q8) * 8;
    float v[8];
    const float* src = x + p * cs + coff + c;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (c + h * 4 + 3 < C) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + h * 4);
        v[h * 4 + 0] = t[0]; v[h * 4 + 1] = t[1]; v[h * 4 + 2] = t[2]; v[h * 4 + 3] = t[3];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[h * 4 + e] = (c + h * 4 + e < C) ? src[h * 4 + e] : 0.f;
      }
    }
    for (int k = 0; k < npl; ++k) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const h16_t hv = (h16_t)v[e]; o[e] = hv; v[e] -= (float)hv; }
      *reinterpret_cast<bf16x8*>(pl + (size_t)k * plane + p * cs16 + coff16 + c) = o;
    }
  }
}

extern "C" int hpri_to_planes(const float* x, int cs, int coff, void* planes, long long plane_stride, int cs16, int coff16,
                              long long P, int C, int cw16, int npl, hipStream_t stream) {
  HPRI_REQUIRE(x && planes, "to_planes: null pointer");
  HPRI_REQUIRE(P > 0 && C > 0 && cw16 >= C && cw16 % 8 == 0 && coff16 % 8 == 0 && coff16 + cw16 <= cs16 && cs16 % 8 == 0,
               "to_planes: bad plane geometry");
  HPRI_REQUIRE(cs % 4 == 0 && coff % 4 == 0 && npl >= 1 && npl <= 3 && plane_stride % 8 == 0, "to_planes: bad arguments");
  const long long total = P * (cw16 >> 3);
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(to_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, cs, coff,
                     reinterpret_cast<h16_t*>(planes), plane_stride, cs16, coff16, P, C, cw16, npl);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}
