// Ingest fast path: ENVI cubes come off the disk band-interleaved-by-pixel, i.e. (H, W, B) --
// already the channels-last order the conv kernels consume.  The reference moves the band axis to the front on the
// host (dataset.py:267 `np.moveaxis(..., -1, 0)`, a strided 560 MB copy), slices [hsi_lo:hsi_hi] (:268) and hands
// (B,H,W); the network then needs channels-last again.  Here the (H,W,B) array goes to the device as it is and ONE
// pass slices the bands, converts (fp16 sources) and zero-pads the channel stride -- or, for fp32 sources, the H2D
// copy itself lands in the padded layout (2-D memcpy: row = one pixel's bands) and no kernel runs at all.
#include "common.h"
#include <hip/hip_fp16.h>

// dst[p][c] = c < C ? src[p*B + lo + c] : 0     (c < dst_cw; dst_cw % 4 == 0; one thread per channel quad)
template <typename T>
__global__ void hwb_ingest_kernel(const T* __restrict__ src, float* __restrict__ dst, long long P, int B, int lo, int C,
                                  int dst_cs, int dst_cw) {
  const int q = dst_cw >> 2;
  const long long total = P * q;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / q;
    const int c = (int)(i - p * q) * 4;
    const T* s = src + p * B + lo + c;
    f32x4 v;
    v[0] = c + 0 < C ? (float)s[0] : 0.f;
    v[1] = c + 1 < C ? (float)s[1] : 0.f;
    v[2] = c + 2 < C ? (float)s[2] : 0.f;
    v[3] = c + 3 < C ? (float)s[3] : 0.f;
    *reinterpret_cast<f32x4*>(dst + p * dst_cs + c) = v;
  }
}

// src_dtype: 0 = float32, 1 = float16
extern "C" int hpri_hwb_ingest(const void* src, int src_dtype, float* dst, long long P, int B, int lo, int C, int dst_cs,
                               int dst_cw, hipStream_t stream) {
  HPRI_REQUIRE(src && dst && P > 0 && B > 0 && lo >= 0 && C > 0 && lo + C <= B, "hwb_ingest: bad band range");
  HPRI_REQUIRE(dst_cw % 4 == 0 && dst_cw >= C && dst_cw <= dst_cs && dst_cs % 4 == 0, "hwb_ingest: bad destination layout");
  HPRI_REQUIRE(src_dtype == 0 || src_dtype == 1, "hwb_ingest: src_dtype must be 0 (f32) or 1 (f16)");
  long long blocks = (P * (dst_cw / 4) + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (src_dtype == 0)
    hipLaunchKernelGGL(hwb_ingest_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)src, dst, P, B,
                       lo, C, dst_cs, dst_cw);
  else
    hipLaunchKernelGGL(hwb_ingest_kernel<__half>, dim3((unsigned)blocks), dim3(256), 0, stream, (const __half*)src, dst, P,
                       B, lo, C, dst_cs, dst_cw);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// fp32 (H,W,B) in (pinned) host memory -> bands [lo, lo+C) of every pixel at stride dst_cs on the device, as one
// asynchronous 2-D copy.  The pad channels [C, dst_cs) are not touched: the caller zeroes them once at allocation.
extern "C" int hpri_hwb_h2d(const float* host_src, float* dst, long long P, int B, int lo, int C, int dst_cs,
                            hipStream_t stream) {
  HPRI_REQUIRE(host_src && dst && P > 0 && B > 0 && lo >= 0 && C > 0 && lo + C <= B && dst_cs >= C,
               "hwb_h2d: bad arguments");
  const hipError_t e = hipMemcpy2DAsync(dst, (size_t)dst_cs * sizeof(float), host_src + lo, (size_t)B * sizeof(float),
                                        (size_t)C * sizeof(float), (size_t)P, hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) return hpri_set_error(HPRI_ERR_LAUNCH, hipGetErrorString(e));
  return HPRI_OK;
}
