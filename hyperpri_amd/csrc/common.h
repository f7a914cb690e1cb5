// Shared definitions for the hyperpri_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HPRI_OK 0
#define HPRI_ERR_ARG (-1)      // bad shape / alignment / null pointer
#define HPRI_ERR_UNSUPPORTED (-2)
#define HPRI_ERR_WORKSPACE (-3)
#define HPRI_ERR_LAUNCH (-4)   // hipGetLastError() after a launch was not hipSuccess

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HPRI_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return hpri_set_error(HPRI_ERR_LAUNCH, hipGetErrorString(e__)); \
  } while (0)

#define HPRI_REQUIRE(cond, msg)                               \
  do {                                                        \
    if (!(cond)) return hpri_set_error(HPRI_ERR_ARG, msg);    \
  } while (0)

int hpri_set_error(int code, const char* msg);
int hpri_option(int idx);   // 0 conv_nbx_min, 1 wgrad_xcd_min_tiles, 2 wgrad_xcd_min_strips, 3 bf16v3_tile_width, 4 bn_wide_cq (api.cpp; thread-safe)
int hpri_cu_count();        // compute units of the current device (api.cpp; cached)

static inline int hpri_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t hpri_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// A-operand / epilogue addressing modes of the implicit-GEMM kernels.
enum { HPRI_A_DIRECT = 0, HPRI_A_S2D = 1 };   // S2D: gather 2x2 stride-2 patches (convT dgrad / wgrad)
enum { HPRI_E_DIRECT = 0, HPRI_E_D2S = 1 };   // D2S: scatter 2x2 stride-2 patches (convT forward)

// Optional second output of the element-wise producers: the same values as bf16 NHWC planes (hi | hi,lo | hi,mid,lo:
// plane k = bf16 of what the previous planes left), which the bf16-mode convolutions stage by LDS-DMA
// (conv_bf16v2.hip).  Channels [C, cw) of the planes are zero-filled by the producer.
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
struct PlaneOut { __bf16* p; long long plane; int cs, coff, cw, npl; };

#ifdef __HIPCC__
__device__ __forceinline__ void plane_store4(const PlaneOut& pl, size_t pix, int c, float o0, float o1, float o2, float o3) {
  float v[4] = {o0, o1, o2, o3};
  for (int k = 0; k < pl.npl; ++k) {
    bf16x4_t h;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const __bf16 t = (__bf16)v[j]; h[j] = t; v[j] -= (float)t; }
    *reinterpret_cast<bf16x4_t*>(pl.p + (size_t)k * pl.plane + pix * pl.cs + pl.coff + c) = h;
  }
}
#endif

// LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... lds): 16 bytes per lane from base + soffset + voffset into
// the wave's 1 KB LDS piece.  A lane whose voffset lies beyond the descriptor's range gets ZEROS written into its LDS slot
// (hardware range check; tools/lds_dma_oob.hip) -- used for halo pixels outside the image.  The instruction's immediate
// offset would be added to the LDS address as well, so every offset goes through soffset / voffset.  (The host pass of a
// templated kernel drops the instantiation when it meets the device-only descriptor type: hence the host-side dummies.)
#define HPRI_DMA_OOB 0xFFFFFFF0u
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t hpri_rsrc_t;
#define HPRI_MAKE_RSRC(ptr_, bytes_) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr_), 0, (int)(bytes_), 0x00020000)
#define HPRI_LDS_DMA16(rs_, lds_, voff_, soff_) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (__attribute__((address_space(3))) void*)(lds_), 16, voff_, soff_, 0, 0)
#else
typedef int hpri_rsrc_t;
#define HPRI_MAKE_RSRC(ptr_, bytes_) 0
#define HPRI_LDS_DMA16(rs_, lds_, voff_, soff_) ((void)(rs_))
#endif

// host: fill a PlaneOut from C-ABI arguments (planes == nullptr: no plane output)
static inline int hpri_plane_out(PlaneOut* po, void* planes, long long plane_stride, int pl_cs, int pl_coff, int pl_cw, int npl, int C) {
  po->p = reinterpret_cast<__bf16*>(planes); po->plane = plane_stride; po->cs = pl_cs; po->coff = pl_coff; po->cw = pl_cw; po->npl = npl;
  if (planes == nullptr) { po->cw = 0; po->npl = 0; return HPRI_OK; }
  HPRI_REQUIRE(npl >= 1 && npl <= 3 && pl_cw >= C && pl_cw % 4 == 0 && pl_coff % 4 == 0 && pl_cs % 4 == 0 && pl_coff + pl_cw <= pl_cs &&
                   plane_stride % 4 == 0 && ((uintptr_t)planes & 7) == 0, "bad plane geometry");
  return HPRI_OK;
}
