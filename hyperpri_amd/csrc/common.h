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

// ---- item queues of the persistent kernels (conv_bf16v3, gemm_bf16v3, gemm_f32v2) ----------------------------------------------
// These kernels start 2 x CUs workgroups (two per CU, all of a SIMD's vector registers and ~75 KB of LDS each).  With FIXED item
// lists a workgroup that cannot become resident at once -- a collective's kernel holds part of its CU (RCCL beside the backward of
// a DDP step) -- starts when another one retires and then walks its whole list alone: a second wave, up to 2 x the launch.  With a
// queue (hpri_set_item_queue: HPRI_Q_WORDS zeroed 32-bit counters owned by the caller, one buffer per stream) every workgroup
// DRAWS its items: counter x of K slice z hands out the items of XCD x's band in the same order the fixed lists walked them, a
// workgroup whose own band is exhausted helps the other XCDs out, and a late workgroup finds the queues empty and leaves.  Which
// workgroup computes an item changes, nothing else: results are bit-identical.  The last workgroup to leave re-arms the counters.
#define HPRI_Q_SLICES 8
#define HPRI_Q_WORDS (8 * HPRI_Q_SLICES + 1)
void* hpri_item_queue(hipStream_t stream);     // the queue registered for this stream, or nullptr (api.cpp)
#ifdef __HIPCC__
__device__ __forceinline__ unsigned hpri_q_draw(unsigned* q, int xcd) {
  return __hip_atomic_fetch_add(q + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (one thread) draw from queue *qx; when that band is exhausted try the other XCDs' in turn.  nvalid(x) = items of band x.
// Returns the item's index inside the band and leaves the band in *qx, or -1: nothing left anywhere.
template <typename NV>
__device__ __forceinline__ int hpri_q_steal(unsigned* q, int* qx, unsigned first, NV nvalid) {
  unsigned v = first;
  int x = *qx;
  for (int j = 0;; ++j) {
    if (v < (unsigned)nvalid(x)) { *qx = x; return (int)v; }
    if (j == 7) return -1;
    x = (x + 1) & 7;
    v = hpri_q_draw(q, x);
  }
}
// (one thread, once per workgroup, after its last draw has returned) count the workgroup out
__device__ __forceinline__ void hpri_q_leave(unsigned* qbase, unsigned total_workgroups, int nslices) {
  unsigned* done = qbase + 8 * HPRI_Q_SLICES;
  if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == total_workgroups) {
    for (int i = 0; i < 8 * nslices; ++i) __hip_atomic_store(qbase + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
#endif

// host: fill a PlaneOut from C-ABI arguments (planes == nullptr: no plane output)
static inline int hpri_plane_out(PlaneOut* po, void* planes, long long plane_stride, int pl_cs, int pl_coff, int pl_cw, int npl, int C) {
  po->p = reinterpret_cast<__bf16*>(planes); po->plane = plane_stride; po->cs = pl_cs; po->coff = pl_coff; po->cw = pl_cw; po->npl = npl;
  if (planes == nullptr) { po->cw = 0; po->npl = 0; return HPRI_OK; }
  HPRI_REQUIRE(npl >= 1 && npl <= 3 && pl_cw >= C && pl_cw % 4 == 0 && pl_coff % 4 == 0 && pl_cs % 4 == 0 && pl_coff + pl_cw <= pl_cs &&
                   plane_stride % 4 == 0 && ((uintptr_t)planes & 7) == 0, "bad plane geometry");
  return HPRI_OK;
}
