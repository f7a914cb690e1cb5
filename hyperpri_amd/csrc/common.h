// Shared definitions for the hyperpri_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HPRI_OK 0
#define HPRI_ERR_ARG (-1)      // bad shape / alignment / null pointer
#define HPRI_ERR_UNSUPPORTED (-2)
#define HPRI_ERR_WORKSPACE (-3)
#define HPRI_ERR_LAUNCH (-4)   // hipGetLastError() after a launch was not hipSuccess

// The 16-bit storage / MFMA operand type of the "plane" paths.  The product library is built twice from the same sources
// (hyperpri_amd/build.py): libhyperpri_hip.so with h16_t = bf16 (precision modes "bf16", "bf16x3", "bf16x6"; fp32 kernels), and
// libhyperpri_hip_f16.so (-DHPRI_H16_F16) with h16_t = IEEE half for precision mode "f16": the same kernels, activations / pre-BN
// tensors / activation gradients stored as fp16, v_mfma_*_f16 -- 11 mantissa bits where bf16 has 8 (max |dlogit|
// 3.5e-3 against 2.3e-2), 5 exponent bits: the engine scales the gradient that enters the head by a power of two (loss scale) and
// takes it out of the parameter gradients again.  Entry points keep their names in both libraries ("bf16" = "the 16-bit type").
#ifdef HPRI_H16_F16
typedef _Float16 h16_t;
#define HPRI_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define HPRI_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define HPRI_DS_READ_TR16_B64 __builtin_amdgcn_ds_read_tr16_b64_v4f16
#define HPRI_H16_IS_F16 1
#else
typedef __bf16 h16_t;
#define HPRI_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define HPRI_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define HPRI_DS_READ_TR16_B64 __builtin_amdgcn_ds_read_tr16_b64_v4bf16
#define HPRI_H16_IS_F16 0
#endif

// (the transposed-read builtin of the half type wants __fp16 vectors)
#ifdef HPRI_H16_F16
typedef __fp16 hpri_tr4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
#else
typedef __bf16 hpri_tr4_t __attribute__((ext_vector_type(4)));
#endif
typedef hpri_tr4_t __attribute__((address_space(3))) * hpri_lds_tr4_ptr;

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
int hpri_option(int idx);   // 0 conv_nbx_min, 1 wgrad_xcd_min_tiles, 2 wgrad_xcd_min_strips, 3 bf16v3_tile_width, 4 bn_wide_cq, 5 wgrad_cu_reserve (api.cpp; thread-safe)
int hpri_cu_count();        // compute units of the current device (api.cpp; cached)
float hpri_loss_scale();    // the calling thread's loss scale (hpri_set_loss_scale; 1 unless set)

static inline int hpri_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t hpri_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// A-operand / epilogue addressing modes of the implicit-GEMM kernels.
enum { HPRI_A_DIRECT = 0, HPRI_A_S2D = 1 };   // S2D: gather 2x2 stride-2 patches (convT dgrad / wgrad)
enum { HPRI_E_DIRECT = 0, HPRI_E_D2S = 1 };   // D2S: scatter 2x2 stride-2 patches (convT forward)

// Optional second output of the element-wise producers: the same values as bf16 NHWC planes (hi | hi,lo | hi,mid,lo:
// plane k = bf16 of what the previous planes left), which the bf16-mode convolutions stage by LDS-DMA
// (conv_bf16v2.hip).  Channels [C, cw) of the planes are zero-filled by the producer.
typedef h16_t bf16x4_t __attribute__((ext_vector_type(4)));
struct PlaneOut { h16_t* p; long long plane; int cs, coff, cw, npl; };

#ifdef __HIPCC__
__device__ __forceinline__ void plane_store4(const PlaneOut& pl, size_t pix, int c, float o0, float o1, float o2, float o3) {
  float v[4] = {o0, o1, o2, o3};
  for (int k = 0; k < pl.npl; ++k) {
    bf16x4_t h;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const h16_t t = (h16_t)v[j]; h[j] = t; v[j] -= (float)t; }
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
#define HPRI_BUFFER_LOAD_F32(rs_, voff_, soff_) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_, voff_, soff_, 0))
#else
typedef int hpri_rsrc_t;
#define HPRI_MAKE_RSRC(ptr_, bytes_) 0
#define HPRI_LDS_DMA16(rs_, lds_, voff_, soff_) ((void)(rs_))
#define HPRI_BUFFER_LOAD_F32(rs_, voff_, soff_) 0.f
#endif

// ---- item queues of the persistent kernels (conv_bf16v3, gemm_bf16v3, gemm_f32v2) ----------------------------------------------
// These kernels start 2 x CUs workgroups (two per CU, all of a SIMD's vector registers and ~75 KB of LDS each).  With FIXED item
// lists a workgroup that cannot become resident at once -- a collective's kernel holds part of its CU (RCCL beside the backward of
// a DDP step) -- starts when another one retires and then walks its whole list alone: ONE such workgroup costs a bf16 step 6-15 %
// (tools/cu_share_probe.py, profiles/r05_cu_share_before.json).  With a queue (hpri_set_item_queue: caller-owned counters, one
// buffer per stream) the workgroups DRAW their items: counter x of K slice z hands out the items of band x (the items the
// workgroups with id mod 8 == x used to walk, in the same order); a workgroup that becomes resident late finds the band
// exhausted and leaves.  Which workgroup computes an item changes, nothing else: results are bit-identical.
//   * Device-scope atomics are executed on the memory side of the fabric (~2 us, and they queue per address): a first version
//     with one synchronous draw per workgroup at kernel start, stealing from the other bands and an exit count cost every launch
//     ~30 us (profiles/r05_queue_ab_v1.jsonl).  So: the FIRST occupant of every CU (workgroup ids below the CU count) takes its
//     first item statically, as in the fixed lists, and draws from its second item on; the second occupants draw their first item
//     while they sit out their start-up stagger; every later ticket is drawn one item ahead, during the previous item's epilogue,
//     and parked in LDS -- no workgroup ever waits for an atomic inside the item loop, and none is issued at the end.
//   * The buffer holds two halves of HPRI_Q_HALF counters; launches on a stream alternate (the launcher keeps the parity), and
//     every launch zeroes the half the NEXT one will use: no exit protocol, no memset node between launches.
//   * Every counter has a 4 KB stretch of the buffer to itself: with the eight band counters side by side in one 64-byte line all
//     512 workgroups of a launch queued at ONE memory channel (+20-40 us on a 100 us launch: profiles/r05_queue_ab_v2.jsonl).
//     Launches with fewer than two items per workgroup keep their fixed lists: there a ticket's latency (same-address atomics take
//     turns at the memory side, a few hundred ns each) sits on the critical path of the second occupants and nothing absorbs it
//     (+23 % on the 72-items-per-band layers at 76x121), and a late workgroup costs one or two items at most anyway.
#define HPRI_Q_SLICES 8
#define HPRI_Q_STRIDE 1024                       // 32-bit words between two counters
#define HPRI_Q_COUNTERS (8 * HPRI_Q_SLICES)      // per half: [K slice][band]
#define HPRI_Q_HALF (HPRI_Q_COUNTERS * HPRI_Q_STRIDE)
#define HPRI_Q_WORDS (2 * HPRI_Q_HALF)
struct HpriQueueHalves { unsigned *use, *clear; };
HpriQueueHalves hpri_item_queue_take(hipStream_t stream);     // this launch's half and the one it must zero ({nullptr, nullptr}: no queue); api.cpp
#ifdef __HIPCC__
// q: the counters of one K slice (half + slice * 8 * HPRI_Q_STRIDE)
__device__ __forceinline__ unsigned hpri_q_draw(unsigned* q, int band, unsigned n) {
  return __hip_atomic_fetch_add(q + band * HPRI_Q_STRIDE, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (the first 64 threads of one workgroup) zero the half the next launch on this stream will draw from
__device__ __forceinline__ void hpri_q_clear(unsigned* other, int tid) {
  if (other != nullptr && tid < HPRI_Q_COUNTERS) __hip_atomic_store(other + tid * HPRI_Q_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

// host: fill a PlaneOut from C-ABI arguments (planes == nullptr: no plane output)
static inline int hpri_plane_out(PlaneOut* po, void* planes, long long plane_stride, int pl_cs, int pl_coff, int pl_cw, int npl, int C) {
  po->p = reinterpret_cast<h16_t*>(planes); po->plane = plane_stride; po->cs = pl_cs; po->coff = pl_coff; po->cw = pl_cw; po->npl = npl;
  if (planes == nullptr) { po->cw = 0; po->npl = 0; return HPRI_OK; }
  HPRI_REQUIRE(npl >= 1 && npl <= 3 && pl_cw >= C && pl_cw % 4 == 0 && pl_coff % 4 == 0 && pl_cs % 4 == 0 && pl_coff + pl_cw <= pl_cs &&
                   plane_stride % 4 == 0 && ((uintptr_t)planes & 7) == 0, "bad plane geometry");
  return HPRI_OK;
}
