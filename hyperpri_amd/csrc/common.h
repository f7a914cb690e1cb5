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
int hpri_option(int idx);   // 0 conv_nbx_min, 1 wgrad_xcd_min_tiles, 2 wgrad_xcd_min_strips (api.cpp)

static inline int hpri_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t hpri_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// A-operand / epilogue addressing modes of the implicit-GEMM kernels.
enum { HPRI_A_DIRECT = 0, HPRI_A_S2D = 1 };   // S2D: gather 2x2 stride-2 patches (convT dgrad / wgrad)
enum { HPRI_E_DIRECT = 0, HPRI_E_D2S = 1 };   // D2S: scatter 2x2 stride-2 patches (convT forward)
