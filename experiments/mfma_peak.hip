// Calibration: back-to-back v_mfma_f32_32x32x2_f32 from registers, NACC independent accumulators per wave,
// WAVES waves per workgroup, BLOCKS workgroups per CU.  Prints TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int threads) {
  float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, 100, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * (threads / 64) * iters * 4.0 * NACC * 2.0 * 32 * 32 * 2;
  printf("NACC=%d blocks/CU=%d waves/block=%d : %.1f TFLOP/s (%.2f ms)\n", NACC, blocks_per_cu, threads / 64, flops / ms / 1e9, ms);
  hipFree(out);
}
int main() {
  run<4>(1, 256); run<4>(2, 256); run<1>(1, 256); run<2>(1, 256); run<4>(1, 512); run<9>(2, 256); run<4>(4, 256);
  return 0;
}
