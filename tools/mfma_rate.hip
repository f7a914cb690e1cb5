// Issue rate of the two MFMA instructions the convolution kernels are built on, measured with s_memtime inside the
// kernel: cycles per MFMA per SIMD as a function of how much of the chip is busy (power management, not the
// instruction's documented pass count, sets the practical ceiling).   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// NV independent vector-ALU instructions after every MFMA (VK = 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_xor_b32): does the
// vector ALU run in the MFMA's shadow, or does it take issue cycles away from the matrix pipe?
template <int KIND, int NV, int VK>
__global__ __launch_bounds__(512) void mix_kernel(unsigned long long* out, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = seed * (float)(threadIdx.x + i + r);
  float a = seed + threadIdx.x, b = seed - threadIdx.x;
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)(b - i); }
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 x[8];
  int xi[8];
  for (int i = 0; i < 8; ++i) { x[i] = f2{seed + i, seed - i}; xi[i] = threadIdx.x + i; }
  const f2 m = {seed, seed}, c = {seed * 0.5f, seed};
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int r = (i * NV + v) & 7;
          if (VK == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[r].x) : "v"(m.x), "v"(c.x));
          else if (VK == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[r]) : "v"(m), "v"(c));
          else asm volatile("v_xor_b32 %0, %1, %0" : "+v"(xi[r]) : "v"(xi[(r + 1) & 7]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  asm volatile("s_nop 0" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y + (float)xi[i];
  if (s == 12345.678f) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NV, int VK>
void run_mix(const char* name, const char* vname, int blocks, int threads, int iters, double ideal) {
  unsigned long long* d;
  hipMalloc(&d, (1 + (size_t)blocks * 8) * 8);
  hipMemset(d, 0, (1 + (size_t)blocks * 8) * 8);
  for (int w = 0; w < 2; ++w) mix_kernel<KIND, NV, VK><<<blocks, threads>>>(d, iters, 0.f);
  hipDeviceSynchronize();
  const int waves = threads / 64;
  std::vector<unsigned long long> h(1 + (size_t)blocks * 8);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> v;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) v.push_back((double)h[1 + b * 8 + w]);
  std::sort(v.begin(), v.end());
  const double per_simd = (double)iters * 16 * (waves / 4.0);
  printf("{\"mfma\": \"%s\", \"valu\": \"%s\", \"valu_per_mfma\": %d, \"workgroups\": %d, \"waves_per_simd\": %.1f, "
         "\"ticks_per_mfma\": %.2f, \"documented\": %.0f}\n", name, vname, NV, blocks, waves / 4.0, v[v.size() / 2] / per_simd, ideal);
  hipFree(d);
}

// Instruction mix of a fused fp32 Winograd main loop, per group of FOUR v_mfma_f32_32x32x2_f32: NV vector-ALU instructions
// (the input transform) and NL ds_read_b128 (halo + transformed-weight fragments), two or one waves per SIMD.
//   F(2x2,3x3) as built (conv_wino4.hip):          4 VALU + 2 reads per 4 MFMAs (32 + 16 per 32-MFMA stage)
//   mixed F(2,3) x F(4,3), 8 waves x 3 frequencies: 11 VALU + 3 reads per 4 MFMAs (64 + 18 per 24-MFMA stage), 0.75 x the MFMAs
// The quotient of the two ticks-per-MFMA figures times 0.75 is the most the mixed form's main loop can gain.
template <int NV, int NL>
__global__ __launch_bounds__(512) void wino_mix_kernel(unsigned long long* out, int iters, float seed) {
  __shared__ float4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_float4(seed, seed + i, seed - i, 1.f);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = seed * (float)(threadIdx.x + i + r);
  float x[12];
  for (int i = 0; i < 12; ++i) x[i] = seed + i;
  const float m = seed, c = seed * 0.5f;
  float4 f[3] = {lds[threadIdx.x], lds[threadIdx.x + 512], lds[threadIdx.x + 1024]};
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int l = 0; l < NL; ++l) f[l] = lds[(threadIdx.x + 64 * (it + u + l)) & 4095];     // conflict-free b128, address varies
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[i % NL].x + x[i], f[(i + 1) % NL].y, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = (i * NV) / 4; v < ((i + 1) * NV) / 4; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[v % 12]) : "v"(m), "v"(c));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 12; ++i) s += x[i];
  if (s == 12345.678f) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NV, int NL>
void run_wino_mix(const char* name, int blocks, int threads, int iters) {
  unsigned long long* d;
  hipMalloc(&d, (1 + (size_t)blocks * 8) * 8);
  hipMemset(d, 0, (1 + (size_t)blocks * 8) * 8);
  for (int w = 0; w < 2; ++w) wino_mix_kernel<NV, NL><<<blocks, threads>>>(d, iters, 0.25f);
  hipDeviceSynchronize();
  const int waves = threads / 64;
  std::vector<unsigned long long> h(1 + (size_t)blocks * 8);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> v;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) v.push_back((double)h[1 + b * 8 + w]);
  std::sort(v.begin(), v.end());
  const double per_simd = (double)iters * 16 * (waves / 4.0);
  printf("{\"winograd_mix\": \"%s\", \"valu_per_4_mfma\": %d, \"ds_read_b128_per_4_mfma\": %d, \"workgroups\": %d, \"waves_per_simd\": %.1f, "
         "\"ticks_per_mfma\": %.2f, \"documented\": 64}\n", name, NV, NL, blocks, waves / 4.0, v[v.size() / 2] / per_simd);
  hipFree(d);
}

__device__ __forceinline__ float rnd(unsigned x) {      // hash -> uniform in (-1, 1) * 2^-6
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return ((float)(x & 0xffffff) / 8388608.f - 1.f) * 0.015625f;
}
template <int KIND>
__global__ __launch_bounds__(512) void rate_kernel(unsigned long long* out, int iters, float seed) {
  f32x16 acc[4];
  const unsigned g = (blockIdx.x * 512 + threadIdx.x) * 64u + (unsigned)seed;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = seed == 0.f ? 0.f : rnd(g + i * 16 + r);
  float a = seed == 0.f ? 0.f : rnd(g + 101), b = seed == 0.f ? 0.f : rnd(g + 102);
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(seed == 0.f ? 0.f : rnd(g + 200 + i)); b8[i] = (__bf16)(seed == 0.f ? 0.f : rnd(g + 300 + i)); }
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[i], 0, 0, 0);
      }
  }
  asm volatile("s_nop 0" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = 1;                       // keep the accumulators alive
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int blocks, int threads, int iters, double ideal, float seed = 0.f) {
  unsigned long long* d;
  hipMalloc(&d, (1 + (size_t)blocks * 8) * 8);
  hipMemset(d, 0, (1 + (size_t)blocks * 8) * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 6; ++w) rate_kernel<KIND><<<blocks, threads>>>(d, iters, seed);   // warm, and load the chip
  hipEventRecord(e0);
  rate_kernel<KIND><<<blocks, threads>>>(d, iters, seed);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int waves = threads / 64;
  std::vector<unsigned long long> h(1 + (size_t)blocks * 8);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> v;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) v.push_back((double)h[1 + b * 8 + w]);
  std::sort(v.begin(), v.end());
  const double med = v[v.size() / 2];
  const double per_simd = (double)iters * 16 * (waves / 4.0);       // MFMAs one SIMD issues (waves/4 waves on it)
  const double flops = (KIND == 0 ? 4096.0 : 32768.0) * iters * 16.0 * waves * blocks;
  printf("{\"mfma\": \"%s\", \"data\": \"%s\", \"workgroups\": %d, \"waves_per_simd\": %.1f, \"ticks_per_mfma\": %.2f, \"documented\": %.0f, "
         "\"kernel_ms\": %.3f, \"ghz_if_ticks_are_clocks\": %.3f, \"tflops\": %.1f}\n", name, seed == 0.f ? "zeros" : "random", blocks,
         waves / 4.0, med / per_simd, ideal, ms, med / (ms * 1e6), flops / (ms * 1e9));
  hipFree(d);
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'w') {           // the Winograd instruction-mix model only
    for (int threads : {256, 512}) {
      run_wino_mix<4, 2>("F(2x2,3x3) as built", 256, threads, 4000);
      run_wino_mix<11, 3>("mixed F(2,3)xF(4,3)", 256, threads, 4000);
      run_wino_mix<0, 2>("no transform, 2 reads", 256, threads, 4000);
      run_wino_mix<0, 1>("no transform, 1 read", 256, threads, 4000);
    }
    return 0;
  }
  const int iters = 20000;
  for (int blocks : {8, 64, 256}) {
    for (int threads : {256, 512}) {
      run<0>("f32_32x32x2", blocks, threads, iters, 64);
      run<1>("bf16_32x32x16", blocks, threads, iters * 2, 32);
    }
  }
  for (int rep = 0; rep < 2; ++rep) {              // sustained: ~100 ms of back-to-back launches each
    run<0>("f32_32x32x2", 256, 512, 60000, 64, 7.f);
    run<1>("bf16_32x32x16", 256, 512, 120000, 32, 7.f);
    run<1>("bf16_32x32x16", 256, 256, 120000, 32, 7.f);
  }
  const int it2 = 4000;
#define MIX(K_, N_, V_, kn_, vn_, ideal_)                                  \
  run_mix<K_, N_, V_>(kn_, vn_, 256, 256, it2, ideal_);                     \
  run_mix<K_, N_, V_>(kn_, vn_, 256, 512, it2, ideal_);
  MIX(0, 2, 0, "f32_32x32x2", "v_fma_f32", 64)
  MIX(0, 4, 0, "f32_32x32x2", "v_fma_f32", 64)
  MIX(0, 8, 0, "f32_32x32x2", "v_fma_f32", 64)
  MIX(0, 2, 1, "f32_32x32x2", "v_pk_fma_f32", 64)
  MIX(0, 4, 1, "f32_32x32x2", "v_pk_fma_f32", 64)
  MIX(0, 4, 2, "f32_32x32x2", "v_xor_b32", 64)
  MIX(0, 8, 2, "f32_32x32x2", "v_xor_b32", 64)
  MIX(1, 2, 0, "bf16_32x32x16", "v_fma_f32", 32)
  MIX(1, 4, 0, "bf16_32x32x16", "v_fma_f32", 32)
  MIX(1, 2, 1, "bf16_32x32x16", "v_pk_fma_f32", 32)
  MIX(1, 4, 2, "bf16_32x32x16", "v_xor_b32", 32)
  return 0;
}
