// tools/cu_share_probe.py: a stand-in for a communication kernel that HOLDS compute units while the step under test runs --
// W workgroups of 512 threads with 64 KB of LDS each, asleep until a deadline on the constant 100 MHz clock (s_memrealtime).
// Sleeping waves issue nothing: what is measured is the occupancy effect alone (the workgroups of a persistent kernel that no
// longer fit on the CUs a hog sits on).  Every wave leaves at the deadline: the grid always drains.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/bin/libcuhog.so tools/cu_hog.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(512) void cu_hog_kernel(unsigned long long ticks, unsigned* out) {
  extern __shared__ unsigned lds[];                     // 64 KB by default (cu_hog_launch_lds: any size >= 2 KB)
  lds[threadIdx.x] = threadIdx.x;                       // (the allocation must be real)
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
  if (out != nullptr && threadIdx.x == 0) out[blockIdx.x] = lds[0] + 1u;
}

extern "C" int cu_hog_launch_lds(int workgroups, double milliseconds, int lds_bytes, void* out, hipStream_t stream) {
  if (workgroups <= 0 || milliseconds <= 0.0 || milliseconds > 5000.0 || lds_bytes < 2048 || lds_bytes > 65536) return -1;      // bounded by construction
  const unsigned long long ticks = (unsigned long long)(milliseconds * 1e5);           // 100 MHz
  hipLaunchKernelGGL(cu_hog_kernel, dim3(workgroups), dim3(512), (size_t)lds_bytes, stream, ticks, reinterpret_cast<unsigned*>(out));
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int cu_hog_launch(int workgroups, double milliseconds, void* out, hipStream_t stream) {
  return cu_hog_launch_lds(workgroups, milliseconds, 65536, out, stream);
}
