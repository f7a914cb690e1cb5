// What does an out-of-range lane of a buffer_load ... lds (LDS-DMA) do to its LDS destination: write zero, or leave it?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* p, float* out, int valid_bytes) {
  __shared__ __attribute__((aligned(1024))) float lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -7.f;            // sentinel
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, valid_bytes, 0x00020000);
  // lanes 0-31: in range; lanes 32-47: offsets beyond num_records; lanes 48-63: voffset 0xFFFFFFF0
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x >= 48) voff = 0xFFFFFFF0u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  float *p, *o; float h[256], src[256];
  for (int i = 0; i < 256; ++i) src[i] = 1.f + i;
  hipMalloc(&p, 1024); hipMalloc(&o, 1024);
  hipMemcpy(p, src, 1024, hipMemcpyHostToDevice);
  k<<<1, 64>>>(p, o, 32 * 16);                                          // 512 valid bytes: lanes 0-31
  hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
  printf("lane 0: %g %g | lane 31: %g | lane 32 (beyond num_records): %g %g | lane 47: %g | lane 48 (voffset ~4G): %g | lane 63: %g\n",
         h[0], h[1], h[31 * 4], h[32 * 4], h[32 * 4 + 1], h[47 * 4], h[48 * 4], h[63 * 4]);
  printf("%s\n", (h[32 * 4] == 0.f && h[48 * 4] == 0.f) ? "OOB lanes WRITE ZERO to LDS" : (h[32 * 4] == -7.f ? "OOB lanes LEAVE LDS untouched" : "OOB lanes: other"));
  return 0;
}
