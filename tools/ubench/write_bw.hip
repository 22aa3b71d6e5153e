// What store bandwidth does a plain kernel reach on this GPU?  (The stem writes 157 MB per batch in ~59 us = 2.7 TB/s and
// reads ~15 MB: is that the memory system or the kernel?)  One float4 store per thread per iteration, grid-stride, sizes
// from 16 MB to 512 MB; also a copy (read + write).  Build: hipcc -O3 --offload-arch=gfx950 write_bw.hip -o write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fill(float4* dst, size_t n4, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_float4(v, v, v, v);
}
__global__ void k_copy(float4* dst, const float4* src, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
  const size_t cap = 512ull << 20;
  float4 *a, *b;
  if (hipMalloc(&a, cap) != hipSuccess || hipMalloc(&b, cap) != hipSuccess) return 2;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (size_t mb : {16, 64, 157, 512}) {
    const size_t n4 = (mb << 20) / 16;
    for (int blocks : {2048, 8192}) {
      for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, a, n4, 1.0f);
      hipEventRecord(e0);
      for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, a, n4, (float)r);
      hipEventRecord(e1), hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double fill = (double)(mb << 20) * 20 / (ms * 1e-3) / 1e12;
      for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, b, (const float4*)a, n4);
      hipEventRecord(e0);
      for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, b, (const float4*)a, n4);
      hipEventRecord(e1), hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      const double copy = (double)(mb << 20) * 2 * 20 / (ms * 1e-3) / 1e12;
      std::printf("%4zu MB, %5d blocks: fill %.2f TB/s (%.1f us), copy %.2f TB/s read+write\n", mb, blocks, fill, (double)(mb << 20) / fill / 1e6, copy);
    }
  }
  return 0;
}
