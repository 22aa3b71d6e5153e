// Does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950 (hip_ext.h says the flag is
// not supported on GFX9xx boards)?  Two spin kernels of 64 blocks each, back to back in one stream: 2T serial, T overlapped.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, int* out) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}
int main() {
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* d;
  (void)hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  const long long cyc = 200000;  // ~2 ms at 100 MHz, ~0.1 ms at shader clock: either way long against launch cost
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(e0, s);
      hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, cyc, d);
      if (mode == 0) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, cyc, d + 64);
      if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d + 64);
      (void)hipEventRecord(e1, s);
      (void)hipStreamSynchronize(s);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%s: %.3f ms (%s)\n", mode == 0 ? "two normal launches" : mode == 1 ? "normal + any-order" : "one launch", ms, hipGetErrorString(hipGetLastError()));
    }
  }
  return 0;
}
