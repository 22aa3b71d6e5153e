// Issue rate of the fp32 MFMAs on gfx950: cycles per instruction (s_memtime), 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  const float a = threadIdx.x * 0.001f, b = a + 1.f;
  floatx16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  floatx4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
  float v0 = a, v1 = b, v2 = a, v3 = b;
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0 || MODE == 2) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    } else if (MODE == 1) {
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d3, 0, 0, 0);
    }
    if (MODE == 2) {  // 4 MFMA + 48 independent fma
#pragma unroll
      for (int u = 0; u < 12; u++) {
        v0 = fmaf(v0, 1.0001f, 0.5f), v1 = fmaf(v1, 1.0001f, 0.5f), v2 = fmaf(v2, 1.0001f, 0.5f), v3 = fmaf(v3, 1.0001f, 0.5f);
      }
    }
  }
  const long long t1 = clock64();
  float s = v0 + v1 + v2 + v3;
  for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
  for (int r = 0; r < 4; r++) s += d0[r] + d1[r] + d2[r] + d3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads, int blocks) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 64 << 20); (void)hipMalloc(&cyc, 8);
  const int iters = 2000;
  k<MODE><<<blocks, threads>>>(out, cyc, iters);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, threads>>>(out, cyc, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h = 0; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-22s threads %4d blocks %5d: %.1f cycles per MFMA per wave, kernel %.3f ms\n", name, threads, blocks, (double)h / (iters * 4), ms);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  for (int threads : {256, 512, 1024}) {
    run<0>("32x32x2 f32", threads, 1); run<1>("16x16x4 f32", threads, 1); run<2>("32x32x2 + 12 fma each", threads, 1);
  }
  run<0>("32x32x2 f32 full chip", 512, 256);
  run<0>("32x32x2 f32 full chip", 512, 1024);
  return 0;
}
