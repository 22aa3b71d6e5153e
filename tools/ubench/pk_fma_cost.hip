// Does packed fp32 math (v_pk_fma_f32: two fma per lane per instruction) buy anything for the depthwise arithmetic that
// shares a SIMD with fp32 MFMAs on gfx950?  Cycles per loop iteration per wave (s_memtime), one and two waves per SIMD:
//   48 independent v_fma_f32 | 24 v_pk_fma_f32 (the same 48 fma) | 4 MFMA 32x32x2 | 4 MFMA + 48 fma | 4 MFMA + 24 pk_fma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  const float a = threadIdx.x * 0.001f, b = a + 1.f;
  floatx16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float v[8];
  float2v p[4], w = {1.0001f, 0.9999f}, x = {0.5f, 0.25f};
  for (int i = 0; i < 8; i++) v[i] = a + i;
  for (int i = 0; i < 4; i++) p[i] = float2v{a + i, b + i};
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    if (MODE >= 2) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    if (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int u = 0; u < 6; u++)
#pragma unroll
        for (int q = 0; q < 8; q++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(w.x), "v"(x.x));
    }
    if (MODE == 1 || MODE == 4) {
#pragma unroll
      for (int u = 0; u < 6; u++)
#pragma unroll
        for (int q = 0; q < 4; q++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[q]) : "v"(w), "v"(x));
    }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; i++) s += v[i];
  for (int i = 0; i < 4; i++) s += p[i].x + p[i].y;
  for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads, int blocks) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 64 << 20); (void)hipMalloc(&cyc, 8);
  const int iters = 2000;
  k<MODE><<<blocks, threads>>>(out, cyc, iters);
  k<MODE><<<blocks, threads>>>(out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long h = 0; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s threads %4d blocks %4d: %7.1f cycles per iteration per wave\n", name, threads, blocks, (double)h / iters);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  for (int threads : {256, 512}) {
    run<0>("48 v_fma_f32", threads, 1);
    run<1>("24 v_pk_fma_f32", threads, 1);
    run<2>("4 MFMA 32x32x2", threads, 1);
    run<3>("4 MFMA + 48 v_fma_f32", threads, 1);
    run<4>("4 MFMA + 24 v_pk_fma_f32", threads, 1);
  }
  run<3>("4 MFMA + 48 fma, full chip", 512, 256);
  run<4>("4 MFMA + 24 pk_fma, full chip", 512, 256);
  return 0;
}
