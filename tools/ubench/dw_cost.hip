// Cycles per depthwise k-step (36 fma + 6 dpp + 6 cndmask + 4 max) on gfx950, 1 / 2 waves per SIMD,
// with and without four fp32 MFMAs beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float lane_prev(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float lane_next(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130, 0xf, 0xf, true)); }
template <int MODE>
__global__ void k(const float* wsrc, float* out, long long* cyc, int iters) {
  float wd[10];
  for (int i = 0; i < 10; i++) wd[i] = wsrc[i];
  float4 m3[3];
  for (int i = 0; i < 3; i++) m3[i] = make_float4(threadIdx.x + i, threadIdx.x * 2.f + i, 1.f + i, 2.f * i);
  const bool leftok = threadIdx.x % 7 != 0, rightok = threadIdx.x % 5 != 0;
  floatx16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float s = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    float t0_ = wd[9], t1 = t0_, t2 = t0_, t3 = t0_;
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      const float w0 = wd[3 * kk], w1 = wd[3 * kk + 1], w2 = wd[3 * kk + 2];
      const float4 m = m3[kk];
      const float fp = lane_prev(m.w), fn = lane_next(m.x);
      const float l = leftok ? fp : 0.f, rr = rightok ? fn : 0.f;
      t0_ = fmaf(w0, l, t0_), t1 = fmaf(w0, m.x, t1), t2 = fmaf(w0, m.y, t2), t3 = fmaf(w0, m.z, t3);
      t0_ = fmaf(w1, m.x, t0_), t1 = fmaf(w1, m.y, t1), t2 = fmaf(w1, m.z, t2), t3 = fmaf(w1, m.w, t3);
      t0_ = fmaf(w2, m.y, t0_), t1 = fmaf(w2, m.z, t1), t2 = fmaf(w2, m.w, t2), t3 = fmaf(w2, rr, t3);
    }
    t0_ = fmaxf(t0_, 0.f), t1 = fmaxf(t1, 0.f), t2 = fmaxf(t2, 0.f), t3 = fmaxf(t3, 0.f);
    if (MODE == 1) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wd[0], t0_, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wd[0], t1, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wd[0], t2, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wd[0], t3, c3, 0, 0, 0);
    } else {
      s += t0_ + t1 + t2 + t3;
    }
    // next window: cheap perturbation so nothing is loop-invariant
#pragma unroll
    for (int i = 0; i < 3; i++) m3[i].x += 1.f, m3[i].w += 0.5f;
  }
  const long long t1c = clock64();
  for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1c - t0;
}
template <int MODE>
void run(const char* name, int threads) {
  float *out, *w; long long* cyc;
  (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&w, 64);
  float hw[10] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f, 0.9f, 0.01f};
  (void)hipMemcpy(w, hw, 40, hipMemcpyHostToDevice);
  const int iters = 2000;
  k<MODE><<<1, threads>>>(w, out, cyc, iters);
  k<MODE><<<1, threads>>>(w, out, cyc, iters);
  long long h = 0; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-18s threads %4d: %.1f cycles per k-step per wave\n", name, threads, (double)h / iters);
}
int main() {
  for (int threads : {256, 512}) { run<0>("dw only", threads); run<1>("dw + 4 mfma", threads); }
  return 0;
}
