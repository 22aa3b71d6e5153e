// The k-loop of the dw->pw kernels in isolation (gfx950): cycles per k-step per wave at two waves per SIMD
// with the pieces switched on one by one: MODE bit 0 = depthwise VALU work, bit 1 = four fp32 MFMAs,
// bit 2 = depthwise + pointwise weights from LDS (else registers), bit 3 = input windows from global
// memory, prefetched two k-steps ahead (else registers), bit 4 = MFMAs of a k-step right behind its own
// depthwise work instead of software-pipelined against the next step's.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float lane_prev(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float lane_next(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130, 0xf, 0xf, true)); }
constexpr int KS = 64;
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, const float* __restrict__ wsrc, float* out, long long* cyc, size_t chan_step) {
  extern __shared__ float s_mem[];
  float* s_dw = s_mem;             // [2*KS][12]
  float* s_w = s_mem + 2 * KS * 12;  // [KS][64]
  for (int i = threadIdx.x; i < 2 * KS * 12 + KS * 64; i += 256) s_mem[i] = wsrc[i % 977] * 0.01f;
  __syncthreads();
  const int lane = threadIdx.x & 63, half = lane >> 5;
  const bool leftok = lane % 7 != 0, rightok = lane % 5 != 0;
  const size_t base = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 * 4 + (size_t)lane * 4;
  const float* p0 = in + base;
  floatx16 acc[4] = {{0}, {0}, {0}, {0}};
  float4 win[2][3];
  auto load_window = [&](int ks, float4 (&m)[3]) {
    if (MODE & 8) {
      const float* b = p0 + (size_t)ks * chan_step;
#pragma unroll
      for (int r = 0; r < 3; r++) m[r] = *reinterpret_cast<const float4*>(b + (size_t)r * 1024 * 1024);
    } else {
#pragma unroll
      for (int r = 0; r < 3; r++) m[r].x += 1.f, m[r].w += 0.5f;
    }
  };
  float wreg[10];
#pragma unroll
  for (int i = 0; i < 10; i++) wreg[i] = s_dw[i];
  auto dw_compute = [&](const float4 (&m3)[3], int ks, float (&t)[4]) {
    float wd[10];
    if (MODE & 4) {
      const float4* wq = reinterpret_cast<const float4*>(s_dw + (2 * ks + half) * 12);
      const float4 q0 = wq[0], q1 = wq[1], q2 = wq[2];
      wd[0] = q0.x, wd[1] = q0.y, wd[2] = q0.z, wd[3] = q0.w, wd[4] = q1.x, wd[5] = q1.y, wd[6] = q1.z, wd[7] = q1.w, wd[8] = q2.x, wd[9] = q2.y;
    } else {
#pragma unroll
      for (int i = 0; i < 10; i++) wd[i] = wreg[i];
    }
    if (!(MODE & 1)) {
      t[0] = m3[0].x + wd[0], t[1] = m3[1].y, t[2] = m3[2].z, t[3] = m3[0].w;
      return;
    }
    float t0 = wd[9], t1 = t0, t2 = t0, t3 = t0;
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      const float w0 = wd[3 * kk], w1 = wd[3 * kk + 1], w2 = wd[3 * kk + 2];
      const float4 m = m3[kk];
      const float fp = lane_prev(m.w), fn = lane_next(m.x);
      const float l = leftok ? fp : 0.f, rr = rightok ? fn : 0.f;
      t0 = fmaf(w0, l, t0), t1 = fmaf(w0, m.x, t1), t2 = fmaf(w0, m.y, t2), t3 = fmaf(w0, m.z, t3);
      t0 = fmaf(w1, m.x, t0), t1 = fmaf(w1, m.y, t1), t2 = fmaf(w1, m.z, t2), t3 = fmaf(w1, m.w, t3);
      t0 = fmaf(w2, m.y, t0), t1 = fmaf(w2, m.z, t1), t2 = fmaf(w2, m.w, t2), t3 = fmaf(w2, rr, t3);
    }
    t[0] = fmaxf(t0, 0.f), t[1] = fmaxf(t1, 0.f), t[2] = fmaxf(t2, 0.f), t[3] = fmaxf(t3, 0.f);
  };
#pragma unroll
  for (int r = 0; r < 3; r++) win[0][r] = win[1][r] = make_float4(lane, 1.f, 2.f, 3.f);
  load_window(0, win[0]);
  load_window(1, win[1]);
  float tcur[4];
  dw_compute(win[0], 0, tcur);
  const long long t0 = clock64();
#pragma unroll 1
  for (int ks0 = 0; ks0 < KS; ks0 += 2) {
#pragma unroll
    for (int d = 0; d < 2; d++) {
      const int ks = ks0 + d;
      load_window(min(ks + 2, KS - 1), win[d]);
      const float w = (MODE & 4) ? s_w[ks * 64 + lane] : wreg[d];
      float tnext[4];
      dw_compute(win[(d + 1) & 1], min(ks + 1, KS - 1), tnext);
      if (MODE & 16) {  // no software pipeline: the MFMAs consume what was just computed (a VALU block, then an MFMA block)
#pragma unroll
        for (int p = 0; p < 4; p++) tcur[p] = tnext[p];
      }
      if (MODE & 2) {
#pragma unroll
        for (int p = 0; p < 4; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, tcur[p], acc[p], 0, 0, 0);
      } else {
#pragma unroll
        for (int p = 0; p < 4; p++) acc[p][0] += w * tcur[p];
      }
#pragma unroll
      for (int p = 0; p < 4; p++) tcur[p] = tnext[p];
    }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int p = 0; p < 4; p++)
    for (int r = 0; r < 16; r++) s += acc[p][r];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 300) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, const float* in, const float* w, float* out, long long* cyc, size_t chan_step) {
  const size_t lds = 72 * 1024;  // two blocks per CU: two waves per SIMD
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<2560, 256, lds>>>(in, w, out, cyc, chan_step);
  (void)hipEventRecord(e0);
  k<MODE><<<2560, 256, lds>>>(in, w, out, cyc, chan_step);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h = 0; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s %7.1f cycles per k-step per wave   kernel %.1f us\n", name, (double)h / KS, ms * 1e3);
}
int main() {
  float *in, *w, *out; long long* cyc;
  const size_t chan_step = 2560ull * 4 * 64 * 4;  // floats between the channels of consecutive k-steps
  (void)hipMalloc(&in, (chan_step * KS + 4ull * 1024 * 1024) * sizeof(float));
  (void)hipMemset(in, 0, (chan_step * KS + 4ull * 1024 * 1024) * sizeof(float));
  (void)hipMalloc(&w, 4096); (void)hipMemset(w, 0, 4096);
  (void)hipMalloc(&out, 2560 * 256 * 4); (void)hipMalloc(&cyc, 8);
  run<2>("mfma", in, w, out, cyc, chan_step);
  run<1>("dw", in, w, out, cyc, chan_step);
  run<3>("dw + mfma", in, w, out, cyc, chan_step);
  run<7>("dw + mfma + lds weights", in, w, out, cyc, chan_step);
  run<11>("dw + mfma + global windows", in, w, out, cyc, chan_step);
  run<15>("dw + mfma + lds + global", in, w, out, cyc, chan_step);
  run<14>("mfma + lds + global (no dw)", in, w, out, cyc, chan_step);
  run<13>("dw + lds + global (no mfma)", in, w, out, cyc, chan_step);
  run<19>("dw + mfma, not pipelined", in, w, out, cyc, chan_step);
  run<23>("dw + mfma + lds, not pipelined", in, w, out, cyc, chan_step);
  run<31>("all, not pipelined", in, w, out, cyc, chan_step);
  return 0;
}
