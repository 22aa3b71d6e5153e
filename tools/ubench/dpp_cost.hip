// Issue cost of cross-lane moves on gfx950, one wave per SIMD and two: cycles per instruction from s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 -o dpp_cost dpp_cost.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 1.0f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      float x0, x1, x2, x3;
      if (MODE == 0) {  // plain fma (baseline)
        x0 = a, x1 = b, x2 = c, x3 = d;
      } else if (MODE == 1) {  // wave_shr:1
        x0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x138, 0xf, 0xf, true));
        x1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), 0x138, 0xf, 0xf, true));
        x2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), 0x130, 0xf, 0xf, true));
        x3 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x130, 0xf, 0xf, true));
      } else if (MODE == 2) {  // row_shr:1
        x0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x111, 0xf, 0xf, true));
        x1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), 0x111, 0xf, 0xf, true));
        x2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), 0x101, 0xf, 0xf, true));
        x3 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x101, 0xf, 0xf, true));
      } else if (MODE == 3) {  // ds_bpermute (shfl_up)
        x0 = __shfl_up(a, 1), x1 = __shfl_up(b, 1), x2 = __shfl_down(c, 1), x3 = __shfl_down(d, 1);
      } else {  // ds_swizzle-free: readlane-free permlane? use __builtin_amdgcn_mov_dpp8
        x0 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a), 0x8000 | 0x93));  // quad perm
        x1 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(b), 0x8000 | 0x93));
        x2 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(c), 0x8000 | 0x39));
        x3 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(d), 0x8000 | 0x39));
      }
      s0 = fmaf(x0, 1.0001f, s0), s1 = fmaf(x1, 1.0001f, s1), s2 = fmaf(x2, 1.0001f, s2), s3 = fmaf(x3, 1.0001f, s3);
      a += 1.f, b += 1.f, c += 1.f, d += 1.f;
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1 + s2 + s3;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const int iters = 1000;
  k<MODE><<<1, threads>>>(out, cyc, iters);
  k<MODE><<<1, threads>>>(out, cyc, iters);
  long long h = 0; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-12s threads %4d: %.1f cycles per group of (4 moves + 8 valu)\n", name, threads, (double)h / (iters * 8));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int threads : {64, 256, 512}) {
    run<0>("fma only", threads); run<1>("wave_shr", threads); run<2>("row_shr", threads);
    run<3>("bpermute", threads); run<4>("swizzle", threads);
  }
  return 0;
}
