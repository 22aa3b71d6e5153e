// Does gfx950 (as this pool configures it) execute a 16-byte global store to an address that is not 16-byte aligned?
// hipcc emits one global_store_dwordx4 for a store through a packed struct (unaligned-access-mode is on for amdhsa);
// this checks the result for every misalignment 0..15.  Build: hipcc -O3 --offload-arch=gfx950 unaligned_store.hip -o unaligned_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
struct __attribute__((packed, aligned(1))) U16 { uint32_t w[4]; };
__global__ void k(unsigned char* dst, const uint4* src, int sh, int n) {
  const int t = threadIdx.x + blockIdx.x * blockDim.x;
  if (t >= n) return;
  const uint4 v = src[t];
  const U16 u = {{v.x, v.y, v.z, v.w}};
  *reinterpret_cast<U16*>(dst + (size_t)t * 16 + sh) = u;
}
int main() {
  const int n = 1 << 16;
  std::vector<unsigned char> h((size_t)n * 16), out((size_t)n * 16 + 64);
  for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)(i * 131 + (i >> 8));
  unsigned char *d_src, *d_dst;
  hipMalloc(&d_src, h.size());
  hipMalloc(&d_dst, out.size());
  hipMemcpy(d_src, h.data(), h.size(), hipMemcpyHostToDevice);
  int bad = 0;
  for (int sh = 0; sh < 16; sh++) {
    hipMemset(d_dst, 0xEE, out.size());
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d_dst, (const uint4*)d_src, sh, n);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("launch failed at shift %d\n", sh); return 2; }
    hipMemcpy(out.data(), d_dst, out.size(), hipMemcpyDeviceToHost);
    if (std::memcmp(out.data() + sh, h.data(), h.size()) != 0) bad++, std::printf("shift %d: WRONG\n", sh);
    for (int i = 0; i < sh; i++) if (out[i] != 0xEE) bad++;
    for (size_t i = h.size() + sh; i < out.size(); i++) if (out[i] != 0xEE) bad++;
  }
  std::printf(bad ? "unaligned 16-byte stores: %d failures\n" : "unaligned 16-byte stores: ok for shifts 0..15\n", bad);
  return bad ? 1 : 0;
}
