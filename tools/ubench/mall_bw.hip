// Where do re-reads land on MI355X?  Read bandwidth of a buffer that is streamed over and over by the whole GPU, as a
// function of its size: up to ~32 MB it fits the eight L2s together (4 MB each; every XCD reads its own contiguous eighth, like
// the tile order of the conv kernels), up to ~256 MB the Infinity Cache (MALL), beyond that HBM.  rocprofv3 on this box has no
// counter for Infinity-Cache hits (--list-avail: no MALL / DF events; TCC_EA0_RDREQ_DRAM counts requests ADDRESSED to local
// memory, hit or miss), so this curve -- and the FETCH_SIZE of the same passes -- is the evidence for the claim that a
// kernel's re-reads within a few hundred MB are fabric-cache hits, not HBM reads (docs/EXPERIMENTS.md, round 5).
// Build: hipcc --offload-arch=gfx950 -O3 -o mall_bw mall_bw.hip ; run on the GPU box (prints one line per size).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ src, size_t n16, float* sink, int passes) {
  // XCD x (blockIdx.x & 7) streams its own eighth of the buffer, its workgroups side by side
  const size_t per_xcd = n16 / 8;
  const size_t wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
  const float4* base = src + (blockIdx.x & 7) * per_xcd;
  float acc = 0.f;
  for (int p = 0; p < passes; p++)
    for (size_t i = wg_in_xcd * 256 + threadIdx.x; i < per_xcd; i += wgs_per_xcd * 256) {
      const float4 v = base[i];
      acc += v.x + v.y + v.z + v.w;
    }
  if (acc == 12345.678f) sink[0] = acc;  // (keeps the loads)
}

int main() {
  const size_t max_bytes = (size_t)2 << 30;
  float4* buf;
  float* sink;
  if (hipMalloc(&buf, max_bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return std::printf("alloc failed\n"), 1;
  (void)hipMemset(buf, 0, max_bytes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  const size_t mb[] = {8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 2048};
  std::printf("%10s %10s %12s\n", "MB", "passes", "GB/s");
  for (size_t m : mb) {
    const size_t bytes = m << 20, n16 = bytes / 16;
    const int passes = (int)std::max<size_t>(2, (size_t)8192 / m);  // ~8 GB of reads per point
    hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, buf, n16, sink, 1);  // warm
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, buf, n16, sink, passes);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::printf("%10zu %10d %12.1f\n", m, passes, (double)bytes * passes / (ms * 1e-3) / 1e9);
  }
  return 0;
}
