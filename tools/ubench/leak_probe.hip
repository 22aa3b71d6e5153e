// Host-memory growth of the HIP runtime under the pipeline's per-batch call patterns (round 4: the soak run grew 2 KB per batch
// on the host-bytes path and not on the staged one).  Patterns, 200 000 iterations each, resident set size before / after:
//   A  hipMemcpyAsync(pinned -> device, copy stream) + hipEventRecord(ev, copy) + hipStreamWaitEvent(compute, ev) + kernel +
//      hipEventRecord(done, compute) + hipEventSynchronize(done)            -- the host-bytes path
//   B  the same copy on the compute stream, no cross-stream wait
//   C  kernel + event only                                                    -- the staged path
//   D  pattern A with a DIFFERENT pinned source address every iteration (a ring of 8 buffers, as the slots are)
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/leak_probe.hip -o /tmp/leak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>

__global__ void k_touch(int* p) { if (threadIdx.x == 0) p[0] += 1; }

static double rss_mb() {
  std::ifstream f("/proc/self/status");
  std::string l;
  while (std::getline(f, l))
    if (l.rfind("VmRSS", 0) == 0) return std::stod(l.substr(6)) / 1024.0;
  return 0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200000;
  const size_t bytes = 1 << 20;
  hipStream_t comp, copy;
  hipStreamCreateWithFlags(&comp, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&copy, hipStreamNonBlocking);
  char* h[8];
  for (auto& p : h) hipHostMalloc(&p, bytes, hipHostMallocDefault);
  char* d;
  hipMalloc(&d, bytes);
  int* di;
  hipMalloc(&di, 64);
  hipMemset(di, 0, 64);
  hipEvent_t ev, done;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  hipEventCreateWithFlags(&done, hipEventDisableTiming);
  const char* names = "ABCD";
  const char* order = argc > 2 ? argv[2] : "ABCD";  // e.g. "DA": a pattern's growth may be another's reuse
  for (const char* o = order; *o; o++) {
    const int pat = (int)(strchr(names, *o) - names);
    hipDeviceSynchronize();
    const double r0 = rss_mb();
    for (int i = 0; i < iters; i++) {
      if (pat == 0 || pat == 3) {
        hipMemcpyAsync(d, h[pat == 3 ? i & 7 : 0], bytes, hipMemcpyHostToDevice, copy);
        hipEventRecord(ev, copy);
        hipStreamWaitEvent(comp, ev, 0);
      } else if (pat == 1) {
        hipMemcpyAsync(d, h[0], bytes, hipMemcpyHostToDevice, comp);
      }
      hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, comp, di);
      hipEventRecord(done, comp);
      hipEventSynchronize(done);
    }
    hipDeviceSynchronize();
    printf("pattern %c: %d iterations, rss %+.1f MB (%.0f bytes per iteration)\n", names[pat], iters, rss_mb() - r0, (rss_mb() - r0) * 1048576.0 / iters);
    fflush(stdout);
  }
  return 0;
}
