// The handle's per-batch call sequence on the HIP runtime alone (three compute streams, one shared copy stream, double-buffered
// landing zones with copied / consumed events, two D2H result copies, six batches in flight), to find which call makes the
// host-bytes path grow 2 KB of host memory per batch (tools/soak.py).  Flags switch single calls off:
//   argv[2] letters: c = skip the copy stream's wait on the consumed event, d = skip the D2H copies, q = wait with
//   hipEventSynchronize instead of polling hipEventQuery, s = H2D on the compute stream (no copy stream at all),
//   p = pageable (malloc) destination for the D2H copies instead of pinned, e = the HOST waits for the copied event (no
//   hipStreamWaitEvent on the compute stream), w = the host waits with hipStreamSynchronize(copy) (no copied event at all),
//   k = a tiny kernel on the copy stream between the copy and the event record
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/leak_probe2.hip -o tools/ubench/leak_probe2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <time.h>

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
  const std::string fl = argc > 2 ? argv[2] : "";
  auto has = [&](char c) { return fl.find(c) != std::string::npos; };
  const size_t bytes = 1100000;
  hipStream_t comp[3], copy;
  for (auto& s : comp) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&copy, hipStreamNonBlocking);
  char* h[8];
  for (auto& p : h) hipHostMalloc(&p, bytes, hipHostMallocDefault);
  char* d[3][2];
  for (auto& a : d)
    for (auto& p : a) hipMalloc(&p, bytes);
  int* di[3];
  for (auto& p : di) hipMalloc(&p, 4096), hipMemset(p, 0, 4096);
  char* hres[8];
  for (auto& p : hres) {
    if (has('p')) p = (char*)malloc(8192);
    else hipHostMalloc(&p, 8192, hipHostMallocDefault);
  }
  hipEvent_t copied[3][2], consumed[3][2], done[8];
  for (auto& a : copied)
    for (auto& e : a) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (auto& a : consumed)
    for (auto& e : a) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (auto& e : done) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  bool valid[3][2] = {};
  int flip[3] = {0, 0, 0};
  hipDeviceSynchronize();
  double r0 = rss_mb();
  for (int i = 0; i < iters; i++) {
    const int slot = i & 7, c = i % 3;
    if (i >= 6) {  // six in flight: wait for the oldest
      const int w = (i - 6) & 7;
      if (has('q')) hipEventSynchronize(done[w]);
      else
        while (hipEventQuery(done[w]) == hipErrorNotReady) {
          timespec ts{0, 20000};
          nanosleep(&ts, nullptr);
        }
    }
    const int buf = flip[c];
    flip[c] ^= 1;
    hipStream_t cs = has('s') ? comp[c] : copy;
    if (!has('s') && !has('c') && valid[c][buf]) hipStreamWaitEvent(copy, consumed[c][buf], 0);
    hipMemcpyAsync(d[c][buf], h[slot], bytes, hipMemcpyHostToDevice, cs);
    if (has('w')) {
      hipStreamSynchronize(copy);
    } else if (!has('s')) {
      if (has('k')) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, copy, di[c] + 512);
      hipEventRecord(copied[c][buf], copy);
      if (has('e')) hipEventSynchronize(copied[c][buf]);
      else hipStreamWaitEvent(comp[c], copied[c][buf], 0);
    }
    for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, comp[c], di[c]);
    hipEventRecord(consumed[c][buf], comp[c]);
    valid[c][buf] = true;
    for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, comp[c], di[c]);
    if (!has('d')) {
      hipMemcpyAsync(hres[slot], di[c], 256, hipMemcpyDeviceToHost, comp[c]);
      hipMemcpy2DAsync(hres[slot] + 256, 160, di[c], 128, 100, 32, hipMemcpyDeviceToHost, comp[c]);
    }
    hipEventRecord(done[slot], comp[c]);
    if ((i + 1) % (iters / 4) == 0) {
      printf("flags '%s': %d iterations, rss %+.1f MB (%.0f bytes per iteration in this quarter)\n", fl.c_str(), i + 1, rss_mb() - r0,
             (rss_mb() - r0) * 1048576.0 / (iters / 4));
      fflush(stdout);
      r0 = rss_mb();
    }
  }
  hipDeviceSynchronize();
  return 0;
}
