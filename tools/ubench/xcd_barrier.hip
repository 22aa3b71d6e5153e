// What does a barrier between the workgroups of ONE XCD cost on MI355X?  (a persistent network kernel for one
// or two frames would replace ~28 launches of ~4.5 us by barriers -- if a barrier inside one XCD, whose 32 CUs share one L2,
// is much cheaper than a launch boundary, which has to make every XCD's stores visible to every other XCD.)
//
// 256 workgroups are launched (dealt round-robin over the 8 XCDs); each reads HW_REG_XCC_ID; those not on the target XCD
// leave at once; the others (32, one per CU) count themselves and then run ROUNDS barrier rounds.  In every round a workgroup
// writes a stamp (round, rank) to its slot of a buffer in global memory, passes the barrier, and checks the stamp its
// neighbour (rank + 1) wrote in the same round -- a barrier that does not make the data visible shows up as errors.
// Barrier = one atomic add per workgroup on a counter in global memory + spinning on it, between two __syncthreads().
// Fence flavours:
//   0  __threadfence() either side (agent scope: buffer_wbl2 sc1 + buffer_inv sc1 -- what a grid barrier over the whole GPU needs)
//   1  release = wait for the stores to be acknowledged by L2 (s_waitcnt vmcnt(0): the vector L1 is write-through), acquire =
//      buffer_inv sc1 (agent-scope invalidate: the vector L1 and, with several XCDs, the L2's non-coherent lines); no L2 write-back
//   2  no fences at all, data read with a plain load (expected to FAIL the check: the control)
//   3  inside one XCD: release as 1, acquire = buffer_inv sc0 (the CU's vector L1 only)
//   4  inside one XCD: release as 1, NO invalidate; the data is read with loads that bypass the vector L1 (agent-scope relaxed
//      atomic loads = global_load ... sc1)
// The stamps are double-buffered by round parity: a neighbour may run ahead into round r + 1 while this workgroup still reads
// round r's slot (it cannot reach round r + 2 before this workgroup has arrived at barrier r + 1).
// Also: all 256 workgroups over the 8 XCDs with flavour 0 (the whole-GPU barrier), and a kernel-launch boundary for scale
// (empty kernels back to back on one stream).
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_barrier xcd_barrier.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Ctl {
  unsigned joined;      // workgroups that stayed
  unsigned bar;         // barrier counter (monotonic)
  unsigned errors;      // stale stamps seen
  unsigned xcc_seen[8]; // workgroups per XCD
  unsigned long long cycles;  // s_memtime of rank 0 over all rounds
  unsigned long long wall;    // wall_clock64 (100 MHz) of rank 0 over all rounds
};

constexpr unsigned kSpinLimit = 1u << 22;  // ~a few ms of polling

__device__ __forceinline__ unsigned xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
  return x;
}

template <int FENCE>
__device__ __forceinline__ void release_fence() {
  if (FENCE == 0) __threadfence();
  if (FENCE == 1 || FENCE == 3 || FENCE == 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}
template <int FENCE>
__device__ __forceinline__ void acquire_fence() {
  if (FENCE == 0) __threadfence();
  if (FENCE == 1) asm volatile("buffer_inv sc1" ::: "memory");
  if (FENCE == 3) asm volatile("buffer_inv sc0" ::: "memory");
}

// target < 0: every workgroup takes part (whole-GPU barrier); expect = number of participants
template <int FENCE>
__global__ __launch_bounds__(256) void k_barrier(Ctl* c, unsigned* stamps, int target, unsigned expect, int rounds) {
  __shared__ unsigned s_rank;
  const unsigned x = xcc_id();
  if (threadIdx.x == 0) atomicAdd(&c->xcc_seen[x & 7], 1u);
  if (target >= 0 && (int)x != target) return;
  if (threadIdx.x == 0) s_rank = atomicAdd(&c->joined, 1u);
  __syncthreads();
  const unsigned rank = s_rank;
  if (rank >= expect) return;  // (more workgroups on the XCD than expected: the extra ones leave, the barrier counts `expect`)
  // wait until all participants have joined (so that the timed rounds start together)
  // (every spin below gives up after kSpinLimit polls and the kernel drains: a barrier whose participants are not all
  // resident -- fewer workgroups on the XCD than expected -- must not hang the GPU)
  __shared__ int s_dead;
  if (threadIdx.x == 0) {
    s_dead = 0;
    unsigned spins = 0;
    while (__hip_atomic_load(&c->joined, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect && ++spins < kSpinLimit) __builtin_amdgcn_s_sleep(1);
    if (spins >= kSpinLimit) s_dead = 1;
  }
  __syncthreads();
  if (s_dead) {
    if (threadIdx.x == 0) atomicAdd(&c->errors, 1u << 30);
    return;
  }
  unsigned errors = 0;
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int r = 0; r < rounds; r++) {
    // every thread writes a word of the workgroup's 1 KB slot (a "tensor row" the neighbour will read)
    unsigned* slots = stamps + (r & 1) * 256 * 256;
    slots[rank * 256 + threadIdx.x] = (unsigned)(r + 1) * 65536u + rank;
    release_fence<FENCE>();
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(&c->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = expect * (unsigned)(r + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(&c->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < kSpinLimit) {
      }
      if (spins >= kSpinLimit) s_dead = 1;
    }
    __syncthreads();
    if (s_dead) {
      if (threadIdx.x == 0) atomicAdd(&c->errors, 1u << 30);
      return;
    }
    acquire_fence<FENCE>();
    const unsigned nb = (rank + 1) % expect;
    const unsigned got = FENCE == 4 ? __hip_atomic_load(&slots[nb * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : slots[nb * 256 + threadIdx.x];
    if (got != (unsigned)(r + 1) * 65536u + nb) errors++;
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (errors) atomicAdd(&c->errors, errors);
  if (rank == 0 && threadIdx.x == 0) c->cycles = t1 - t0, c->wall = w1 - w0;
}

__global__ void k_empty() {}

template <int FENCE>
static void run(const char* what, int target, unsigned expect, int rounds) {
  Ctl* c;
  unsigned* stamps;
  hipMalloc(&c, sizeof(Ctl));
  hipMalloc(&stamps, 2 * 256 * 256 * sizeof(unsigned));
  hipMemset(c, 0, sizeof(Ctl));
  hipMemset(stamps, 0, 2 * 256 * 256 * sizeof(unsigned));
  hipLaunchKernelGGL(k_barrier<FENCE>, dim3(256), dim3(256), 0, 0, c, stamps, target, expect, rounds);
  if (hipDeviceSynchronize() != hipSuccess) {
    std::printf("%s: kernel failed\n", what);
    return;
  }
  Ctl h;
  hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost);
  std::printf("%-58s participants %3u  %8.1f ns / barrier (wall clock)  %7.0f cycles  stale reads %u\n", what, h.joined < expect ? h.joined : expect,
              (double)h.wall * 10.0 / rounds, (double)h.cycles / rounds, h.errors);
  if (target == 0 && FENCE == 0) {
    std::printf("  workgroups per XCD:");
    for (int i = 0; i < 8; i++) std::printf(" %u", h.xcc_seen[i]);
    std::printf("\n");
  }
  hipFree(c), hipFree(stamps);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 2000;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  std::printf("%s, %d CUs, %d rounds\n", p.name, p.multiProcessorCount, rounds);
  run<0>("one XCD, 32 workgroups, agent-scope fences", 0, 32, rounds);
  run<1>("one XCD, 32 workgroups, L2-ack release + buffer_inv sc1", 0, 32, rounds);
  run<3>("one XCD, 32 workgroups, L2-ack release + buffer_inv sc0", 0, 32, rounds);
  run<4>("one XCD, 32 workgroups, L2-ack release + L1-bypassing loads", 0, 32, rounds);
  run<2>("one XCD, 32 workgroups, no fences (control)", 0, 32, rounds);
  run<3>("one XCD, 8 workgroups, L2-ack release + buffer_inv sc0", 0, 8, rounds);
  run<4>("one XCD, 8 workgroups, L2-ack release + L1-bypassing loads", 0, 8, rounds);
  run<0>("whole GPU, 256 workgroups, agent-scope fences", -1, 256, rounds);
  run<3>("whole GPU, 256 workgroups, XCD-local fences (must fail)", -1, 256, rounds);
  // launch boundary: empty kernels back to back
  hipStream_t s;
  hipStreamCreate(&s);
  for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
  hipStreamSynchronize(s);
  const int n = 5000;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_empty, dim3(32), dim3(256), 0, s);
  hipStreamSynchronize(s);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  std::printf("%-58s %8.1f ns / launch (empty 32-workgroup kernels back to back on one stream)\n", "launch boundary", us * 1e3 / n);
  return 0;
}
