// pipeline_gate.cpp -- keeps the GPU-filling stretch of consecutive batches from running in lock step (round 6).
//
// A handle rotates its batches over four contexts (four streams), so that the latency-bound stages of one batch (entropy
// chain, small-map layers, NMS) run beside the GPU-filling ones of the others (stem, m1->m2, m3->m4, m5, m6, the RFB
// block).  After an idle period -- the first batches of a server, the driver's 20-step sample -- the four contexts enter
// their entropy chains together, reach the stem together, share the GPU-filling kernels four ways and finish together: the
// next four batches start in lock step again, and every round of four leaves the GPU to one latency-bound stage for a
// quarter of a millisecond (20-step sample 55.6 k frames/s for 61.2 k steady state on one box, unchanged since round 3).
// The gate orders ONE thing across contexts: batch n + 1 does not begin its network (the stem) before batch n has passed
// `gate_layer` (default: m4.pw -- behind the stem, m1->m2 and m3->m4, the three launches that take a CU whole).  Everything
// in front of the stem (staging, entropy chain, IDCT) and everything behind gate_layer still overlaps freely; results are
// untouched (the order of independent kernels).  Those three launches take ~180 us of a 500 us period, so in steady state
// the wait is already over when a batch gets there.  Measured (tools/ab/r6_gate.py, profiles/r6b/gate_layers.txt; median of
// twelve 20-step samples | 300-step steady state, frames/s, one box): no gate 59 958 (56.8 - 61.3 k) | 65 111; behind the stem
// 58 806 | 63 772; behind m1->m2 61 212 | 64 784; behind m3->m4 61 675 (61.3 - 62.1 k) | 65 021; behind m6 60 848 | 63 972;
// behind the RFB block 57 258 | 60 335; behind m10 51 680 | 53 386 -- the further back, the more of the latency-bound layers
// are serialised too.  Batches that go in and out by kernels (a frame or a few at a time) are not gated.
//
// Mechanism: a ring of events indexed by the batch's sequence number (given at submit, under the handle lock).  The worker
// issuing batch n records event[n] behind the gate layer's launch and publishes it; the worker issuing batch n + 1 waits
// (host side, bounded) until event[n] is published and puts a hipStreamWaitEvent in front of its stem.  A batch that issues
// no network (nothing decodable, an error) publishes "nothing to wait for".  Synchronous entry points (sequence 0) neither
// wait nor record.
#include <chrono>
#include <thread>

#include "experiments.hpp"
#include "model_parts.hpp"

namespace ufd {

int gate_init(ufd_model* m) {
  PipelineGate& g = m->gate;
  g.layer = kGateDefaultLayer;
  if (const char* e = experiment_env("UFD_GATE_LAYER")) g.layer = std::atoi(e);  // (measurement build: < 0 turns the gate off)
  if (m->cfg.flags & UFD_FLAG_NO_GATE) g.layer = -1;
  if (g.layer < 0 || m->num_ctx < 2) {
    g.layer = -1;
    return UFD_OK;
  }
  for (int i = 0; i < PipelineGate::kRing; i++) {
    if (hipEventCreateWithFlags(&g.ev[i], hipEventDisableTiming) != hipSuccess) return UFD_E_DEVICE;
    g.published[i].store(0, std::memory_order_relaxed);
  }
  return UFD_OK;
}

void gate_destroy(ufd_model* m) {
  for (int i = 0; i < PipelineGate::kRing; i++)
    if (m->gate.ev[i]) (void)hipEventDestroy(m->gate.ev[i]), m->gate.ev[i] = nullptr;
}

// In front of the batch's first network launch, on the issuing worker's thread.
void gate_wait_for_previous(ufd_model* m, const Slot& s, hipStream_t st) {
  PipelineGate& g = m->gate;
  if (g.layer < 0 || s.seq < 2 || s.small_batch) return;
  const uint64_t prev = s.seq - 1;
  const int i = (int)(prev % PipelineGate::kRing);
  // published[i] = 2 * seq + (1: an event was recorded | 0: nothing to wait for).  Bounded: a worker that died on an
  // exception never publishes, and an ordering hint must never hang a handle.
  const auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  for (;;) {
    const uint64_t p = g.published[i].load(std::memory_order_acquire);
    if ((p >> 1) == prev) {
      if (p & 1) (void)hipStreamWaitEvent(st, g.ev[i], 0), g.waits.fetch_add(1, std::memory_order_relaxed);
      return;
    }
    if ((p >> 1) > prev) return;  // the ring has moved on: that batch is long past its gate
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
      g.timeouts.fetch_add(1, std::memory_order_relaxed);
      return;
    }
    // (the other worker is a few launches away from its gate layer: yield first; if it is busy for longer -- a host-entropy
    // batch decoding on the pool -- sleep, so that a rank with two CPUs does not lose one of them to this loop)
    if (++spins < 200) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
}

// Behind the launch that covers layer i.
void gate_pass(ufd_model* m, Slot& s, int i, hipStream_t st) {
  PipelineGate& g = m->gate;
  if (g.layer < 0 || s.seq == 0 || i != g.layer || s.gate_published || s.small_batch) return;
  const int k = (int)(s.seq % PipelineGate::kRing);
  const bool ok = hipEventRecord(g.ev[k], st) == hipSuccess;
  g.published[k].store(2 * s.seq + (ok ? 1 : 0), std::memory_order_release);
  s.gate_published = true;
}

// When the worker is done with the batch, whatever happened: a batch that never reached the gate layer must not be waited for.
void gate_close(ufd_model* m, Slot& s) {
  PipelineGate& g = m->gate;
  if (g.layer < 0 || s.seq == 0 || s.gate_published) return;
  g.published[s.seq % PipelineGate::kRing].store(2 * s.seq, std::memory_order_release);
  s.gate_published = true;
}

}  // namespace ufd
