// stats.cpp -- what the library measures about itself: per-kernel device time (HIP events on the handle's own streams:
// ufd_profile_*, bench.py's roofline object), the device-time span of every batch on its context and the idle gap in front of
// it, and the always-on host counters behind ufd_host_stats_* (bench.py's `host` object; the reference's meter.rs for the GPU
// path).  Split from model.cpp in round 5.
#include "model_types.hpp"
#include "model_internal.hpp"
#include "model_parts.hpp"

// ---------------------------------------------------------------- profiling
namespace ufd {
thread_local LaunchShape tl_launch_shape;

int prof_name_id(ufd_model* m, const std::string& name) {
  for (size_t i = 0; i < m->prof_names.size(); i++)
    if (m->prof_names[i] == name) return (int)i;
  m->prof_names.push_back(name);
  ufd_kernel_stat st;
  std::memset(&st, 0, sizeof(st));
  std::snprintf(st.name, sizeof(st.name), "%s", name.c_str());
  m->prof_stats.push_back(st);
  return (int)m->prof_names.size() - 1;
}

hipEvent_t prof_event(ufd_model* m) {
  if (!m->prof_free.empty()) {
    hipEvent_t e = m->prof_free.back();
    m->prof_free.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace ufd


namespace ufd {
void prof_flush(ufd_model* m) {
  std::lock_guard<std::mutex> lk(m->shared_mu);
  for (auto& pe : m->prof_pending) {
    float ms = 0;
    if (hipEventSynchronize(pe.e1) == hipSuccess && hipEventElapsedTime(&ms, pe.e0, pe.e1) == hipSuccess) {
      auto& st = m->prof_stats[pe.name_id];
      st.launches++;
      st.total_ms += ms;
      st.bytes += pe.bytes;
      st.flops += pe.flops;
    }
    m->prof_free.push_back(pe.e0);
    m->prof_free.push_back(pe.e1);
  }
  m->prof_pending.clear();
}


// ---------------------------------------------------------------- spans
// Host statistics: the batch's first kernel is about to be enqueued on the context's stream (its H2D, if any, is already
// waited for on that stream) -- the time between the previous batch's end event and this one is time the stream had
// nothing to run.
void span_begin(Slot& s) {
  Ctx& c = *tl_cur;
  if (s.span_idx >= 0 || !c.ev_span[0][0]) return;
  s.span_seq = c.span_issued++;
  s.span_idx = (int)(s.span_seq % Ctx::kSpanRing);
  (void)hipEventRecord(c.ev_span[s.span_idx][0], c.stream);
}

// ... and its last operation has been enqueued
void span_end(Slot& s) {
  if (s.span_idx >= 0) (void)hipEventRecord(tl_cur->ev_span[s.span_idx][1], tl_cur->stream);
}

// The slot's batch is complete: fold its span (and the gap in front of it) into the context's sums.
void span_fold(ufd_model* m, Slot& s) {
  if (s.span_idx < 0 || !s.ctx) return;
  Ctx& c = *s.ctx;
  float span = 0, gap = 0;
  const int prev = (int)((s.span_seq + Ctx::kSpanRing - 1) % Ctx::kSpanRing);
  const bool ok = hipEventElapsedTime(&span, c.ev_span[s.span_idx][0], c.ev_span[s.span_idx][1]) == hipSuccess;
  std::lock_guard<std::mutex> lk(m->shared_mu);
  // (a gap only between consecutive batches of the context, both since the last reset)
  const bool have_prev = s.span_seq > 0 && c.span_last_done == s.span_seq &&
                         hipEventElapsedTime(&gap, c.ev_span[prev][1], c.ev_span[s.span_idx][0]) == hipSuccess;
  if (ok) c.gpu_batches++, c.gpu_span_ms += span;
  if (ok && have_prev && c.gpu_batches > 1) c.gpu_gap_ms += std::max(gap, 0.0f);
  c.span_last_done = s.span_seq + 1;
  s.span_idx = -1;
}

}  // namespace ufd

using namespace ufd;

extern "C" {

int ufd_host_stats_reset(ufd_model* m) {
  if (!m) return UFD_E_ARG;
  std::lock_guard<std::mutex> lk(m->shared_mu);
  for (int c = 0; c < m->num_ctx; c++) {
    Worker& w = m->workers[c];
    w.ns_busy = 0, w.ns_plan = 0, w.ns_copy = 0, w.batches = 0, w.launches = 0;
    m->ctx[c].gpu_batches = 0, m->ctx[c].gpu_span_ms = 0, m->ctx[c].gpu_gap_ms = 0;
  }
  m->ns_wait = 0, m->waits = 0;
  m->stats_t0 = now_ns();
  return UFD_OK;
}

int ufd_host_stats_read(ufd_model* m, ufd_host_stats* out) {
  if (!m || !out || out->struct_size != sizeof(ufd_host_stats)) return UFD_E_ARG;
  std::memset(out, 0, sizeof(*out));
  out->struct_size = sizeof(*out);
  std::lock_guard<std::mutex> lk(m->shared_mu);
  out->num_ctx = (uint32_t)m->num_ctx;
  out->wall_ms = (double)(now_ns() - m->stats_t0) * 1e-6;
  uint64_t busy = 0, plan = 0, copy = 0;
  for (int c = 0; c < m->num_ctx; c++) {
    const Worker& w = m->workers[c];
    busy += w.ns_busy, plan += w.ns_plan, copy += w.ns_copy;
    out->batches += w.batches, out->launches += w.launches;
    out->worker_busy_ms[c] = (double)w.ns_busy * 1e-6;
    out->gpu_batches[c] = m->ctx[c].gpu_batches;
    out->gpu_span_ms[c] = m->ctx[c].gpu_span_ms;
    out->gpu_gap_ms[c] = m->ctx[c].gpu_gap_ms;
  }
  out->plan_ms = (double)plan * 1e-6, out->copy_ms = (double)copy * 1e-6;
  out->issue_ms = (double)(busy - std::min(busy, plan + copy)) * 1e-6;
  out->waits = m->waits, out->wait_ms = (double)m->ns_wait * 1e-6;
  return UFD_OK;
}

int ufd_profile_reset(ufd_model* m) {
  return guarded(m, [&]() -> int {
    HIPC(m, hipStreamSynchronize(tl_cur->stream));
    prof_flush(m);
    for (auto& st : m->prof_stats) st.launches = 0, st.total_ms = 0, st.bytes = 0, st.flops = 0;
    return UFD_OK;
  });
}

int ufd_profile_sampling(ufd_model* m, uint32_t every_n) {
  return guarded(m, [&]() -> int {
    if (!every_n) return m->fail(UFD_E_ARG, "every_n must be >= 1");
    m->prof_every = every_n;
    m->prof_batch = 0;
    return UFD_OK;
  });
}

int ufd_profile_shapes(ufd_model* m, ufd_launch_shape* shapes, uint32_t cap, uint32_t* n) {
  return guarded(m, [&]() -> int {
    if (!n) return m->fail(UFD_E_ARG, "null argument");
    std::vector<LaunchShape> sh;
    std::vector<std::string> names;
    {
      std::lock_guard<std::mutex> lk(m->shared_mu);
      sh = m->prof_shapes;
      names = m->prof_names;
    }
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->cfg.device_id);
    uint32_t k = 0;
    for (size_t i = 0; i < sh.size(); i++) {
      if (!sh[i].fn || !sh[i].threads) continue;
      if (k < cap) {
        ufd_launch_shape& o = shapes[k];
        std::memset(&o, 0, sizeof(o));
        std::snprintf(o.name, sizeof(o.name), "%s", names[i].c_str());
        o.workgroups = sh[i].blocks, o.threads = sh[i].threads, o.compute_units = (uint32_t)cus;
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, sh[i].fn) == hipSuccess) {
          o.registers = (uint32_t)fa.numRegs;
          o.lds_bytes = (uint32_t)fa.sharedSizeBytes + sh[i].lds;
        } else {
          o.lds_bytes = sh[i].lds;
        }
        int res = 0;  // the runtime's own occupancy rule: registers, LDS, waves and workgroup slots of a CU
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&res, sh[i].fn, (int)sh[i].threads, sh[i].lds) == hipSuccess)
          o.resident_per_cu = (uint32_t)res;
      }
      k++;
    }
    *n = k;
    return UFD_OK;
  });
}

int ufd_profile_read(ufd_model* m, ufd_kernel_stat* stats, uint32_t cap, uint32_t* n) {
  return guarded(m, [&]() -> int {
    if (!n) return m->fail(UFD_E_ARG, "null argument");
    HIPC(m, hipStreamSynchronize(tl_cur->stream));
    prof_flush(m);
    *n = (uint32_t)m->prof_stats.size();
    for (uint32_t i = 0; i < std::min<uint32_t>(cap, *n); i++) stats[i] = m->prof_stats[i];
    return UFD_OK;
  });
}

}  // extern "C"
