// The multi-stream scheduler (infercam_onnx_amd/csrc/sched.cpp: router.rs:64-71 + inferer.rs:29-50 for many streams) linked
// against a MOCK of the handle entry points it calls, so that its locking and bookkeeping run on the CPU under
// ThreadSanitizer and AddressSanitizer + UBSan (tests/test_sched.py builds and runs both; GPU sanitizers are not available
// on the pool, and sched.cpp makes no HIP call of its own).  The mock "model" sleeps a little in ufd_wait and returns
// detections derived from the JPEG bytes, so every delivered frame can be checked without a GPU:
//   n = jpeg[0] % 4 detections, dets[k].conf = jpeg[1] / 255, x_tl = len / 1e6.
// Scenarios: concurrent producers on their own streams, streams that come and go meanwhile (removal with frames queued /
// in flight / being copied), stale handles, drop-on-full, flush, destroy with work queued.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ufd.h"

#if defined(__SANITIZE_THREAD__)
// GCC 11's libtsan does not intercept pthread_cond_clockwait (what condition_variable::wait_until on steady_clock calls):
// it then misses the unlock / relock inside the wait and reports "double lock" plus a flood of false races behind it.
// Route it to pthread_cond_timedwait, which is intercepted (the executable's definition is the one libstdc++ binds to).
#include <pthread.h>
#include <time.h>
extern "C" int pthread_cond_clockwait(pthread_cond_t* c, pthread_mutex_t* m, clockid_t clock, const struct timespec* abstime) {
  struct timespec now_c, now_r, t;
  clock_gettime(clock, &now_c);
  clock_gettime(CLOCK_REALTIME, &now_r);
  long long ns = (abstime->tv_sec - now_c.tv_sec) * 1000000000ll + (abstime->tv_nsec - now_c.tv_nsec);
  if (ns < 0) ns = 0;
  ns += now_r.tv_sec * 1000000000ll + now_r.tv_nsec;
  t.tv_sec = ns / 1000000000ll, t.tv_nsec = ns % 1000000000ll;
  return pthread_cond_timedwait(c, m, &t);
}
#endif

// ------------------------------------------------------------------ mock of the handle side
struct ufd_model {
  uint32_t width, max_batch;
  std::mutex mu;
  struct Job {
    const uint8_t* const* jpegs;
    const size_t* lens;
    uint32_t count, cap;
    ufd_det* out;
    uint32_t* n;
    int32_t* status;
    bool annot;
    ufd_annotate a;
  };
  std::map<uint32_t, Job> jobs;
  uint32_t next = 1;
  std::atomic<uint64_t> frames{0};
};

extern "C" {
int ufd_model_info(const ufd_model* m, uint32_t* w, uint32_t* h, uint32_t* k) {
  if (w) *w = m->width;
  if (h) *h = m->width * 3 / 4;
  if (k) *k = 4420;
  return UFD_OK;
}
int ufd_model_limits(const ufd_model* m, uint32_t* mb, uint32_t* mw, uint32_t* mh) {
  if (mb) *mb = m->max_batch;
  if (mw) *mw = 1920;
  if (mh) *mh = 1088;
  return UFD_OK;
}
static int submit(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, uint32_t count, const ufd_annotate* a, ufd_det* out,
                  uint32_t cap, uint32_t* n, int32_t* status, uint32_t* ticket) {
  std::lock_guard<std::mutex> lk(m->mu);
  if (m->jobs.size() >= UFD_MAX_SLOTS) return UFD_E_STATE;
  ufd_model::Job j{jpegs, lens, count, cap, out, n, status, a != nullptr, a ? *a : ufd_annotate{}};
  *ticket = m->next++;
  m->jobs[*ticket] = j;
  return UFD_OK;
}
int ufd_submit_jpeg_batch(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, uint32_t count, ufd_det* out, uint32_t cap,
                          uint32_t* n, int32_t* status, uint32_t* ticket) {
  return submit(m, jpegs, lens, count, nullptr, out, cap, n, status, ticket);
}
int ufd_submit_annotate_batch(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, uint32_t count, const ufd_annotate* a,
                              ufd_det* out, uint32_t cap, uint32_t* n, int32_t* status, uint32_t* ticket) {
  return submit(m, jpegs, lens, count, a, out, cap, n, status, ticket);
}
int ufd_wait(ufd_model* m, uint32_t ticket) {
  ufd_model::Job j;
  {
    std::lock_guard<std::mutex> lk(m->mu);
    auto it = m->jobs.find(ticket);
    if (it == m->jobs.end()) return UFD_E_STATE;
    j = it->second;
  }
  std::this_thread::sleep_for(std::chrono::microseconds(200));  // "the GPU"
  size_t off = 0;
  for (uint32_t i = 0; i < j.count; i++) {  // reads the borrowed JPEG bytes: a slot recycled too early is caught here
    const uint8_t* p = j.jpegs[i];
    const size_t len = j.lens[i];
    const bool bad = len < 2 || p[0] == 0xEE;
    j.status[i] = bad ? UFD_E_DECODE : UFD_OK;
    j.n[i] = bad ? 0 : p[0] % 4;
    for (uint32_t k = 0; k < j.n[i] && k < j.cap; k++) j.out[(size_t)i * j.cap + k] = ufd_det{(float)len / 1e6f, 0, 1, 1, p[1] / 255.0f};
    if (j.annot) {
      j.a.jpeg_off[i] = off, j.a.jpeg_len[i] = 0;
      if (!bad && off + len <= j.a.jpeg_cap) {  // "annotated stream" = a copy of the input
        std::memcpy(j.a.jpeg_out + off, p, len);
        j.a.jpeg_len[i] = len;
        off += (len + 15) & ~(size_t)15;
      }
    }
    m->frames++;
  }
  std::lock_guard<std::mutex> lk(m->mu);
  m->jobs.erase(ticket);
  return UFD_OK;
}
void* ufd_host_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
void* ufd_model_host_alloc(ufd_model*, size_t bytes) { return std::malloc(bytes ? bytes : 1); }
void ufd_host_free(void* p) { std::free(p); }
}

// ------------------------------------------------------------------ the test
#define CHECK(c)                                                         \
  do {                                                                   \
    if (!(c)) {                                                          \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);         \
      std::exit(1);                                                      \
    }                                                                    \
  } while (0)

struct Sink {
  std::mutex mu;
  std::map<uint64_t, std::vector<uint64_t>> tags;  // stream_id -> tags in delivery order
  uint64_t frames = 0, bad = 0;
};
static void on_result(void* user, const ufd_frame_result* r) {
  Sink* s = static_cast<Sink*>(user);
  std::lock_guard<std::mutex> lk(s->mu);
  s->tags[r->stream_id].push_back(r->tag);
  s->frames++;
  // the frame's content is encoded in its tag: first byte = tag % 251, second = stream id
  const uint8_t b0 = (uint8_t)(r->tag % 251), b1 = (uint8_t)r->stream_id;
  if (b0 == 0xEE) {
    if (r->status != UFD_E_DECODE) s->bad++;
  } else {
    if (r->status != UFD_OK || r->n != (uint32_t)(b0 % 4)) s->bad++;
    for (uint32_t k = 0; k < r->n; k++)
      if (r->dets[k].conf != b1 / 255.0f) s->bad++;
    if (r->jpeg && (r->jpeg_len < 2 || r->jpeg[0] != b0 || r->jpeg[1] != b1)) s->bad++;
  }
}
static std::vector<uint8_t> frame_bytes(uint64_t stream_id, uint64_t tag, size_t len) {
  std::vector<uint8_t> v(len, 0x5A);
  v[0] = (uint8_t)(tag % 251), v[1] = (uint8_t)stream_id;
  return v;
}

int main() {
  ufd_model m320, m640;
  m320.width = 320, m320.max_batch = 8;
  m640.width = 640, m640.max_batch = 4;
  Sink sink;
  ufd_sched_config cfg{};
  cfg.struct_size = sizeof(cfg);
  cfg.model_320 = &m320, cfg.model_640 = &m640;
  cfg.ring_slots = 4, cfg.max_wait_us = 300, cfg.max_inflight = 3, cfg.det_cap = 8, cfg.jpeg_bytes_per_frame = 4096;
  cfg.on_result = on_result, cfg.user = &sink;
  ufd_sched* s = nullptr;
  CHECK(ufd_sched_create(&cfg, &s) == UFD_OK);
  // argument paths
  ufd_stream_config bad_sc{};
  uint32_t h = 0;
  CHECK(ufd_sched_add_stream(s, &bad_sc, &h) == UFD_E_ARG);  // struct_size 0
  CHECK(ufd_sched_push(s, 12345, (const uint8_t*)"ab", 2, 0) == UFD_E_STATE);
  CHECK(ufd_sched_remove_stream(s, 12345) == UFD_E_ARG);

  std::mutex acc_mu;
  std::map<uint64_t, std::vector<uint64_t>> accepted;
  auto producer = [&](uint64_t sid, uint32_t variant, bool annotate, int n) {
    ufd_stream_config sc{};
    sc.struct_size = sizeof(sc), sc.stream_id = sid, sc.variant = variant, sc.annotate = annotate, sc.quality = 90;
    sc.label_width = 1280, sc.label_height = 720;
    uint32_t hs = 0;
    CHECK(ufd_sched_add_stream(s, &sc, &hs) == UFD_OK);
    std::vector<uint64_t> mine;
    for (int t = 0; t < n; t++) {
      const auto f = frame_bytes(sid, (uint64_t)t, 64 + (size_t)(t % 200));
      const int rc = ufd_sched_push(s, hs, f.data(), f.size(), (uint64_t)t);
      CHECK(rc == UFD_OK || rc == UFD_E_FULL);
      if (rc == UFD_OK) mine.push_back((uint64_t)t);
      if (t % 3 == 2) std::this_thread::sleep_for(std::chrono::microseconds(120));
    }
    std::lock_guard<std::mutex> lk(acc_mu);
    accepted[sid] = mine;
  };
  std::vector<std::thread> threads;
  for (int k = 0; k < 4; k++) threads.emplace_back(producer, (uint64_t)(10 + k), k % 2 ? 320u : 640u, k >= 2, 1500);
  // streams that come and go meanwhile, removed with frames still queued or in flight; a racing pusher on each
  std::atomic<int> refused{0};
  for (int rnd = 0; rnd < 300; rnd++) {
    ufd_stream_config sc{};
    sc.struct_size = sizeof(sc), sc.stream_id = (uint64_t)(100 + rnd % 100), sc.variant = rnd % 3 ? 320u : 640u;
    uint32_t hs = 0;
    CHECK(ufd_sched_add_stream(s, &sc, &hs) == UFD_OK);
    std::vector<uint64_t> ok;
    std::thread racer([&, hs] {  // pushes while the main thread removes the stream: delivered or refused, never lost
      for (int t = 0; t < 6; t++) {
        const auto f = frame_bytes(sc.stream_id, (uint64_t)(1000 * rnd + t), 32768);  // (a copy long enough to be caught mid-way)
        const int rc = ufd_sched_push(s, hs, f.data(), f.size(), (uint64_t)(1000 * rnd + t));
        if (rc == UFD_OK) ok.push_back((uint64_t)(1000 * rnd + t));
        else if (rc == UFD_E_STATE) refused++;
        else CHECK(rc == UFD_E_FULL);
      }
    });
    if (rnd % 2) std::this_thread::sleep_for(std::chrono::microseconds(100));
    CHECK(ufd_sched_remove_stream(s, hs) == UFD_OK);
    racer.join();
    CHECK(ufd_sched_remove_stream(s, hs) == UFD_E_ARG);  // already removed (or its entry reused under another generation)
    const auto f = frame_bytes(sc.stream_id, 7, 80);
    CHECK(ufd_sched_push(s, hs, f.data(), f.size(), 7) == UFD_E_STATE);
    std::lock_guard<std::mutex> lk(acc_mu);
    auto& v = accepted[sc.stream_id];
    v.insert(v.end(), ok.begin(), ok.end());
  }
  for (auto& t : threads) t.join();
  CHECK(ufd_sched_flush(s) == UFD_OK);
  ufd_sched_stats st{};
  CHECK(ufd_sched_get_stats(s, &st) == UFD_OK);
  uint32_t live = 0, allocated = 0;
  CHECK(ufd_sched_debug_table(s, &live, &allocated) == UFD_OK);
  CHECK(live == 4 && allocated <= 4 + 300);
  uint64_t total = 0;
  for (auto& kv : accepted) total += kv.second.size();
  {
    std::lock_guard<std::mutex> lk(sink.mu);
    CHECK(sink.bad == 0);
    CHECK(sink.frames == total && st.delivered == total && st.pushed - st.dropped == total);
    for (auto& kv : accepted) {
      if (kv.first < 100) CHECK(sink.tags[kv.first] == kv.second);  // (a long-lived stream: exact order; ids >= 100 are reused by several short-lived streams)
      else CHECK(sink.tags[kv.first].size() == kv.second.size());
    }
  }
  // a corrupt frame is reported, not dropped; destroy delivers what is still queued
  ufd_stream_config sc{};
  sc.struct_size = sizeof(sc), sc.stream_id = 77, sc.variant = 320;
  CHECK(ufd_sched_add_stream(s, &sc, &h) == UFD_OK);
  auto f = frame_bytes(77, 0xEE, 32);
  CHECK(ufd_sched_push(s, h, f.data(), f.size(), 0xEE) == UFD_OK);
  for (int t = 0; t < 3; t++) {
    f = frame_bytes(77, (uint64_t)t, 48);
    CHECK(ufd_sched_push(s, h, f.data(), f.size(), (uint64_t)t) == UFD_OK);
  }
  ufd_sched_destroy(s);
  CHECK(sink.bad == 0 && sink.tags[77].size() == 4);
  std::printf("ok: %llu frames delivered, %llu dropped on full rings, %d pushes refused behind a removal, table %u entries\n",
              (unsigned long long)sink.frames, (unsigned long long)st.dropped, refused.load(), allocated);
  return 0;
}
