// sched.cpp -- SURVEY 8(f) row N4: multi-stream batching scheduler on the C side of the ABI.
// Replaces, for many camera streams per GPU, the reference's leg
//   FrameRouter::run  -> infer_tx.try_send_ref() (drop on a full ring)      infer_server/src/router.rs:64-71
//   INFER_IMAGES_CHANNEL: StaticChannel<StaticImage, 10>                     infer_server/src/lib.rs:32-37
//   Inferer::run: one task, one fixed model, one slot at a time              infer_server/src/inferer.rs:23,29-50
// with per-stream rings, per-stream model variant / output kind, round-robin batch formation under a deadline, and
// several batches in flight per model.  ONE scheduler serves every GPU of the node: each variant may have several
// replicas (the handles ufd_create_replicas returns, one per GPU); a stream is placed on one replica when it is added
// (stream i -> replica i mod n, or the least loaded one) and stays there, batches are formed per (replica, output kind),
// and every replica has its own completion thread -- a slow GPU never holds back another one's results.
// Pure host code over the public entry points of ufd.h (no HIP calls here).
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ufd.h"

namespace {

using Clock = std::chrono::steady_clock;
double ms_between(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

struct Frame {  // one ring slot: the reference's StaticImage with the bytes copied in place (router.rs:66-69)
  std::vector<uint8_t> jpeg;
  uint64_t tag = 0;
  Clock::time_point pushed;
};

struct Stream {
  ufd_stream_config cfg{};
  bool alive = false;
  int klass = -1;
  uint32_t replica = 0;  // index (within its variant) of the replica the stream lives on
  std::deque<Frame*> queued;   // waiting for a batch, oldest first
  std::vector<Frame*> free_;   // ring slots not in use
  std::vector<std::unique_ptr<Frame>> slots;
  uint32_t in_batches = 0;     // frames currently inside dispatched batches
  uint32_t filling = 0;        // ring slots a ufd_sched_push is copying into outside the lock
};

// A stream handle = table index (low kIndexBits) | generation of that table entry: an index is reused once its stream has
// been removed and has drained, and a stale handle of the old stream then fails instead of reaching the new one.
// (16 + 16 bits, indices reused first-in first-out: a stale handle can only alias a newer stream after 65 536 reuses of ONE
// index, which the FIFO spreads over the whole free list)
constexpr uint32_t kIndexBits = 16, kIndexMask = (1u << kIndexBits) - 1;

struct Batch;

// One handle = one GPU's replica of one variant: its batches in flight (dispatch order) and its completion thread.
struct Replica {
  ufd_model* model = nullptr;
  uint32_t variant = 0, index = 0, max_batch = 0;
  uint32_t inflight = 0;      // batches dispatched and not yet delivered
  uint32_t streams = 0;       // live streams placed here
  uint64_t batches = 0, frames = 0, detections = 0;  // delivered
  std::deque<Batch*> q;
  std::condition_variable cv;  // (waits on ufd_sched::mu)
  std::thread completer;
};

// Streams that can share a batch: same replica (model + GPU) and same kind of output.
struct Klass {
  uint32_t variant = 0, annotate = 0, quality = 95, flags = 0;
  float label_w = 0, label_h = 0;
  Replica* rep = nullptr;
  std::vector<uint32_t> streams;
  uint32_t last = 0;  // position (in streams) of the stream served last
};

struct Batch {
  int klass = -1;
  uint32_t count = 0, ticket = 0;
  int why = 0;  // 0 full, 1 deadline, 2 idle
  std::vector<uint32_t> stream;
  std::vector<uint64_t> stream_id;  // copied at dispatch: the completion thread never touches the stream table without the lock
  std::vector<Frame*> frame;
  std::vector<const uint8_t*> ptrs;
  std::vector<size_t> lens, joff, jlen;
  std::vector<ufd_det> dets;
  std::vector<uint32_t> n;
  std::vector<int32_t> status;
  uint8_t* jpeg_out = nullptr;  // pinned (ufd_host_alloc), allocated for the first annotate batch that uses this object
  size_t jpeg_cap = 0;
  Clock::time_point dispatched;
};

}  // namespace

struct ufd_sched {
  ufd_sched_config cfg{};
  uint32_t max_batch[2] = {0, 0};  // largest max_batch over the replicas of each variant (pinned-output sizing)
  std::vector<std::unique_ptr<Replica>> replicas[2];  // [variant index][replica]
  uint64_t placed[2] = {0, 0};     // streams ever added per variant (round-robin placement)
  std::mutex mu;
  std::condition_variable cv_dispatch, cv_flush;
  // stream table: entry i is a live (or draining) stream, or null with i on free_ids.  Always accessed with the lock held.
  std::vector<std::unique_ptr<Stream>> streams;
  std::vector<uint32_t> gen;       // generation of every table entry
  std::deque<uint32_t> free_ids;   // reclaimed entries, reused (oldest first) by ufd_sched_add_stream
  std::vector<Klass> klasses;
  std::vector<std::unique_ptr<Batch>> pool;
  std::vector<Batch*> free_batches;
  bool stop = false, dispatcher_done = false;
  ufd_sched_stats stats{};   // (stats.pushed is derived: accepted + dropped)
  uint64_t accepted = 0;     // frames queued by ufd_sched_push (counted when they ARE queued, under the lock)
  std::thread dispatcher;
};

extern "C" uint32_t ufd_sched_debug_plan(const uint32_t* queued, uint32_t n_streams, uint32_t last, uint32_t max_batch, uint32_t* take) {
  if (!queued || !take || !n_streams) return 0;
  for (uint32_t i = 0; i < n_streams; i++) take[i] = 0;
  uint32_t total = 0;
  bool any = true;
  while (total < max_batch && any) {  // one frame per stream per pass, starting after the stream served last
    any = false;
    for (uint32_t k = 1; k <= n_streams && total < max_batch; k++) {
      const uint32_t i = (last + k) % n_streams;
      if (take[i] < queued[i]) take[i]++, total++, any = true;
    }
  }
  return total;
}

namespace {

int variant_index(uint32_t v) { return v == 320 ? 0 : (v == 640 ? 1 : -1); }

// With the lock held: the next batch of class k if one should leave now, else nullptr; *wake = when to look again.
Batch* form_batch(ufd_sched* s, int k, Clock::time_point now, Clock::time_point* wake) {
  Klass& kl = s->klasses[k];
  Replica& rep = *kl.rep;
  if (kl.streams.empty() || rep.inflight >= s->cfg.max_inflight) return nullptr;
  std::vector<uint32_t> queued(kl.streams.size()), take(kl.streams.size());
  uint32_t waiting = 0;
  Clock::time_point oldest = Clock::time_point::max();
  for (size_t i = 0; i < kl.streams.size(); i++) {
    const Stream& st = *s->streams[kl.streams[i]];
    queued[i] = (uint32_t)st.queued.size();
    waiting += queued[i];
    if (!st.queued.empty()) oldest = std::min(oldest, st.queued.front()->pushed);
  }
  if (!waiting) return nullptr;
  const auto deadline = oldest + std::chrono::microseconds(s->cfg.max_wait_us);
  int why;
  if (waiting >= rep.max_batch) why = 0;
  else if (now >= deadline) why = 1;
  else if (rep.inflight == 0) why = 2;  // the replica is idle: a lone frame does not wait for company
  else {
    *wake = std::min(*wake, deadline);
    return nullptr;
  }
  if (s->free_batches.empty()) {
    s->pool.emplace_back(new Batch);
    s->free_batches.push_back(s->pool.back().get());
  }
  // an annotate class takes a batch object that already owns pinned output, the others one that does not
  size_t pick = s->free_batches.size() - 1;
  for (size_t i = 0; i < s->free_batches.size(); i++)
    if ((s->free_batches[i]->jpeg_cap != 0) == (kl.annotate != 0)) pick = i;
  Batch* b = s->free_batches[pick];
  s->free_batches.erase(s->free_batches.begin() + (long)pick);
  const uint32_t count = ufd_sched_debug_plan(queued.data(), (uint32_t)queued.size(), kl.last, rep.max_batch, take.data());
  b->klass = k, b->count = count, b->why = why;
  b->stream.clear(), b->stream_id.clear(), b->frame.clear(), b->ptrs.clear(), b->lens.clear();
  // frames in round-robin order too, so that a stream's frames keep their order and streams share the head of the batch
  std::vector<uint32_t> left = take;
  const uint32_t start = kl.last;
  bool any = true;
  while (any) {
    any = false;
    for (uint32_t q = 1; q <= kl.streams.size(); q++) {
      const uint32_t i = (start + q) % (uint32_t)kl.streams.size();
      if (!left[i]) continue;
      kl.last = i;  // the stream served last: the next batch starts after it
      Stream& st = *s->streams[kl.streams[i]];
      Frame* f = st.queued.front();
      st.queued.pop_front();
      st.in_batches++;
      left[i]--, any = true;
      b->stream.push_back(kl.streams[i]);
      b->stream_id.push_back(st.cfg.stream_id);
      b->frame.push_back(f);
      b->ptrs.push_back(f->jpeg.data());
      b->lens.push_back(f->jpeg.size());
    }
  }
  b->dets.resize((size_t)count * s->cfg.det_cap);
  b->n.assign(count, 0);
  b->status.assign(count, 0);
  b->joff.assign(count, 0);
  b->jlen.assign(count, 0);
  return b;
}

// With the lock held: a removed stream that has nothing queued, nothing in a dispatched batch and no push in progress
// leaves the tables -- its ring slots (each keeps the capacity of the largest JPEG it held) are freed, its class forgets
// it (the round-robin position is kept on the same neighbour) and its table index goes back on the free list.
void reclaim_if_drained(ufd_sched* s, uint32_t idx) {
  Stream* st = s->streams[idx].get();
  if (!st || st->alive || !st->queued.empty() || st->in_batches || st->filling) return;
  Klass& kl = s->klasses[st->klass];
  for (size_t i = 0; i < kl.streams.size(); i++) {
    if (kl.streams[i] != idx) continue;
    kl.streams.erase(kl.streams.begin() + (long)i);
    const uint32_t left = (uint32_t)kl.streams.size();
    // the next batch still starts behind the stream served last (or, if that was this one, at its successor)
    if (!left) kl.last = 0;
    else if (kl.last >= i) kl.last = (kl.last + left - 1) % left;
    break;
  }
  s->streams[idx].reset();
  s->gen[idx] = (s->gen[idx] + 1) & ((1u << (32 - kIndexBits)) - 1);
  s->free_ids.push_back(idx);
}

// table index of a stream handle, or -1 (unknown, reclaimed, or a stale handle of a reused index)
long stream_index(const ufd_sched* s, uint32_t handle) {
  const uint32_t idx = handle & kIndexMask;
  if (idx >= s->streams.size() || !s->streams[idx] || s->gen[idx] != (handle >> kIndexBits)) return -1;
  return (long)idx;
}

void dispatcher_main(ufd_sched* s) {
  std::unique_lock<std::mutex> lk(s->mu);
  for (;;) {
    bool sent = false;
    auto wake = Clock::time_point::max();
    const auto now = Clock::now();
    for (int k = 0; k < (int)s->klasses.size(); k++) {
      Batch* b = form_batch(s, k, now, &wake);
      if (!b) continue;
      Klass& kl = s->klasses[k];
      Replica* rep = kl.rep;
      b->dispatched = now;
      rep->inflight++;
      s->stats.batches++, s->stats.frames_in_batches += b->count;
      (b->why == 0 ? s->stats.sent_full : (b->why == 1 ? s->stats.sent_deadline : s->stats.sent_idle))++;
      // The batch is detached from the stream queues: submit WITHOUT the scheduler lock (ufd_submit_* takes the handle's
      // lock, which a ufd_wait of the completion thread may hold across its result copies -- pushes must not stall on that),
      // and allocate its pinned output, if it has none yet, out here too (normally done at ufd_sched_add_stream).
      struct {
        ufd_model* model;
        uint32_t annotate, quality, flags;
        float label_w, label_h;
      } kc = {rep->model, kl.annotate, kl.quality, kl.flags, kl.label_w, kl.label_h};
      const size_t need = (size_t)s->cfg.jpeg_bytes_per_frame * std::max(s->max_batch[0], s->max_batch[1]);
      const bool grow = kc.annotate && b->jpeg_cap < need;
      uint8_t* old_out = b->jpeg_out;
      lk.unlock();
      if (grow) {
        // (the batch's buffer fields are read by ufd_sched_add_stream under the lock: allocate into locals out here,
        // publish under the lock)
        if (old_out) ufd_host_free(old_out);
        uint8_t* fresh = static_cast<uint8_t*>(ufd_model_host_alloc(kc.model, need));
        lk.lock();
        b->jpeg_out = fresh, b->jpeg_cap = fresh ? need : 0;
        lk.unlock();
      }
      int rc;
      if (kc.annotate) {
        ufd_annotate a;
        std::memset(&a, 0, sizeof(a));
        a.struct_size = sizeof(a);
        a.label_width = kc.label_w, a.label_height = kc.label_h;
        a.quality = kc.quality, a.flags = kc.flags;
        a.jpeg_out = b->jpeg_out, a.jpeg_cap = b->jpeg_cap;
        a.jpeg_off = b->joff.data(), a.jpeg_len = b->jlen.data();
        rc = ufd_submit_annotate_batch(kc.model, b->ptrs.data(), b->lens.data(), b->count, &a, b->dets.data(), s->cfg.det_cap,
                                       b->n.data(), b->status.data(), &b->ticket);
      } else {
        rc = ufd_submit_jpeg_batch(kc.model, b->ptrs.data(), b->lens.data(), b->count, b->dets.data(), s->cfg.det_cap, b->n.data(),
                                   b->status.data(), &b->ticket);
      }
      if (rc != UFD_OK) {  // nothing ran (all slots busy, bad handle): every frame of the batch reports the failure
        for (auto& st : b->status) st = rc;
        b->ticket = 0;
      }
      lk.lock();
      rep->q.push_back(b);
      rep->cv.notify_all();
      sent = true;
    }
    if (sent) continue;
    if (s->stop) {  // destroy: leave once everything queued has been dispatched
      bool empty = true;
      for (const auto& st : s->streams) empty = empty && (!st || st->queued.empty());
      if (empty) break;
    }
    if (wake == Clock::time_point::max()) s->cv_dispatch.wait(lk);
    else s->cv_dispatch.wait_until(lk, wake);
  }
  s->dispatcher_done = true;
  for (auto& v : s->replicas)
    for (auto& r : v) r->cv.notify_all();
}

// Completion thread of ONE replica: waits for its batches in dispatch order and hands the results out.
void completer_main(ufd_sched* s, Replica* rep) {
  std::unique_lock<std::mutex> lk(s->mu);
  for (;;) {
    rep->cv.wait(lk, [&] { return !rep->q.empty() || s->dispatcher_done; });
    if (rep->q.empty()) break;  // the dispatcher has left and nothing is in flight here
    Batch* b = rep->q.front();
    struct {
      ufd_model* model;
      uint32_t variant, annotate;
    } kl = {rep->model, rep->variant, s->klasses[b->klass].annotate};
    lk.unlock();
    if (b->ticket) {
      const int rc = ufd_wait(kl.model, b->ticket);
      if (rc != UFD_OK && rc != UFD_E_TRUNCATED)
        for (uint32_t i = 0; i < b->count; i++)
          if (b->status[i] == UFD_OK) b->status[i] = rc;
    }
    const auto done = Clock::now();
    if (s->cfg.on_result) {
      for (uint32_t i = 0; i < b->count; i++) {
        ufd_frame_result r;
        std::memset(&r, 0, sizeof(r));
        r.stream_id = b->stream_id[i];
        r.tag = b->frame[i]->tag;
        r.status = b->status[i];
        r.variant = kl.variant;
        r.replica = rep->index;
        const bool ok = r.status == UFD_OK || r.status == UFD_E_TRUNCATED;
        r.n = ok ? b->n[i] : 0;
        r.batch_fill = b->count;
        r.dets = ok ? b->dets.data() + (size_t)i * s->cfg.det_cap : nullptr;
        if (kl.annotate && ok && b->jlen[i]) r.jpeg = b->jpeg_out + b->joff[i], r.jpeg_len = b->jlen[i];
        r.queue_ms = ms_between(b->frame[i]->pushed, b->dispatched);
        r.total_ms = ms_between(b->frame[i]->pushed, done);
        s->cfg.on_result(s->cfg.user, &r);
      }
    }
    lk.lock();
    rep->q.pop_front();
    rep->inflight--;
    rep->batches++, rep->frames += b->count;
    for (uint32_t i = 0; i < b->count; i++)
      if (b->status[i] == UFD_OK || b->status[i] == UFD_E_TRUNCATED) rep->detections += b->n[i];
    for (uint32_t i = 0; i < b->count; i++) {
      Stream& st = *s->streams[b->stream[i]];  // (alive or draining: an entry with frames in a batch is never reclaimed)
      st.free_.push_back(b->frame[i]);
      st.in_batches--;
    }
    for (uint32_t i = 0; i < b->count; i++) reclaim_if_drained(s, b->stream[i]);  // removed streams whose last frames these were
    s->stats.delivered += b->count;
    s->free_batches.push_back(b);
    s->cv_dispatch.notify_all();
    s->cv_flush.notify_all();
  }
}

}  // namespace

extern "C" {

int ufd_sched_create(const ufd_sched_config* cfg, ufd_sched** out) {
  if (!cfg || !out || cfg->struct_size != sizeof(ufd_sched_config)) return UFD_E_ARG;
  // handles per variant: the replica arrays (one per GPU, ufd_create_replicas), or the single-handle shorthand
  std::vector<ufd_model*> models[2];
  const struct {
    ufd_model* one;
    ufd_model* const* many;
    uint32_t n;
  } src[2] = {{cfg->model_320, cfg->models_320, cfg->n_320}, {cfg->model_640, cfg->models_640, cfg->n_640}};
  for (int i = 0; i < 2; i++) {
    if (src[i].n > UFD_MAX_REPLICAS || (src[i].n && !src[i].many) || (src[i].n && src[i].one)) return UFD_E_ARG;
    for (uint32_t r = 0; r < src[i].n; r++) {
      if (!src[i].many[r]) return UFD_E_ARG;
      for (ufd_model* seen : models[i])
        if (seen == src[i].many[r]) return UFD_E_ARG;  // a handle listed twice
      models[i].push_back(src[i].many[r]);
    }
    if (src[i].one) models[i].push_back(src[i].one);
  }
  if (models[0].empty() && models[1].empty()) return UFD_E_ARG;
  if (cfg->placement != UFD_SCHED_PLACE_ROUND_ROBIN && cfg->placement != UFD_SCHED_PLACE_LEAST_LOADED) return UFD_E_ARG;
  std::unique_ptr<ufd_sched> s(new ufd_sched);
  s->cfg = *cfg;
  if (!s->cfg.ring_slots) s->cfg.ring_slots = 10;
  if (!s->cfg.max_wait_us) s->cfg.max_wait_us = 2000;
  if (s->cfg.max_wait_us == UFD_SCHED_NO_WAIT) s->cfg.max_wait_us = 0;  // a batch leaves as soon as the model has a free slot
  if (!s->cfg.max_inflight) s->cfg.max_inflight = 6;
  s->cfg.max_inflight = std::min<uint32_t>(s->cfg.max_inflight, UFD_MAX_SLOTS);
  if (!s->cfg.det_cap) s->cfg.det_cap = 256;
  if (!s->cfg.jpeg_bytes_per_frame) s->cfg.jpeg_bytes_per_frame = 512 * 1024;
  for (int i = 0; i < 2; i++) {
    for (size_t r = 0; r < models[i].size(); r++) {
      uint32_t w = 0, mb = 0;
      if (ufd_model_info(models[i][r], &w, nullptr, nullptr) != UFD_OK || w != (i ? 640u : 320u)) return UFD_E_ARG;
      if (ufd_model_limits(models[i][r], &mb, nullptr, nullptr) != UFD_OK || !mb) return UFD_E_ARG;
      std::unique_ptr<Replica> rep(new Replica);
      rep->model = models[i][r], rep->variant = i ? 640u : 320u, rep->index = (uint32_t)r, rep->max_batch = mb;
      s->max_batch[i] = std::max(s->max_batch[i], mb);
      s->replicas[i].push_back(std::move(rep));
    }
  }
  ufd_sched* p = s.release();
  p->dispatcher = std::thread(dispatcher_main, p);
  for (auto& v : p->replicas)
    for (auto& r : v) r->completer = std::thread(completer_main, p, r.get());
  *out = p;
  return UFD_OK;
}

void ufd_sched_destroy(ufd_sched* s) {
  if (!s) return;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    s->stop = true;
  }
  s->cv_dispatch.notify_all();
  if (s->dispatcher.joinable()) s->dispatcher.join();
  for (auto& v : s->replicas)
    for (auto& r : v) {
      r->cv.notify_all();
      if (r->completer.joinable()) r->completer.join();
    }
  for (auto& b : s->pool)
    if (b->jpeg_out) ufd_host_free(b->jpeg_out);
  delete s;
}

int ufd_sched_add_stream(ufd_sched* s, const ufd_stream_config* cfg, uint32_t* stream) {
  if (!s || !cfg || !stream || cfg->struct_size != sizeof(ufd_stream_config)) return UFD_E_ARG;
  const int vi = variant_index(cfg->variant);
  if (vi < 0 || s->replicas[vi].empty()) return UFD_E_ARG;
  if (cfg->replica > s->replicas[vi].size()) return UFD_E_ARG;  // (r + 1 of a replica that does not exist)
  const uint32_t quality = cfg->quality ? cfg->quality : 95;
  if (cfg->annotate && (quality < 1 || quality > 100)) return UFD_E_ARG;
  // Pinned output of the batches the annotate classes can have in flight (max_inflight per replica): allocated now and
  // without the scheduler lock -- not by the dispatcher in the middle of the stream (a first-use hipHostMalloc of tens of
  // MB used to stall every push behind it).  Portable memory (ufd_model_host_alloc): a batch object serves any replica.
  std::vector<std::unique_ptr<Batch>> fresh;
  if (cfg->annotate) {
    const size_t need = (size_t)s->cfg.jpeg_bytes_per_frame * std::max(s->max_batch[0], s->max_batch[1]);
    const size_t want = (size_t)s->cfg.max_inflight * s->replicas[vi].size();
    size_t have = 0;
    {
      std::lock_guard<std::mutex> lk(s->mu);
      for (auto& b : s->pool)
        if (b->jpeg_cap >= need) have++;
    }
    ufd_model* model = s->replicas[vi][0]->model;
    for (; have < want; have++) {
      std::unique_ptr<Batch> b(new Batch);
      b->jpeg_out = static_cast<uint8_t*>(ufd_model_host_alloc(model, need));
      if (!b->jpeg_out) break;  // (the dispatcher tries again for the batch that needs it)
      b->jpeg_cap = need;
      fresh.push_back(std::move(b));
    }
  }
  std::lock_guard<std::mutex> lk(s->mu);
  for (auto& b : fresh) {
    s->free_batches.push_back(b.get());
    s->pool.push_back(std::move(b));
  }
  // placement (stream i -> GPU i mod G): the caller's choice, else round-robin in order of arrival, else the
  // replica with the fewest live streams (ties: the lowest index)
  auto& reps = s->replicas[vi];
  uint32_t r;
  if (cfg->replica) {
    r = cfg->replica - 1;
  } else if (s->cfg.placement == UFD_SCHED_PLACE_LEAST_LOADED) {
    r = 0;
    for (uint32_t i = 1; i < reps.size(); i++)
      if (reps[i]->streams < reps[r]->streams) r = i;
  } else {
    r = (uint32_t)(s->placed[vi] % reps.size());
  }
  s->placed[vi]++;
  Replica* rep = reps[r].get();
  int k = -1;
  for (size_t i = 0; i < s->klasses.size(); i++) {
    const Klass& c = s->klasses[i];
    if (c.rep == rep && c.annotate == (cfg->annotate ? 1u : 0u) &&
        (!cfg->annotate || (c.quality == quality && c.flags == cfg->flags && c.label_w == cfg->label_width && c.label_h == cfg->label_height)))
      k = (int)i;
  }
  if (k < 0) {
    Klass c;
    c.variant = cfg->variant, c.annotate = cfg->annotate ? 1 : 0, c.quality = quality, c.flags = cfg->flags;
    c.label_w = cfg->label_width, c.label_h = cfg->label_height;
    c.rep = rep;
    s->klasses.push_back(c);
    k = (int)s->klasses.size() - 1;
  }
  uint32_t idx;
  if (!s->free_ids.empty()) {
    idx = s->free_ids.front();
    s->free_ids.pop_front();
  } else {
    if (s->streams.size() > kIndexMask) return UFD_E_TOO_LARGE;
    idx = (uint32_t)s->streams.size();
    s->streams.emplace_back();
    s->gen.push_back(0);
  }
  s->streams[idx].reset(new Stream);
  Stream& st = *s->streams[idx];
  st.cfg = *cfg, st.alive = true, st.klass = k, st.replica = r;
  for (uint32_t i = 0; i < s->cfg.ring_slots; i++) {
    st.slots.emplace_back(new Frame);
    st.free_.push_back(st.slots.back().get());
  }
  rep->streams++;
  *stream = idx | (s->gen[idx] << kIndexBits);
  s->klasses[k].streams.push_back(idx);
  return UFD_OK;
}

int ufd_sched_remove_stream(ufd_sched* s, uint32_t stream) {
  if (!s) return UFD_E_ARG;
  std::lock_guard<std::mutex> lk(s->mu);
  const long idx = stream_index(s, stream);
  if (idx < 0 || !s->streams[idx]->alive) return UFD_E_ARG;
  s->streams[idx]->alive = false;  // frames already queued are still delivered; nothing new is accepted
  s->klasses[s->streams[idx]->klass].rep->streams--;  // (placement counts live streams)
  reclaim_if_drained(s, (uint32_t)idx);  // (otherwise the completion thread reclaims it behind its last frame)
  return UFD_OK;
}

int ufd_sched_push(ufd_sched* s, uint32_t stream, const uint8_t* jpeg, size_t len, uint64_t tag) {
  if (!s || !jpeg || !len) return UFD_E_ARG;
  Frame* f = nullptr;
  long idx;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    idx = stream_index(s, stream);
    if (s->stop || idx < 0 || !s->streams[idx]->alive) return UFD_E_STATE;
    Stream& st = *s->streams[idx];
    if (st.free_.empty()) {  // router.rs:65: `if let Ok(mut frame) = self.infer_tx.try_send_ref()` -- else the frame is dropped
      s->stats.dropped++;
      return UFD_E_FULL;
    }
    f = st.free_.back();
    st.free_.pop_back();
    st.filling++;  // the slot is ours: the stream cannot be reclaimed while the copy below runs without the lock
  }
  f->jpeg.assign(jpeg, jpeg + len);  // frame.2.clear(); frame.2.extend_from_slice(..) (router.rs:68-69)
  f->tag = tag;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    Stream& st = *s->streams[idx];
    st.filling--;
    if (!st.alive || s->stop) {  // removed (or the scheduler stopped) during the copy: as if the push had come too late
      st.free_.push_back(f);
      reclaim_if_drained(s, (uint32_t)idx);
      return UFD_E_STATE;  // (never counted: a ufd_sched_flush that ran meanwhile does not wait for this frame)
    }
    f->pushed = Clock::now();
    st.queued.push_back(f);
    s->accepted++;  // counted only now that the frame IS queued: ufd_sched_flush's target never includes a frame that is withdrawn
  }
  s->cv_dispatch.notify_all();
  return UFD_OK;
}

int ufd_sched_push_batch(ufd_sched* s, uint32_t stream, const uint8_t* const* jpegs, const size_t* lens, const uint64_t* tags,
                         uint32_t count, uint32_t* accepted) {
  if (accepted) *accepted = 0;
  if (!s || !jpegs || !lens || !count) return UFD_E_ARG;
  for (uint32_t i = 0; i < count; i++)
    if (!jpegs[i] || !lens[i]) return UFD_E_ARG;
  std::vector<Frame*> got;
  long idx;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    idx = stream_index(s, stream);
    if (s->stop || idx < 0 || !s->streams[idx]->alive) return UFD_E_STATE;
    Stream& st = *s->streams[idx];
    const uint32_t k = std::min<uint32_t>(count, (uint32_t)st.free_.size());
    s->stats.dropped += count - k;  // router.rs:65, frame by frame: the ones that find no slot are dropped
    if (!k) return UFD_E_FULL;
    got.assign(st.free_.end() - k, st.free_.end());
    st.free_.resize(st.free_.size() - k);
    st.filling += k;
  }
  for (size_t i = 0; i < got.size(); i++) {
    got[i]->jpeg.assign(jpegs[i], jpegs[i] + lens[i]);
    got[i]->tag = tags ? tags[i] : 0;
  }
  {
    std::lock_guard<std::mutex> lk(s->mu);
    Stream& st = *s->streams[idx];
    st.filling -= (uint32_t)got.size();
    if (!st.alive || s->stop) {
      for (Frame* f : got) st.free_.push_back(f);
      reclaim_if_drained(s, (uint32_t)idx);
      return UFD_E_STATE;
    }
    const auto now = Clock::now();
    for (Frame* f : got) {
      f->pushed = now;
      st.queued.push_back(f);
    }
    s->accepted += got.size();
  }
  s->cv_dispatch.notify_all();
  if (accepted) *accepted = (uint32_t)got.size();
  return got.size() == count ? UFD_OK : UFD_E_FULL;
}

int ufd_sched_flush(ufd_sched* s) {
  if (!s) return UFD_E_ARG;
  std::unique_lock<std::mutex> lk(s->mu);
  const uint64_t target = s->accepted;  // every frame a returned ufd_sched_push has queued
  s->cv_flush.wait(lk, [&] { return s->stats.delivered >= target; });
  return UFD_OK;
}

int ufd_sched_debug_table(ufd_sched* s, uint32_t* live, uint32_t* allocated) {
  if (!s) return UFD_E_ARG;
  std::lock_guard<std::mutex> lk(s->mu);
  uint32_t n = 0;
  for (const auto& st : s->streams) n += st ? 1u : 0u;
  if (live) *live = n;
  if (allocated) *allocated = (uint32_t)s->streams.size();
  return UFD_OK;
}

int ufd_sched_get_stats(ufd_sched* s, ufd_sched_stats* out) {
  if (!s || !out) return UFD_E_ARG;
  std::lock_guard<std::mutex> lk(s->mu);
  *out = s->stats;
  out->pushed = s->accepted + s->stats.dropped;
  return UFD_OK;
}

int ufd_sched_stream_replica(ufd_sched* s, uint32_t stream, uint32_t* replica) {
  if (!s || !replica) return UFD_E_ARG;
  std::lock_guard<std::mutex> lk(s->mu);
  const long idx = stream_index(s, stream);
  if (idx < 0) return UFD_E_ARG;
  *replica = s->streams[idx]->replica;
  return UFD_OK;
}

int ufd_sched_get_replica_stats(ufd_sched* s, uint32_t variant, ufd_sched_replica_stats* out, uint32_t cap, uint32_t* n) {
  const int vi = variant_index(variant);
  if (!s || vi < 0 || !n || (cap && !out)) return UFD_E_ARG;
  std::lock_guard<std::mutex> lk(s->mu);
  *n = (uint32_t)s->replicas[vi].size();
  for (uint32_t i = 0; i < *n && i < cap; i++) {
    const Replica& r = *s->replicas[vi][i];
    out[i].replica = r.index, out[i].streams = r.streams, out[i].inflight = r.inflight;
    out[i].batches = r.batches, out[i].frames = r.frames, out[i].detections = r.detections;
  }
  return UFD_OK;
}

}  // extern "C"
