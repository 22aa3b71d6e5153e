// model.cpp -- host side of libufacehip.so: the C ABI of include/ufd.h and the per-batch pipeline that replaces the
// reference's single Inferer task (infer_server/src/inferer.rs:29-50).  A handle owns four device contexts (a stream and a
// working set each = the runtime's four hardware queues; NO copy stream, see below) and an issue worker per context:
//   worker: header + marker scan of the batch's JPEGs, their bytes into one pinned block, fetched by a kernel on the
//           context's own stream (k_stage_in), then every launch -- device entropy decoding -> IDCT -> upsample / colour / normalise (fused into the stem) [-> Triangle resize] ->
//           the network (plan.cpp) -> softmax / prior decode / threshold -> sort + greedy NMS [-> rectangles + re-encode]
//   ufd_wait: a few hundred bytes of detections per frame come back (and the annotated streams).
// Weights (1.1 MB) and priors stay resident in HBM for the life of the handle.  State and shared helpers: model_types.hpp.
#include "model_types.hpp"
#include "model_internal.hpp"
#include "model_parts.hpp"

namespace ufd {
thread_local std::string g_create_error;
thread_local Ctx* tl_cur = nullptr;
thread_local ThreadPool* tl_pool = nullptr;
thread_local bool tl_prof = true;  // record kernel events for the batch being issued by this thread
thread_local bool tl_force_rider = false;  // enqueue_layer_launch: issue a riding layer on its own (its host could not take it)
thread_local Worker* tl_worker = nullptr;  // issue worker this thread is (host statistics go to it), or null on API threads
thread_local uint64_t tl_launches = 0;     // launches + copies enqueued by this thread (ProfScope counts them)
}  // namespace ufd

using namespace ufd;

namespace {

// ---------------------------------------------------------------- events behind copies
// ROCm 7.2's runtime keeps ~2 KB of host memory for every event recorded DIRECTLY behind an asynchronous copy-engine
// transfer on a stream and never gives it back: the host-bytes path grew 2.1 KB per batch, 12 GB per hour at 54 k frames/s
// (tools/soak.py; reproduced on the runtime alone by tools/ubench/leak_probe2.hip: copy + hipEventRecord grows whether the
// event is waited for by a stream or by the host, copy + hipStreamSynchronize does not, and neither does copy + any kernel
// + hipEventRecord).  So an empty kernel goes between a copy and the event that marks it.
__global__ void k_copy_fence() {}
}  // namespace
void ufd::launch_copy_fence(hipStream_t stream) { ufd_launch(k_copy_fence, dim3(1), dim3(64), 0, stream); }
hipError_t ufd::record_behind_copy(hipEvent_t ev, hipStream_t stream) {
  launch_copy_fence(stream);
  return hipEventRecord(ev, stream);
}
namespace {

// (k_stage_in, the staging block's way in: entropy_stage.cpp)
__global__ __launch_bounds__(256) void k_results_out(const uint32_t* __restrict__ d_status, const uint32_t* __restrict__ d_ndet,
                                                     const float* __restrict__ d_dets, uint32_t det_stride_floats, uint32_t* h_status,
                                                     uint32_t* h_ndet, float* h_dets, uint32_t max_rows) {
  const uint32_t f = blockIdx.x;
  const uint32_t n = d_ndet[f];
  if (threadIdx.x == 0) {
    h_ndet[f] = n;
    if (d_status) h_status[f] = d_status[f];
  }
  const uint32_t words = min(n, max_rows) * (uint32_t)(sizeof(Det) / 4);
  const float* src = d_dets + (size_t)f * det_stride_floats;
  float* dst = h_dets + (size_t)f * max_rows * (sizeof(Det) / 4);
  for (uint32_t i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
}
}  // namespace

namespace ufd {  // (library-internal names: what debug_taps.cpp and entropy_stage.cpp share is declared in model_parts.hpp)

int alloc_slot(ufd_model* m, Slot& s) {
  if (s.h_descs) return UFD_OK;
  HIPC(m, hipHostMalloc(&s.h_stage, m->stage_cap, hipHostMallocDefault));
  {
    void* dp = nullptr;  // the same pages as the device sees them (k_stage_in)
    s.h_stage_dev = hipHostGetDevicePointer(&dp, s.h_stage, 0) == hipSuccess ? static_cast<uint8_t*>(dp) : nullptr;
  }
  s.h_descs = reinterpret_cast<JpegFrameDesc*>(s.h_stage);
  s.h_scans = reinterpret_cast<HuffScan*>(s.h_stage + m->scans_off);
  s.h_ivs = reinterpret_cast<HuffInterval*>(s.h_stage + m->ivs_off);
  HIPC(m, hipHostMalloc(&s.h_dets, sizeof(Det) * kDetCopy * m->B, hipHostMallocDefault));
  HIPC(m, hipMalloc(&s.d_dets, sizeof(Det) * m->K * m->B));
  HIPC(m, hipMemset(s.d_dets, 0, sizeof(Det) * m->K * m->B));
  // (decode status and detection counts side by side, as on the device: ONE copy brings both back)
  HIPC(m, hipHostMalloc(&s.h_gpu_status, sizeof(uint32_t) * 2 * m->B, hipHostMallocDefault));
  s.h_ndet = s.h_gpu_status + m->B;
  {
    void *ds = nullptr, *dd = nullptr;  // the result arrays as the device sees them (k_results_out)
    s.h_status_dev = hipHostGetDevicePointer(&ds, s.h_gpu_status, 0) == hipSuccess ? static_cast<uint32_t*>(ds) : nullptr;
    s.h_dets_dev = hipHostGetDevicePointer(&dd, s.h_dets, 0) == hipSuccess ? static_cast<Det*>(dd) : nullptr;
  }
  s.plans.resize(m->B);
  HIPC(m, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  s.st.resize(m->B);
  return UFD_OK;
}

// ---------------------------------------------------------------- resize taps (image 0.24.5 sample.rs)
float tri_kernel(float x) {
  float a = std::fabs(x);
  return a < 1.0f ? 1.0f - a : 0.0f;
}

int build_axis_taps(ufd_model* m, int S, int D, TapsDev* out) {
  const float ratio = (float)S / (float)D;
  const float sratio = ratio < 1.0f ? 1.0f : ratio;
  const float support = 1.0f * sratio;
  std::vector<int32_t> left(D), cnt(D);
  std::vector<std::vector<float>> ws(D);
  int maxn = 1;
  for (int o = 0; o < D; o++) {
    float in = ((float)o + 0.5f) * ratio;
    long l = (long)std::floor(in - support);
    l = std::max(0L, std::min(l, (long)S - 1));
    long r = (long)std::ceil(in + support);
    r = std::max(l + 1, std::min(r, (long)S));
    in = in - 0.5f;
    float sum = 0.0f;
    for (long i = l; i < r; i++) {
      float w = tri_kernel(((float)i - in) / sratio);
      ws[o].push_back(w);
      sum += w;
    }
    for (float& w : ws[o]) w /= sum;
    left[o] = (int32_t)l;
    cnt[o] = (int32_t)ws[o].size();
    maxn = std::max(maxn, (int)ws[o].size());
  }
  std::vector<float> flat((size_t)D * maxn, 0.0f);
  for (int o = 0; o < D; o++) std::copy(ws[o].begin(), ws[o].end(), flat.begin() + (size_t)o * maxn);
  HIPC(m, hipMalloc(&out->left, sizeof(int32_t) * D));
  HIPC(m, hipMalloc(&out->cnt, sizeof(int32_t) * D));
  HIPC(m, hipMalloc(&out->w, sizeof(float) * flat.size()));
  HIPC(m, hipMemcpy(out->left, left.data(), sizeof(int32_t) * D, hipMemcpyHostToDevice));
  HIPC(m, hipMemcpy(out->cnt, cnt.data(), sizeof(int32_t) * D, hipMemcpyHostToDevice));
  HIPC(m, hipMemcpy(out->w, flat.data(), sizeof(float) * flat.size(), hipMemcpyHostToDevice));
  out->stride = maxn;
  return UFD_OK;
}

int get_taps(ufd_model* m, int sw, int sh, ResizeTaps* vert, ResizeTaps* horz) {
  std::lock_guard<std::mutex> lk(m->shared_mu);
  auto key = std::make_pair(sw, sh);
  auto it = m->taps.find(key);
  m->taps_used[key] = ++m->plan_seq;
  if (it == m->taps.end()) {
    if (m->taps.size() >= ufd_model::kMaxTapSets) {
      // a stream that sweeps frame sizes must not grow the cache without bound: drop the table set unused the longest.
      // Kernels already enqueued may still read it, so every context drains first (rare: a new size beyond the cap).
      for (int c = 0; c < m->num_ctx; c++)
        if (m->ctx[c].stream) (void)hipStreamSynchronize(m->ctx[c].stream);
      auto victim = m->taps.begin();
      for (auto jt = m->taps.begin(); jt != m->taps.end(); ++jt)
        if (m->taps_used[jt->first] < m->taps_used[victim->first]) victim = jt;
      for (TapsDev* t : {&victim->second.first, &victim->second.second}) {
        (void)hipFree(t->left), (void)hipFree(t->cnt), (void)hipFree(t->w);
      }
      m->taps_used.erase(victim->first);
      m->taps.erase(victim);
    }
    std::pair<TapsDev, TapsDev> e;
    int rc = build_axis_taps(m, sh, m->H, &e.first);
    if (rc) return rc;
    rc = build_axis_taps(m, sw, m->W, &e.second);
    if (rc) return rc;
    it = m->taps.emplace(key, e).first;
  }
  *vert = ResizeTaps{it->second.first.left, it->second.first.cnt, it->second.first.w, it->second.first.stride};
  *horz = ResizeTaps{it->second.second.left, it->second.second.cnt, it->second.second.w, it->second.second.stride};
  return UFD_OK;
}

// ---------------------------------------------------------------- GPU stages
// [count][3][H][W] in d_input (or the sample planes, fused stem) -> every conv output the heads need.
void enqueue_forward(ufd_model* m, uint32_t count, Slot* s) {
  if (s) gate_wait_for_previous(m, *s, tl_cur->stream);
  for (int i = 0; i < kNumConv; i++) {
    enqueue_layer(m, i, count);
    if (s) gate_pass(m, *s, i, tl_cur->stream);
  }
  tl_cur->last_forward_count = count;
}

void enqueue_heads(ufd_model* m, uint32_t count, bool raw_outputs) {
  HeadArgs h;
  int base = 0;
  for (int i = 0; i < 4; i++) {
    const Layer& c = m->layers[kHeadCls[i]];
    h.cls[i] = tensor_ptr(m, c.out_tensor);
    h.reg[i] = tensor_ptr(m, m->layers[kHeadReg[i]].out_tensor);
    h.plane[i] = c.oh * c.ow;
    h.anchors[i] = kHeadAnchors[i];
    h.base[i] = base;
    base += h.plane[i] * kHeadAnchors[i];
  }
  h.base[4] = base;
  // (d_counts is zero here: allocated zeroed, and k_sort_nms puts every counter back to zero once it has read it)
  ProfScope ps(m, "head_decode", (double)count * m->K * (6 + 4) * 4, 0);
  launch_head_decode(h, m->d_priors, count, m->cfg.min_confidence, raw_outputs ? tl_cur->d_scores : nullptr, tl_cur->d_boxes, tl_cur->d_keys, m->key_stride,
                     tl_cur->d_counts, tl_cur->stream);
}

void enqueue_nms(ufd_model* m, Slot& s, uint32_t count) {
  ProfScope ps(m, "sort_nms", 0, 0);
  // A frame or a few at a time (the batch goes in and out by kernels): ONE launch for A7-A10 and the way out -- every frame
  // is finished inside k_sort_nms (a frame with more than 256 candidates by its in-kernel block loop instead of the
  // matrix path's two extra launches, which a lone frame pays 4.6 us each for whether it needs them or not: 3 % of the
  // bench's frames do), and the kernel writes statuses, counts and detections to the slot's pinned arrays itself.
  if (s.small_batch && s.h_status_dev && s.h_dets_dev) {
    NmsHostOut ho;
    ho.d_status = s.gpu_entropy ? tl_cur->d_status : nullptr;
    ho.h_status = s.h_status_dev, ho.h_ndet = s.h_status_dev + m->B;
    ho.h_dets = reinterpret_cast<float*>(s.h_dets_dev), ho.max_rows = kDetCopy;
    launch_sort_nms(tl_cur->d_keys, m->key_stride, tl_cur->d_counts, tl_cur->d_boxes, m->K, m->cfg.max_iou, s.d_dets, m->K, tl_cur->d_ndet,
                    tl_cur->d_spill, nullptr, count, tl_cur->stream, ho);
    s.results_in_nms = true;
    return;
  }
  launch_sort_nms(tl_cur->d_keys, m->key_stride, tl_cur->d_counts, tl_cur->d_boxes, m->K, m->cfg.max_iou, s.d_dets, m->K, tl_cur->d_ndet,
                  tl_cur->d_spill, tl_cur->d_nms_mat, count, tl_cur->stream);
}

int enqueue_results_copy(ufd_model* m, Slot& s, uint32_t count) {
  // [B decode statuses][B detection counts] are one allocation on both sides: one copy of B + count words (statuses past
  // `count` are stale and never read), or the counts alone when the host decoded the entropy stage
  if (s.results_in_nms) {
    s.results_in_nms = false;  // (k_sort_nms wrote them: enqueue_nms)
  } else if (s.small_batch && s.h_status_dev && s.h_dets_dev) {
    static_assert(sizeof(Det) % 4 == 0, "Det is copied as words");
    ufd_launch(k_results_out, dim3(count), dim3(256), 0, tl_cur->stream, s.gpu_entropy ? tl_cur->d_status : nullptr, tl_cur->d_ndet,
                       reinterpret_cast<const float*>(s.d_dets), (uint32_t)(sizeof(Det) / 4 * m->K), s.h_status_dev, s.h_status_dev + m->B,
                       reinterpret_cast<float*>(s.h_dets_dev), kDetCopy);
    tl_launches++;
  } else {
    if (s.gpu_entropy)
      HIPC(m, hipMemcpyAsync(s.h_gpu_status, tl_cur->d_status, sizeof(uint32_t) * ((size_t)m->B + count), hipMemcpyDeviceToHost, tl_cur->stream));
    else
      HIPC(m, hipMemcpyAsync(s.h_ndet, tl_cur->d_ndet, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, tl_cur->stream));
    HIPC(m, hipMemcpy2DAsync(s.h_dets, sizeof(Det) * kDetCopy, s.d_dets, sizeof(Det) * m->K, sizeof(Det) * kDetCopy, count,
                             hipMemcpyDeviceToHost, tl_cur->stream));
  }
  span_end(s);
  HIPC(m, hipEventRecord(s.done, tl_cur->stream));
  s.ctx = tl_cur;
  return UFD_OK;
}

// Releases the slot on every exit path of finish_slot (an early HIP error return must not leak it for the life of the
// handle).  `locked`: the caller holds m->mu (synchronous entry points); otherwise the release takes it -- ufd_wait runs
// the copies below WITHOUT the handle lock, so that other threads can submit while a batch's annotated streams cross PCIe.
struct SlotRelease {
  ufd_model* m;
  Slot& s;
  bool locked;
  ~SlotRelease() {
    if (locked) {
      s.busy = s.waiting = false;
    } else {
      std::lock_guard<std::mutex> lk(m->mu);
      s.busy = s.waiting = false;
    }
  }
};

// waits for the slot's batch and hands results to the caller's arrays
int finish_slot(ufd_model* m, Slot& s, bool locked) {
  SlotRelease release{m, s, locked};
  if (s.issue_rc != UFD_OK) {  // the worker could not issue the batch
    const int rc = s.issue_rc;
    m->fail(rc, s.issue_err);
    return rc;
  }
  {
    const uint64_t t0 = now_ns();
    if (s.relaxed_wait) {
      // A pipelined caller (other batches of the handle are queued behind this one): the GPU has work whatever happens
      // here, so waking up some tens of microseconds after the event costs no throughput -- and a waiter spinning inside
      // hipEventSynchronize would keep one CPU of a host that eight ranks share busy for nothing.
      for (;;) {
        const hipError_t q = hipEventQuery(s.done);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return m->fail(UFD_E_DEVICE, std::string("hipEventQuery: ") + hipGetErrorString(q));
        timespec ts{0, 20000};
        nanosleep(&ts, nullptr);
      }
    } else {
      HIPC(m, hipEventSynchronize(s.done));  // one batch at a time: the latency form, the runtime spins on the signal
    }
    m->ns_wait.fetch_add(now_ns() - t0, std::memory_order_relaxed);
  }
  span_fold(m, s);
  // timing events are resolved lazily (ufd_profile_read): querying ~100 events per batch here
  // would stall the submit/wait pipeline
  bool flush = false;
  {
    std::lock_guard<std::mutex> lk(m->shared_mu);
    flush = m->prof_pending.size() > 16384;
  }
  if (flush) prof_flush(m);
  int rc = UFD_OK;
  for (uint32_t i = 0; i < s.count; i++) {
    int32_t st = s.st[i];
    uint32_t nd = 0;
    if (st == UFD_OK && s.gpu_entropy && s.h_gpu_status[i]) st = UFD_E_DECODE;  // device decoder hit a corrupt stream
    if (st == UFD_OK) {
      nd = s.h_ndet[i];
      const uint32_t ncopy = std::min(nd, s.cap);
      ufd_det* dst = s.out + (size_t)i * s.cap;
      const uint32_t fast = std::min(ncopy, kDetCopy);
      std::memcpy(dst, s.h_dets + (size_t)i * kDetCopy, sizeof(Det) * fast);
      if (ncopy > fast) {  // rare: more than kDetCopy detections requested for one frame.  The slot's own
        // device buffer: no later batch can have written it (the slot is busy until this returns)
        HIPC(m, hipMemcpy(dst + fast, s.d_dets + (size_t)i * m->K + fast, sizeof(Det) * (ncopy - fast),
                          hipMemcpyDeviceToHost));
      }
      if (nd > s.cap) st = UFD_E_TRUNCATED;
    }
    s.st[i] = st;  // the frame's final status (fetch_streams reads it)
    if (s.n) s.n[i] = nd;
    if (s.status) s.status[i] = st;
    if (st != UFD_OK && rc == UFD_OK && !s.status) rc = st;
  }
  if (s.annot) {
    const int arc = fetch_streams(m, s);
    if (arc != UFD_OK && rc == UFD_OK) rc = arc;
  }
  return rc;
}

Slot* find_free_slot(ufd_model* m) {
  for (auto& s : m->slots)
    if (!s.busy) {
      // a fresh job: nothing of the slot's previous batch (a failed issue included) may leak into it
      s.issue_rc = UFD_OK;
      s.issue_err.clear();
      s.job_jpegs = nullptr, s.job_lens = nullptr, s.job_staged = nullptr;
      s.annot = false, s.annot_ran = false;
      s.state = 0;
      s.seq = 0, s.gate_published = false;
      s.results_in_nms = false;
      s.span_idx = -1;
      s.relaxed_wait = false;
      return &s;
    }
  return nullptr;
}

int check_outputs(ufd_model* m, const void* out, uint32_t cap, const void* n) {
  if ((!out && cap) || !n) return m->fail(UFD_E_ARG, "null output pointer");
  return UFD_OK;
}


int submit_jpegs(ufd_model* m, Slot& s, const uint8_t* const* jpegs, const size_t* lens, uint32_t count) {
  int rc = alloc_slot(m, s);
  if (rc) return rc;
  s.count = count;
  int buf = 0;
  bool any_ok = false;
  rc = entropy_stage(m, s, jpegs, lens, count, &buf, &any_ok);
  if (rc) return rc;
  return run_decoded(m, s, count, any_ok, tl_cur->d_descs_buf[buf], tl_cur->d_coef_buf[buf], buf);
}

int submit_staged(ufd_model* m, Slot& s, const ufd_staged& g) {
  int rc = alloc_slot(m, s);
  if (rc) return rc;
  Ctx& c = *tl_cur;
  const uint32_t count = g.count;
  s.count = count;
  std::memcpy(s.h_descs, g.h_descs.data(), sizeof(JpegFrameDesc) * count);
  for (uint32_t i = 0; i < count; i++) s.st[i] = g.st[i];
  s.gpu_entropy = true;
  s.small_batch = false;
  s.coef_zigzag = true;
  int buf = 0;
  if (g.plan.any_ok) {
    buf = c.flip;
    c.flip ^= 1;
    // (the slab is written on the context's own stream: ordered behind its previous readers)
    span_begin(s);
    rc = enqueue_device_entropy(m, c, g.plan, count, g.d_blob, g.d_descs, g.d_scans, g.d_ivs, c.d_coef_buf[buf]);
    if (rc) return rc;
  }
  return run_decoded(m, s, count, g.plan.any_ok, g.d_descs, c.d_coef_buf[buf], buf);
}

// Every decodable frame of the batch is 3-component YCbCr 4:2:0 (what the fancy-upsampling fast paths take).
// ... or 4:2:2 (h2v1, chroma at full height): what the fused stem takes besides 4:2:0, frame by frame
bool batch_is_h2(const Slot& s, uint32_t count) {
  for (uint32_t i = 0; i < count; i++) {
    if (s.st[i] != UFD_OK) continue;
    const JpegFrameDesc& d = s.h_descs[i];
    if (!(d.ncomp == 3 && d.color == kColorYCbCr && d.h[0] == 2 && (d.v[0] == 2 || d.v[0] == 1) && d.h[1] == 1 && d.v[1] == 1 &&
          d.h[2] == 1 && d.v[2] == 1 && d.dw[1] > 2))
      return false;
  }
  return true;
}

bool batch_is_420(const Slot& s, uint32_t count) {
  for (uint32_t i = 0; i < count; i++) {
    if (s.st[i] != UFD_OK) continue;
    const JpegFrameDesc& d = s.h_descs[i];
    if (!(d.ncomp == 3 && d.color == kColorYCbCr && d.h[0] == 2 && d.v[0] == 2 && d.h[1] == 1 && d.v[1] == 1 && d.h[2] == 1 &&
          d.v[2] == 1 && d.dw[1] > 2))
      return false;
  }
  return true;
}

// Decoded sample planes -> interleaved RGB8 frames in the context's d_rgb (tight pitch).
void enqueue_upsample_rgb(ufd_model* m, const Slot& s, const JpegFrameDesc* d_descs, uint32_t mw, uint32_t mh, uint32_t count) {
  bool fast = batch_is_420(s, count) && (m->rgb_stride % 4) == 0;
  for (uint32_t i = 0; i < count && fast; i++)
    if (s.st[i] == UFD_OK && (s.h_descs[i].width % 8) != 0) fast = false;
  ProfScope ps(m, fast ? "upsample_rgb_420" : "upsample_rgb", 0, 0);
  if (fast)
    launch_upsample_rgb_420(d_descs, tl_cur->d_planes, m->plane_stride, tl_cur->d_rgb, m->rgb_stride, mw, mh, count, tl_cur->stream);
  else
    launch_upsample_rgb(d_descs, tl_cur->d_planes, m->plane_stride, tl_cur->d_rgb, m->rgb_stride, mw, mh, count, tl_cur->stream);
}

// Coefficient slabs -> detections: IDCT, upsampling + colour + normalisation (+ resize), the
// network, head decode, NMS and the result copy, all on the context's stream.
int run_decoded(ufd_model* m, Slot& s, uint32_t count, bool any_ok, const JpegFrameDesc* d_descs_in, int16_t* d_coef_in, int buf) {
  int rc = UFD_OK;
  bool fused_stem = false;
  uint32_t max_blocks = 0, mw = 0, mh = 0;
  bool all_model_size = true;
  for (uint32_t i = 0; i < count; i++) {
    if (s.st[i] != UFD_OK) continue;
    const JpegFrameDesc& d = s.h_descs[i];
    max_blocks = std::max(max_blocks, d.total_blocks);
    mw = std::max(mw, (uint32_t)d.width);
    mh = std::max(mh, (uint32_t)d.height);
    if (d.width != m->W || d.height != m->H) all_model_size = false;
  }
  if (any_ok) {
    const JpegFrameDesc* d_descs = d_descs_in;
    int16_t* d_coef = d_coef_in;
    {
      ProfScope ps(m, "idct", 0, 0);
      launch_idct(d_descs, d_coef, m->coef_stride, tl_cur->d_planes, m->plane_stride, max_blocks, count, s.coef_zigzag, tl_cur->stream,
                  s.gpu_entropy ? tl_cur->sync.dc : nullptr, tl_cur->sync.dc_stride);
    }
    if (all_model_size && s.annot)  // N1 encodes the decoded frame: the RGB image the fused paths never make
      enqueue_upsample_rgb(m, s, d_descs, mw, mh, count);
    if (all_model_size) {
      // failed frames keep stale input; their results are never reported
      // camera streams are 4:2:0 YCbCr: specialised kernel when every decoded frame qualifies
      const bool all_420 = (m->W % 8) == 0 && batch_is_420(s, count);
      // 4:2:0 and 4:2:2 frames at the model size: the stem conv reads the sample planes itself (no f32 input
      // tensor); UFD_NO_STEM_FUSE=1 at ufd_create keeps the two-kernel path
      if ((m->W % 8) == 0 && m->stem_fusable && (all_420 || batch_is_h2(s, count))) {
        tl_cur->stem_descs = d_descs;
        fused_stem = true;
      } else {
        const double bytes = (double)count * (m->W * m->H * 1.5 + m->W * m->H * 12.0);
        ProfScope ps(m, all_420 ? "upsample_norm_420" : "upsample_norm", bytes, 0);
        if (all_420)
          launch_upsample_norm_420(d_descs, tl_cur->d_planes, m->plane_stride, m->d_lut, tl_cur->d_input, m->W, m->H, count, tl_cur->stream);
        else
          launch_upsample_norm(d_descs, tl_cur->d_planes, m->plane_stride, m->d_lut, tl_cur->d_input, m->W, m->H, count, tl_cur->stream);
      }
    } else {
      enqueue_upsample_rgb(m, s, d_descs, mw, mh, count);
      // one camera stream = one frame size: the whole batch in one launch (failed frames resample
      // stale pixels, their results are never reported); mixed sizes go frame by frame
      bool same_size = true;
      int sw = 0, sh = 0;
      for (uint32_t i = 0; i < count; i++) {
        if (s.st[i] != UFD_OK) continue;
        const JpegFrameDesc& d = s.h_descs[i];
        if (!sw) sw = d.width, sh = d.height;
        if (d.width != sw || d.height != sh) same_size = false;
      }
      if (same_size && sw && count > 1) {
        ResizeTaps v, h;
        rc = get_taps(m, sw, sh, &v, &h);
        if (rc) return rc;
        ProfScope ps(m, "resize_norm", (double)count * (3.0 * sw * sh + 12.0 * m->W * m->H), 0);
        launch_resize_norm(tl_cur->d_rgb, sw, sh, sw * 3, m->rgb_stride, v, h, m->d_lut, tl_cur->d_input, m->W, m->H, count,
                           tl_cur->stream);
      } else
      for (uint32_t i = 0; i < count; i++) {
        if (s.st[i] != UFD_OK) continue;
        const JpegFrameDesc& d = s.h_descs[i];
        float* dst = tl_cur->d_input + (size_t)i * 3 * m->W * m->H;
        const uint8_t* src = tl_cur->d_rgb + (size_t)i * m->rgb_stride;
        ProfScope ps(m, "resize_norm", 0, 0);
        if (d.width == m->W && d.height == m->H) {
          launch_norm_only(src, d.width, d.height, d.width * 3, 0, m->d_lut, dst, 1, tl_cur->stream);
        } else {
          ResizeTaps v, h;
          rc = get_taps(m, d.width, d.height, &v, &h);
          if (rc) return rc;
          launch_resize_norm(src, d.width, d.height, d.width * 3, 0, v, h, m->d_lut, dst, m->W, m->H, 1, tl_cur->stream);
        }
      }
    }
    enqueue_forward(m, count, &s);
    if (fused_stem) tl_cur->stem_descs = nullptr;  // (the stem was the last reader of the descriptors / planes of this buffer)
    enqueue_heads(m, count);
    enqueue_nms(m, s, count);
    if (s.annot) {
      rc = enqueue_annotate(m, s, d_descs, mw, mh, count);
      if (rc) return rc;
    }
  }
  return enqueue_results_copy(m, s, count);
}

// `count` same-size RGB frames already in d_rgb (tight pitch) -> pipeline
int run_rgb_on_device(ufd_model* m, Slot& s, uint32_t w, uint32_t h, uint32_t count) {
  span_begin(s);
  {
    ProfScope ps(m, "resize_norm", 0, 0);
    if ((int)w == m->W && (int)h == m->H) {
      launch_norm_only(tl_cur->d_rgb, w, h, w * 3, m->rgb_stride, m->d_lut, tl_cur->d_input, count, tl_cur->stream);
    } else {
      ResizeTaps v, hz;
      int rc = get_taps(m, w, h, &v, &hz);
      if (rc) return rc;
      launch_resize_norm(tl_cur->d_rgb, w, h, w * 3, m->rgb_stride, v, hz, m->d_lut, tl_cur->d_input, m->W, m->H, count, tl_cur->stream);
    }
  }
  enqueue_forward(m, count);
  enqueue_heads(m, count);
  enqueue_nms(m, s, count);
  return enqueue_results_copy(m, s, count);
}

int upload_rgb(ufd_model* m, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t pitch, uint32_t count) {
  if (!rgb || !w || !h || pitch < 3 * w) return m->fail(UFD_E_ARG, "bad RGB frame arguments");
  if (w > m->max_w || h > m->max_h) return m->fail(UFD_E_TOO_LARGE, "frame larger than max_src_width/height");
  if (count > m->B) return m->fail(UFD_E_TOO_LARGE, "batch larger than max_batch");
  for (uint32_t i = 0; i < count; i++)
    HIPC(m, hipMemcpy2DAsync(tl_cur->d_rgb + (size_t)i * m->rgb_stride, (size_t)w * 3, rgb + (size_t)i * h * pitch, pitch,
                             (size_t)w * 3, h, hipMemcpyHostToDevice, tl_cur->stream));
  return UFD_OK;
}

// Worker of one context: issues queued batches (host entropy stage + every GPU enqueue) so that
// the caller's submit returns at once and the host work of consecutive batches runs in parallel
// on the two contexts.
void worker_main(ufd_model* m, Worker* w) {
  pin_this_thread(m);
  (void)hipSetDevice(m->cfg.device_id);
  tl_cur = w->ctx;
  tl_pool = w->pool.get();
  tl_worker = w;
  for (;;) {
    Slot* s = nullptr;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return w->stop || !w->q.empty(); });
      if (w->q.empty()) return;  // stop requested and nothing left to issue
      s = w->q.front();
    }
    tl_prof = s->job_prof;
    int rc = UFD_OK;
    const uint64_t t_busy0 = now_ns();
    const uint64_t launches0 = tl_launches;
    try {
      HostScope hs(m, "host_issue");  // everything the worker does for one batch (host_plan included)
      rc = s->job_staged ? submit_staged(m, *s, *s->job_staged) : submit_jpegs(m, *s, s->job_jpegs, s->job_lens, s->count);
    } catch (const std::exception& e) {
      rc = m->fail(UFD_E_DEVICE, std::string("exception: ") + e.what());
    } catch (...) {
      rc = m->fail(UFD_E_DEVICE, "unknown exception");
    }
    gate_close(m, *s);
    w->ns_busy.fetch_add(now_ns() - t_busy0, std::memory_order_relaxed);
    w->launches.fetch_add(tl_launches - launches0, std::memory_order_relaxed);
    w->batches.fetch_add(1, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> lk(w->mu);
      s->issue_rc = rc;
      if (rc != UFD_OK) {
        std::lock_guard<std::mutex> lk2(m->err_mu);
        s->issue_err = m->err;
      }
      s->state = 2;
      w->q.pop_front();
    }
    w->cv.notify_all();
  }
}

// Blocks until the worker has issued the slot's batch (state 2).
void wait_issued(ufd_model* m, Slot& s) {
  Worker& w = m->workers[s.ctx - m->ctx];
  std::unique_lock<std::mutex> lk(w.mu);
  w.cv.wait(lk, [&] { return s.state != 1; });
}

// Synchronous entry points run on context 0 from the calling thread: wait until its worker has
// nothing queued (stream order then protects the context's buffers).
void drain_worker0(ufd_model* m) {
  Worker& w = m->workers[0];
  std::unique_lock<std::mutex> lk(w.mu);
  w.cv.wait(lk, [&] { return w.q.empty(); });
}

void destroy(ufd_model* m) {
  if (!m) return;
  for (Worker& w : m->workers) {
    if (!w.th.joinable()) continue;
    {
      std::lock_guard<std::mutex> lk(w.mu);
      w.stop = true;
    }
    w.cv.notify_all();
    w.th.join();
  }
  for (Ctx& c : m->ctx)
    if (c.stream) (void)hipStreamSynchronize(c.stream);
  auto dfree = [](void* p) {
    if (p) (void)hipFree(p);
  };
  dfree(m->d_weights), dfree(m->d_priors), dfree(m->d_lut), dfree(m->d_sync_luts);
  for (Ctx& c : m->ctx) {
    dfree(c.d_arena), dfree(c.d_input), dfree(c.d_status);
    for (int i = 0; i < 2; i++) {
      dfree(c.d_stage_buf[i]), dfree(c.d_coef_buf[i]);
    }
    for (auto& pair : c.ev_span) {
      for (auto& e : pair)
        if (e) (void)hipEventDestroy(e);
    }
    dfree(c.d_planes), dfree(c.d_rgb), dfree(c.d_sync);
    dfree(c.d_scores), dfree(c.d_boxes), dfree(c.d_keys), dfree(c.d_counts);  // (d_ndet lives behind d_status)
    dfree(c.d_spill);
    dfree(c.d_nms_mat);
    dfree(c.enc.planes), dfree(c.enc.coef), dfree(c.enc.bits), dfree(c.enc.total_bits), dfree(c.enc.words), dfree(c.enc.chunk_ff);
    dfree(c.d_enc_tables), dfree(c.d_enc_descs);
    for (auto& q : c.enc_setups) dfree(q.d_header);
    dfree(c.d_label_ops), dfree(c.d_glyphs), dfree(c.d_coverage);
  }
  for (float* t : m->tap_buf) dfree(t);
  for (auto& kv : m->taps)
    for (TapsDev* t : {&kv.second.first, &kv.second.second}) dfree(t->left), dfree(t->cnt), dfree(t->w);
  for (auto& s : m->slots) {
    if (s.h_stage) (void)hipHostFree(s.h_stage);
    if (s.h_coef) (void)hipHostFree(s.h_coef);
    if (s.h_dets) (void)hipHostFree(s.h_dets);
    if (s.d_dets) (void)hipFree(s.d_dets);
    if (s.h_gpu_status) (void)hipHostFree(s.h_gpu_status);  // (h_ndet lives behind it)
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.d_enc_out) (void)hipFree(s.d_enc_out);
    if (s.d_enc_meta) (void)hipFree(s.d_enc_meta);
    if (s.h_enc_meta) (void)hipHostFree(s.h_enc_meta);
    if (s.enc_copied) (void)hipEventDestroy(s.enc_copied);
  }
  gate_destroy(m);
  for (auto& pe : m->prof_pending) m->prof_free.push_back(pe.e0), m->prof_free.push_back(pe.e1);
  for (auto e : m->prof_free) (void)hipEventDestroy(e);
  for (Ctx& c : m->ctx) {
    if (c.stream) (void)hipStreamDestroy(c.stream);
  }
  delete m;
}

int create(const ufd_config* cfg, ufd_model** out) {
  if (!cfg || !out || cfg->struct_size != sizeof(ufd_config)) {
    g_create_error = "ufd_create: null argument or struct_size mismatch";
    return UFD_E_ARG;
  }
  if (cfg->variant != 640 && cfg->variant != 320) {
    g_create_error = "ufd_create: variant must be 640 or 320";
    return UFD_E_ARG;
  }
  if (cfg->max_batch < 1 || cfg->max_batch > 1024) {
    g_create_error = "ufd_create: max_batch must be in 1..1024";
    return UFD_E_ARG;
  }
  if (const char* knob = stray_experiment_knob()) {  // (before the device check: a CPU test covers it)
    g_create_error = std::string(knob) + " is set, but this build of libufacehip has no experiment hooks (make EXPERIMENTS=1 builds the one that has)";
    return UFD_E_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_error = "no HIP device: libufacehip needs a gfx950 GPU (there is no CPU fallback)";
    return UFD_E_DEVICE;
  }
  if (cfg->device_id < 0 || cfg->device_id >= ndev) {
    g_create_error = "ufd_create: device_id out of range";
    return UFD_E_ARG;
  }
  ufd_model* m = new ufd_model();
  auto bail = [&](int rc) {
    g_create_error = m->err;
    destroy(m);
    return rc;
  };
  m->cfg = *cfg;
  m->W = cfg->variant == 640 ? 640 : 320;  // UltrafaceVariant::width_height, nn.rs:36-41
  m->H = cfg->variant == 640 ? 480 : 240;
  m->B = cfg->max_batch;
  m->max_w = cfg->max_src_width ? cfg->max_src_width : 1920;
  m->max_h = cfg->max_src_height ? cfg->max_src_height : 1088;
  m->max_w = std::max<uint32_t>(m->max_w, m->W);
  m->max_h = std::max<uint32_t>(m->max_h, m->H);
  m->profile = (cfg->flags & UFD_FLAG_PROFILE) != 0;
  if (const char* e = experiment_env("UFD_PLAN_PARALLEL")) m->plan_parallel = std::atoi(e) != 0;
  m->force_sub_floor = (cfg->flags & UFD_FLAG_SUBSEQ_32) ? 32u : ((cfg->flags & UFD_FLAG_SUBSEQ_64) ? 64u : 0u);
  unsigned threads = cfg->host_threads ? cfg->host_threads : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
  m->host_threads = threads;
#define HIPB(expr)                                                                      \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      m->err = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
      return bail(UFD_E_DEVICE);                                                        \
    }                                                                                   \
  } while (0)
  HIPB(hipSetDevice(cfg->device_id));
  {
    hipDeviceProp_t prop;
    HIPB(hipGetDeviceProperties(&prop, cfg->device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      m->err = std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only";
      return bail(UFD_E_DEVICE);
    }
  }
  resolve_placement(m);
  m->pool.reset(new ThreadPool(std::min(threads, 8u), [m] { pin_this_thread(m); }));  // synchronous entry points and taps
  // Four contexts = four streams = the runtime's four hardware queues, one each (a fifth stream would share a queue with
  // a context and serialise with its kernels: no copy stream, see k_stage_in above).
  // HIGHEST stream priority (round 6): a process may hold another copy of the HIP runtime (torch's wheels bundle one), and with
  // default-priority streams a handle created behind that runtime's first touch of the device -- one 32-byte copy -- ran 21 %
  // slower for the life of the process (65.4 k -> 51.5 k frames/s); at the highest priority the order does not matter (65.6 k
  // either way, and unchanged without the other runtime): tools/ab/r6_torch_queue.py, profiles/r6e/torch_queue4.txt.
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  for (int ci = 0; ci < m->num_ctx; ci++) HIPB(hipStreamCreateWithPriority(&m->ctx[ci].stream, hipStreamNonBlocking, prio_hi));
  if (gate_init(m) != UFD_OK) {
    m->err = "hipEventCreate failed (pipeline gate)";
    return bail(UFD_E_DEVICE);
  }

  // ---- weights + priors
  std::vector<float> blob, priors;
  if (cfg->weights) {
    if (cfg->weights_floats != total_weight_floats()) {
      m->err = "weights blob must hold " + std::to_string(total_weight_floats()) + " floats";
      return bail(UFD_E_WEIGHTS);
    }
    blob.assign(cfg->weights, cfg->weights + cfg->weights_floats);
    if (cfg->priors) priors.assign(cfg->priors, cfg->priors + cfg->priors_floats);
  } else {
    std::string path = cfg->weights_path ? cfg->weights_path : default_weights_path(cfg->variant);
    std::string why;
    if (!load_ultraface_onnx(path, m->W, m->H, &blob, &priors, &why)) {
      m->err = "cannot load " + path + ": " + why;
      return bail(UFD_E_WEIGHTS);
    }
  }
  {
    std::vector<float> gen;
    gen_priors(m->W, m->H, gen);
    m->K = (int)gen.size() / 4;
    if (priors.empty()) priors = gen;
    if (priors.size() != gen.size()) {
      m->err = "priors must hold " + std::to_string(gen.size()) + " floats";
      return bail(UFD_E_WEIGHTS);
    }
  }
  plan_tensors(m, (cfg->flags & UFD_FLAG_KEEP_LAYERS) != 0);
  {
    const Layer& L0 = m->layers[0];
    ConvArgs probe{};
    probe.k = L0.spec.k, probe.stride = L0.spec.stride, probe.dil = L0.spec.dil, probe.pad = L0.spec.pad;
    probe.cin = L0.spec.cin, probe.cout = L0.spec.cout, probe.depthwise = L0.spec.groups > 1;
    probe.ih = L0.ih, probe.iw = L0.iw, probe.oh = L0.oh, probe.ow = L0.ow;
    m->stem_fusable = L0.kind == kKindConv3x3 && L0.leader == 0 && stem_planes_supported(probe) && !(cfg->flags & UFD_FLAG_NO_STEM_FUSE);
  }
  // shapes the kernels rely on (checked here once, not per launch)
  for (const Layer& L : m->layers) {
    if ((L.kind == kKindPointwise || L.kind == kKindDwPw || L.kind == kKindDwPw2) && (((L.oh * L.ow) & 3) || (L.spec.cin & 1))) {
      m->err = std::string("layer ") + L.spec.name + ": pointwise kernel needs H*W % 4 == 0 and even Cin";
      return bail(UFD_E_WEIGHTS);
    }
  }
  int rc = upload_weights(m, blob.data());
  if (rc) return bail(rc);
  m->priors_floats = priors.size();
  HIPB(hipMalloc(&m->d_priors, priors.size() * sizeof(float)));
  HIPB(hipMemcpy(m->d_priors, priors.data(), priors.size() * sizeof(float), hipMemcpyHostToDevice));
  {
    // (v as f32 / 255.0 - mean) / std with f32 literals and IEEE division (nn.rs:86-88)
    static const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    std::vector<float> lut(768);
    for (int c = 0; c < 3; c++)
      for (int v = 0; v < 256; v++) {
        volatile float q = (float)v / 255.0f;
        volatile float d = q - mean[c];
        lut[c * 256 + v] = d / stdv[c];
      }
    HIPB(hipMalloc(&m->d_lut, 768 * sizeof(float)));
    HIPB(hipMemcpy(m->d_lut, lut.data(), 768 * sizeof(float), hipMemcpyHostToDevice));
  }
  // ---- buffers
  const size_t B = m->B;
  {
    // worst-case slab: every component at full resolution, padded to the largest MCU there is (sampling factor 4 = 32 pixels)
    const size_t pw = (m->max_w + 31) / 32 * 32, ph = (m->max_h + 31) / 32 * 32;
    m->coef_stride = pw * ph * 3;
    m->plane_stride = pw * ph * 3;
    m->rgb_stride = ((size_t)m->max_w * m->max_h * 3 + 15) & ~(size_t)15;
  }
  m->key_stride = 1;
  while (m->key_stride < (size_t)m->K) m->key_stride <<= 1;
  // a JPEG is rarely larger than one byte per pixel; bigger frames take the host entropy path
  m->blob_stride = (((size_t)m->max_w * m->max_h) + 64 + 4095) & ~(size_t)4095;
  m->iv_cap = (uint32_t)B * 160;
  m->scans_off = (sizeof(JpegFrameDesc) * B + 255) & ~(size_t)255;
  m->ivs_off = (m->scans_off + sizeof(HuffScan) * B + 255) & ~(size_t)255;
  m->stage_cap = ((m->ivs_off + sizeof(HuffInterval) * m->iv_cap + 255) & ~(size_t)255) + m->blob_stride * B;
  m->gpu_entropy_enabled = (cfg->flags & UFD_FLAG_HOST_ENTROPY) == 0;
  // The convolution kernels address a tensor with a wave-uniform base and 32-bit per-lane byte offsets: every
  // activation tensor of a batch (and the network input) must stay below 4 GiB.
  {
    size_t worst = (size_t)3 * m->W * m->H;
    for (const auto& t : m->tensors) worst = std::max(worst, t.per_frame());
    if (worst * (size_t)B * sizeof(float) >= ((size_t)1 << 32)) {
      m->err = "max_batch " + std::to_string(B) + " is too large for this variant: an activation tensor would reach 4 GiB (limit " +
               std::to_string((((size_t)1 << 32) - 1) / (worst * sizeof(float))) + ")";
      return bail(UFD_E_TOO_LARGE);
    }
  }
  HIPB(hipMalloc(&m->d_sync_luts, sizeof(SyncLutImage) * ufd_model::kMaxLutSets));
  for (int ci = 0; ci < m->num_ctx; ci++) {
    Ctx& c = m->ctx[ci];
    HIPB(hipMalloc(&c.d_status, sizeof(uint32_t) * 2 * B));  // [B decode statuses][B detection counts]
    HIPB(hipMemset(c.d_status, 0, sizeof(uint32_t) * 2 * B));
    c.d_ndet = c.d_status + B;
    HIPB(hipMalloc(&c.d_arena, std::max<size_t>(m->arena_floats, 64) * sizeof(float)));
    HIPB(hipMalloc(&c.d_input, B * 3 * m->W * m->H * sizeof(float)));
    for (int i = 0; i < 2; i++) {
      HIPB(hipMalloc(&c.d_stage_buf[i], m->stage_cap));
      c.d_descs_buf[i] = reinterpret_cast<JpegFrameDesc*>(c.d_stage_buf[i]);
      HIPB(hipMalloc(&c.d_coef_buf[i], sizeof(int16_t) * m->coef_stride * B));
    }
    for (auto& pair : c.ev_span)
      for (auto& e : pair) HIPB(hipEventCreate(&e));
    if (m->gpu_entropy_enabled) {
      HIPB(hipMalloc(&c.d_sync, sync_buffers_bytes((uint32_t)B, m->blob_stride, m->coef_stride / 64, nullptr)));
      c.sync.stream = c.d_sync;
      sync_buffers_bytes((uint32_t)B, m->blob_stride, m->coef_stride / 64, &c.sync);
    }
    HIPB(hipMalloc(&c.d_planes, m->plane_stride * B));
    HIPB(hipMalloc(&c.d_rgb, m->rgb_stride * B));
    HIPB(hipMalloc(&c.d_scores, B * m->K * 2 * sizeof(float)));
    HIPB(hipMalloc(&c.d_boxes, B * m->K * 4 * sizeof(float)));
    HIPB(hipMalloc(&c.d_keys, B * m->key_stride * sizeof(unsigned long long)));
    HIPB(hipMalloc(&c.d_counts, B * sizeof(uint32_t)));
    HIPB(hipMemset(c.d_counts, 0, B * sizeof(uint32_t)));
    HIPB(hipMalloc(&c.d_spill, B * m->K * sizeof(float4)));
    HIPB(hipMalloc(&c.d_nms_mat, nms_matrix_bytes((uint32_t)B, (uint32_t)m->K)));
  }
  if (cfg->flags & UFD_FLAG_TAP_LAYERS) {
    m->tap_buf.assign(m->tensors.size(), nullptr);
    for (size_t t = 0; t < m->tensors.size(); t++)
      HIPB(hipMalloc(&m->tap_buf[t], std::max<size_t>(m->tensors[t].per_frame() * B, 64) * sizeof(float)));
  }
  HIPB(hipDeviceSynchronize());
#undef HIPB
  for (int ci = 0; ci < m->num_ctx; ci++) {
    Worker& w = m->workers[ci];
    w.ctx = &m->ctx[ci];
    w.pool.reset(new ThreadPool(std::max(1u, m->host_threads / (unsigned)m->num_ctx), [m] { pin_this_thread(m); }));
    w.th = std::thread(worker_main, m, &w);
  }
  // pinned staging of every slot now: a first-use allocation (milliseconds) would land inside the
  // caller's first batches
  for (auto& sl : m->slots) {
    const int rc = alloc_slot(m, sl);
    if (rc) {
      g_create_error = m->err;
      destroy(m);
      return rc;
    }
  }
  m->stats_t0 = now_ns();
  *out = m;
  return UFD_OK;
}

}  // namespace ufd

// =================================================================== C ABI
extern "C" {

int ufd_create(const ufd_config* cfg, ufd_model** out) {
  try {
    return create(cfg, out);
  } catch (const std::exception& e) {
    g_create_error = std::string("exception: ") + e.what();
    return UFD_E_DEVICE;
  } catch (...) {
    g_create_error = "unknown exception";
    return UFD_E_DEVICE;
  }
}

void ufd_destroy(ufd_model* m) {
  try {
    if (m) (void)hipSetDevice(m->cfg.device_id);
    destroy(m);
  } catch (...) {
  }
}

const char* ufd_last_error(const ufd_model* m) { return m ? m->err.c_str() : g_create_error.c_str(); }

int ufd_model_info(const ufd_model* m, uint32_t* width, uint32_t* height, uint32_t* num_priors) {
  if (!m) return UFD_E_ARG;
  if (width) *width = m->W;
  if (height) *height = m->H;
  if (num_priors) *num_priors = m->K;
  return UFD_OK;
}

int ufd_model_placement(const ufd_model* m, int32_t* device_id, int32_t* numa_node, uint32_t* pinned_cpus, char* pci_bdf,
                        size_t pci_cap, char* cpu_list, size_t cpu_cap) {
  if (!m) return UFD_E_ARG;
  if (device_id) *device_id = m->cfg.device_id;
  if (numa_node) *numa_node = m->numa_node;
  if (pinned_cpus) *pinned_cpus = (uint32_t)m->pin_cpus.size();
  if (pci_bdf && pci_cap) std::snprintf(pci_bdf, pci_cap, "%s", m->pci_bdf.c_str());
  if (cpu_list && cpu_cap) std::snprintf(cpu_list, cpu_cap, "%s", m->cpu_list.c_str());
  return UFD_OK;
}

int ufd_prime_device(int32_t device_id) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_error = "no HIP device: libufacehip needs a gfx950 GPU (there is no CPU fallback)";
    return UFD_E_DEVICE;
  }
  if (device_id < 0 || device_id >= ndev) {
    g_create_error = "ufd_prime_device: device_id out of range";
    return UFD_E_ARG;
  }
  int saved = -1;
  if (hipGetDevice(&saved) != hipSuccess) saved = -1;
  int rc = UFD_OK;
  if (hipSetDevice(device_id) != hipSuccess) rc = UFD_E_DEVICE;
  hipStream_t st[4] = {};
  for (int i = 0; i < 4 && rc == UFD_OK; i++)
    if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) rc = UFD_E_DEVICE;
  for (int i = 0; i < 4 && rc == UFD_OK; i++) launch_copy_fence(st[i]);
  for (int i = 0; i < 4; i++)
    if (st[i]) {
      if (hipStreamSynchronize(st[i]) != hipSuccess) rc = UFD_E_DEVICE;
      (void)hipStreamDestroy(st[i]);
    }
  if (saved >= 0) (void)hipSetDevice(saved);
  if (rc != UFD_OK) g_create_error = "ufd_prime_device: the HIP runtime refused a stream or a launch";
  return rc;
}

int ufd_model_limits(const ufd_model* m, uint32_t* max_batch, uint32_t* max_src_width, uint32_t* max_src_height) {
  if (!m) return UFD_E_ARG;
  if (max_batch) *max_batch = m->B;
  if (max_src_width) *max_src_width = m->max_w;
  if (max_src_height) *max_src_height = m->max_h;
  return UFD_OK;
}

int ufd_infer_rgb_batch(ufd_model* m, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t pitch, uint32_t count,
                        ufd_det* out, uint32_t cap, uint32_t* n) {
  return guarded(m, [&]() -> int {
    drain_worker0(m);
    int rc = check_outputs(m, out, cap, n);
    if (rc) return rc;
    if (!count) return UFD_OK;
    Slot* s = find_free_slot(m);
    if (!s) return m->fail(UFD_E_STATE, "all slots busy: call ufd_wait first");
    rc = alloc_slot(m, *s);
    if (rc) return rc;
    rc = upload_rgb(m, rgb, w, h, pitch, count);
    if (rc) return rc;
    s->count = count, s->cap = cap, s->out = out, s->n = n, s->status = nullptr;
    s->gpu_entropy = false;
    std::fill(s->st.begin(), s->st.begin() + count, UFD_OK);
    rc = run_rgb_on_device(m, *s, w, h, count);
    if (rc) return rc;
    return finish_slot(m, *s);
  });
}

int ufd_infer_rgb(ufd_model* m, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t pitch, ufd_det* out, uint32_t cap,
                  uint32_t* n) {
  return ufd_infer_rgb_batch(m, rgb, w, h, pitch, 1, out, cap, n);
}

static int submit_common(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, const ufd_staged* staged, uint32_t count,
                         ufd_det* out, uint32_t cap, uint32_t* n, int32_t* status, uint32_t* ticket,
                         const ufd_annotate* annot = nullptr) {
  return guarded(m, [&]() -> int {
    int rc = check_outputs(m, out, cap, n);
    if (rc) return rc;
    if (annot) {
      if (annot->struct_size != sizeof(ufd_annotate)) return m->fail(UFD_E_ARG, "ufd_annotate.struct_size mismatch");
      if ((!annot->jpeg_out && annot->jpeg_cap) || !annot->jpeg_off || !annot->jpeg_len)
        return m->fail(UFD_E_ARG, "null annotate output pointer");
      if (annot->quality < 1 || annot->quality > 100) return m->fail(UFD_E_ARG, "quality must be in 1..100");
    }
    if ((!staged && (!jpegs || !lens)) || !ticket) return m->fail(UFD_E_ARG, "null argument");
    if (count < 1 || count > m->B) return m->fail(UFD_E_TOO_LARGE, "count must be in 1..max_batch");
    Slot* s = find_free_slot(m);
    if (!s) return m->fail(UFD_E_STATE, "all slots busy: call ufd_wait first");
    rc = alloc_slot(m, *s);
    if (rc) return rc;
    // alternate contexts: consecutive batches overlap on the GPU, and their host stages run in
    // parallel on the contexts' worker threads
    Worker& w = m->workers[m->next_ctx];
    m->next_ctx = (m->next_ctx + 1) % m->num_ctx;
    s->cap = cap, s->out = out, s->n = n, s->status = status;
    s->count = count;
    s->job_jpegs = jpegs, s->job_lens = lens, s->job_staged = staged;
    s->job_prof = (m->prof_batch++ % m->prof_every) == 0;
    s->issue_rc = UFD_OK;
    s->annot = annot != nullptr, s->annot_ran = false;
    if (annot) s->annot_args = *annot;
    s->ctx = w.ctx;
    s->busy = true;
    s->seq = m->next_seq++, s->gate_published = false;
    s->ticket = m->next_ticket++;
    if (!m->next_ticket) m->next_ticket = 1;
    *ticket = s->ticket;
    {
      std::lock_guard<std::mutex> lk(w.mu);
      s->state = 1;
      w.q.push_back(s);
    }
    w.cv.notify_all();
    return UFD_OK;
  });
}

int ufd_submit_jpeg_batch(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, uint32_t count, ufd_det* out,
                          uint32_t cap, uint32_t* n, int32_t* status, uint32_t* ticket) {
  return submit_common(m, jpegs, lens, nullptr, count, out, cap, n, status, ticket);
}

int ufd_submit_annotate_batch(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, uint32_t count,
                              const ufd_annotate* annot, ufd_det* out, uint32_t cap, uint32_t* n, int32_t* status,
                              uint32_t* ticket) {
  if (m && !annot) return m->fail(UFD_E_ARG, "null ufd_annotate");
  return submit_common(m, jpegs, lens, nullptr, count, out, cap, n, status, ticket, annot);
}

int ufd_annotate_jpeg_batch(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, uint32_t count,
                            const ufd_annotate* annot, ufd_det* out, uint32_t cap, uint32_t* n, int32_t* status) {
  if (m && count == 0) return UFD_OK;
  uint32_t ticket = 0;
  std::vector<int32_t> local;
  if (!status) {
    local.resize(count ? count : 1);
    status = local.data();
  }
  int rc = ufd_submit_annotate_batch(m, jpegs, lens, count, annot, out, cap, n, status, &ticket);
  if (rc) return rc;
  return ufd_wait(m, ticket);
}


void* ufd_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) return nullptr;  // (every GPU of the process may write it)
  return p;
}
void* ufd_model_host_alloc(ufd_model* m, size_t bytes) {
  if (!m) return nullptr;
  int saved = -1;  // the caller's current device is put back (the scheduler calls this on its caller's thread)
  if (hipGetDevice(&saved) != hipSuccess) saved = -1;
  if (hipSetDevice(m->cfg.device_id) != hipSuccess) return nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) p = nullptr;
  if (saved >= 0 && saved != m->cfg.device_id) (void)hipSetDevice(saved);
  return p;
}
void ufd_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int ufd_stage_jpeg_batch(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, uint32_t count, ufd_staged** staged) {
  if (staged) *staged = nullptr;
  return guarded(m, [&]() -> int {
    if (!jpegs || !lens || !staged) return m->fail(UFD_E_ARG, "null argument");
    if (count < 1 || count > m->B) return m->fail(UFD_E_TOO_LARGE, "count must be in 1..max_batch");
    if (!m->gpu_entropy_enabled) return m->fail(UFD_E_STATE, "staging needs the device entropy decoder (handle created with UFD_FLAG_HOST_ENTROPY)");
    drain_worker0(m);
    // host plan in the pinned arrays of a free slot, then blocking uploads into the staged batch
    Slot* fs = find_free_slot(m);
    if (!fs) return m->fail(UFD_E_STATE, "all slots busy: call ufd_wait first");
    int rc = alloc_slot(m, *fs);
    if (rc) return rc;
    Slot& tmp = *fs;
    std::unique_ptr<ufd_staged> g(new ufd_staged);
    auto release = [] {};
    const DevicePlan p = plan_device_entropy(m, tmp, jpegs, lens, count);
    if (!p.ok) {
      release();
      return m->fail(UFD_E_UNSUPPORTED, "batch not eligible for the device entropy decoder (progressive, multi-scan or oversized frame)");
    }
    g->count = count;
    g->plan = p;
    g->h_descs.assign(tmp.h_descs, tmp.h_descs + count);
    g->st.assign(tmp.st.begin(), tmp.st.begin() + count);
    bool ok = hipMalloc(&g->d_stage, p.stage_bytes) == hipSuccess &&
              hipMemcpy(g->d_stage, tmp.h_stage, p.stage_bytes, hipMemcpyHostToDevice) == hipSuccess;
    release();
    if (!ok) {
      (void)hipFree(g->d_stage);
      return m->fail(UFD_E_DEVICE, "staging a batch in device memory failed");
    }
    g->h_scans.assign(tmp.h_scans, tmp.h_scans + count);
    pin_lut_sets(m, g->h_scans.data(), count, +1);
    g->d_descs = reinterpret_cast<JpegFrameDesc*>(g->d_stage);
    g->d_scans = reinterpret_cast<HuffScan*>(g->d_stage + m->scans_off);
    g->d_ivs = reinterpret_cast<HuffInterval*>(g->d_stage + m->ivs_off);
    g->d_blob = g->d_stage + p.blob_base;
    *staged = g.release();
    return UFD_OK;
  });
}

int ufd_submit_staged(ufd_model* m, const ufd_staged* staged, ufd_det* out, uint32_t cap, uint32_t* n, int32_t* status,
                      uint32_t* ticket) {
  if (m && !staged) return m->fail(UFD_E_ARG, "null staged batch");
  return submit_common(m, nullptr, nullptr, staged, staged ? staged->count : 0, out, cap, n, status, ticket);
}

void ufd_staged_free(ufd_model* m, ufd_staged* staged) {
  if (!staged) return;
  if (m) (void)hipSetDevice(m->cfg.device_id);
  if (m) pin_lut_sets(m, staged->h_scans.data(), (uint32_t)staged->h_scans.size(), -1);
  (void)hipFree(staged->d_stage);
  delete staged;
}

int ufd_wait(ufd_model* m, uint32_t ticket) {
  if (!m) return UFD_E_ARG;
  Slot* s = nullptr;
  {
    std::lock_guard<std::mutex> lk(m->mu);
    for (auto& c : m->slots)
      if (c.busy && c.ticket == ticket) s = &c;
    if (!s) return m->fail(UFD_E_STATE, "unknown ticket");
    if (s->waiting) return m->fail(UFD_E_STATE, "another thread is already waiting for this ticket");
    s->waiting = true;
    uint32_t inflight = 0;
    for (auto& c : m->slots) inflight += c.busy ? 1u : 0u;
    s->relaxed_wait = inflight > 1 && !(m->cfg.flags & UFD_FLAG_SPIN_WAIT);
  }
  // not holding the handle lock while the worker and the GPU finish: other threads may submit
  {
    const uint64_t t0 = now_ns();
    wait_issued(m, *s);
    m->ns_wait.fetch_add(now_ns() - t0, std::memory_order_relaxed);
    m->waits.fetch_add(1, std::memory_order_relaxed);
  }
  try {
    (void)hipSetDevice(m->cfg.device_id);  // (a failure shows up in the event wait below, which also releases the slot)
    tl_cur = s->ctx;
    return finish_slot(m, *s, /*locked=*/false);  // (copies and waits outside m->mu; the slot is released under it)
  } catch (...) {
    return m->fail(UFD_E_DEVICE, "unknown exception");
  }
}

int ufd_infer_jpeg_batch(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, uint32_t count, ufd_det* out,
                         uint32_t cap, uint32_t* n, int32_t* status) {
  if (m && count == 0) return UFD_OK;
  uint32_t ticket = 0;
  std::vector<int32_t> local;
  if (!status) {
    local.resize(count ? count : 1);
    status = local.data();
  }
  int rc = ufd_submit_jpeg_batch(m, jpegs, lens, count, out, cap, n, status, &ticket);
  if (rc) return rc;
  return ufd_wait(m, ticket);
}

int ufd_infer_jpeg(ufd_model* m, const uint8_t* jpeg, size_t len, ufd_det* out, uint32_t cap, uint32_t* n,
                   uint32_t* img_w, uint32_t* img_h) {
  if (!m) return UFD_E_ARG;
  if (!jpeg || !len) {
    std::lock_guard<std::mutex> lk(m->mu);
    return m->fail(UFD_E_ARG, "null JPEG");
  }
  if (img_w || img_h) {
    JpegFrameDesc d;
    if (jpeg_parse_header(jpeg, len, &d) == kJpegOk) {
      if (img_w) *img_w = d.width;
      if (img_h) *img_h = d.height;
    }
  }
  int32_t st = 0;
  int rc = ufd_infer_jpeg_batch(m, &jpeg, &len, 1, out, cap, n, &st);
  if (rc) return rc;
  if (st != UFD_OK) {
    std::lock_guard<std::mutex> lk(m->mu);
    if (st == UFD_E_DECODE) m->err = "corrupt JPEG";
    if (st == UFD_E_UNSUPPORTED) m->err = "unsupported JPEG feature";
    if (st == UFD_E_TOO_LARGE) m->err = "frame larger than max_src_width/height";
    if (st == UFD_E_TRUNCATED) m->err = "more detections than cap";
  }
  return st;
}

}  // extern "C"

// ---------------------------------------------------------------- hooks of replicas.cpp (not part of the C ABI)
namespace ufd {
int create_handle(const ufd_config* cfg, ufd_model** out) { return create(cfg, out); }
void set_create_error(const std::string& msg) { g_create_error = msg; }
std::string get_create_error() { return g_create_error; }
void weight_buffers(ufd_model* m, float** d_weights, size_t* weight_floats, float** d_priors, size_t* prior_floats) {
  *d_weights = m->d_weights, *weight_floats = m->weight_img_floats;
  *d_priors = m->d_priors, *prior_floats = m->priors_floats;
}
}  // namespace ufd

