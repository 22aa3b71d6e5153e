// model.cpp -- host side of libufacehip.so: the C ABI of include/ufd.h and the per-batch
// pipeline that replaces the reference's single Inferer task (infer_server/src/inferer.rs:29-50):
//   host workers: marker parse + Huffman decode -> coefficient slabs in pinned memory
//   GPU (one HIP stream per handle): IDCT -> upsample/colour(+normalise) [-> Triangle resize]
//        -> 52 convolutions -> softmax/prior decode/threshold -> sort + greedy NMS
//   D2H: a few hundred bytes of detections per frame.
// Weights (1.1 MB) and priors stay resident in HBM for the life of the handle.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <pthread.h>
#include <time.h>
#include <sched.h>

#include <fstream>
#include <functional>

#include "../../include/ufd.h"
#include "jpeg_host.hpp"
#include "kernels.hpp"
#include "onnx_loader.hpp"
#include "thread_pool.hpp"
#include "topology.hpp"
#include "model_internal.hpp"

namespace ufd {
namespace {

thread_local std::string g_create_error;

struct Ctx;
// Per-thread view of the handle: API threads use context 0 and the handle's pool; each context's
// worker thread uses its own context and pool.
thread_local Ctx* tl_cur = nullptr;
thread_local ThreadPool* tl_pool = nullptr;
thread_local bool tl_prof = true;  // record kernel events for the batch being issued by this thread
thread_local bool tl_force_rider = false;  // enqueue_layer_launch: issue a riding layer on its own (its host could not take it)
struct Worker;
thread_local Worker* tl_worker = nullptr;  // issue worker this thread is (host statistics go to it), or null on API threads
thread_local uint64_t tl_launches = 0;     // launches + copies enqueued by this thread (ProfScope counts them)

inline uint64_t now_ns() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Tensor {
  size_t off = 0;  // float offset in the activation arena (for the whole batch)
  int c = 0, h = 0, w = 0;
  // liveness in the issued plan, in layer turns: written at `first` (kNumConv: no launch writes it, it has no storage),
  // read last at `last` (kNumConv: until the head decode); ufd_debug_plan reports them, the CPU suite checks that no two
  // tensors share arena bytes while both are live
  int first = 0, last = -1;
  bool stored = false;
  size_t per_frame() const { return (size_t)c * h * w; }
};

enum LayerKind {
  kKindPointwise,  // 1x1 on fp32 MFMA
  kKindDwPw,       // 1x1 whose depthwise producer is fused in (the dw layer itself is kKindFusedAway)
  kKindDwPw2,      // second 1x1 of two chained dw->pw blocks run as one launch (Layer::chain_first)
  kKindFusedAway,  // depthwise layer computed inside the following kKindDwPw launch
  kKindConv3x3,    // dense 3x3 implicit GEMM on fp32 MFMA
  kKindDirect,     // VALU fallback
};

struct Layer {
  ConvSpec spec;
  int ih, iw, oh, ow;
  int in_tensor, out_tensor, res_tensor;
  int out_coff;
  LayerKind kind;
  int fused_dw = -1;       // kKindDwPw / kKindDwPw2: index of the depthwise layer
  int chain_first = -1;    // kKindDwPw2: the kKindDwPw layer of the first block
  bool chained = false;    // kKindDwPw layer computed inside a later kKindDwPw2 launch: its output never exists
  int leader = -1;         // first layer of the launch this layer is issued in (itself when not merged)
  int rider = -1;          // leader only: leader of an independent launch that rides in this one's grid (dual launch)
  int ride = -1;           // leader only: the launch it rides in (issued at that layer's turn, not at its own)
  int group[3] = {-1, -1, -1};  // leader only: members of its launch (itself first)
  bool materialize = true; // kKindFusedAway: also run the stand-alone kernel (KEEP_LAYERS debugging)
  const float* d_w = nullptr;  // kernel-specific packing
  const float* d_w_rows = nullptr;    // dense 3x3 layers: packing of the row kernel
  const float* d_w_dwpack = nullptr;  // depthwise layers: [c][12] image for the fused dw->pw kernel
  int sum_with = -1;                  // pointwise layer whose 1x1 conv is summed into this launch (RFB shortcut + ConvLinear)
  int stack[3] = {-1, -1, -1};        // 1x1 convs of the same input run as ONE conv with their output channels stacked (itself first)
  int in_coff = 0;                    // this layer reads channels [in_coff, in_coff + cin) of its input tensor
  const float* d_w_sum = nullptr;     // ... packed weights over both inputs' channels / the stacked output channels, and their bias
  const float* d_b_sum = nullptr;
  const float* d_w_tail = nullptr;    // RFB shortcut layer: ConvLinear + shortcut weights in k_rfb_tail's chunk order
  const float* d_b = nullptr;
  double bytes_per_frame = 0, flops_per_frame = 0, weight_bytes = 0;
  int tap_tensor = -1, tap_coff = 0;  // where this layer's output lives in the issued plan (-1: it never exists)
};

struct ProfEntry {
  int name_id;
  hipEvent_t e0, e1;
  double bytes, flops;
};

struct Slot {
  bool busy = false;
  bool waiting = false;  // a ufd_wait is finishing this slot outside the handle lock (guarded by ufd_model::mu)
  bool relaxed_wait = false;  // ufd_wait found other batches in flight behind this one: sleep between polls instead of spinning
  uint32_t ticket = 0, count = 0, cap = 0;
  JpegFrameDesc* h_descs = nullptr;
  int16_t* h_coef = nullptr;
  Det* h_dets = nullptr;
  Det* d_dets = nullptr;  // [B][K] detections of this slot's batch: stays valid until the slot is released (tail reads at ufd_wait)
  uint32_t* h_ndet = nullptr;
  uint32_t* h_gpu_status = nullptr;  // per frame: device entropy decoder flagged a corrupt stream
  // One pinned block per slot, copied to the device with ONE hipMemcpyAsync:
  //   [frame descriptors][scan layouts][restart intervals (n_iv)][JPEG bytes, frames packed back to back]
  uint8_t* h_stage = nullptr;
  uint8_t* h_blob = nullptr;        // = h_stage + blob_base of the batch (set by plan_device_entropy)
  HuffScan* h_scans = nullptr;
  HuffInterval* h_ivs = nullptr;
  std::vector<GpuScanPlan> plans;
  bool gpu_entropy = false;
  bool coef_zigzag = false;  // the slabs hold zigzag-ordered blocks (self-synchronising decoder)
  ufd_det* out = nullptr;
  uint32_t* n = nullptr;
  int32_t* status = nullptr;
  std::vector<int32_t> st;
  hipEvent_t done = nullptr;
  Ctx* ctx = nullptr;  // context whose stream produced this slot's results
  // asynchronous submission: the context's worker thread issues the batch
  const uint8_t* const* job_jpegs = nullptr;
  const size_t* job_lens = nullptr;
  const ufd_staged* job_staged = nullptr;  // non-null: the batch is resident in HBM
  bool job_prof = true;
  // N1 (ufd_submit_annotate_batch): rectangles + re-encode after NMS.  The finished streams of the batch land in the
  // slot's own device buffer (it stays valid until the slot is released: the host fetches them in ufd_wait, when
  // their total size is known) -- the encoder's scratch belongs to the context.
  bool annot = false, annot_ran = false;
  ufd_annotate annot_args{};
  uint8_t* d_enc_out = nullptr;
  size_t enc_out_cap = 0;
  uint32_t* d_enc_meta = nullptr;  // [B] length, [B] offset, [1] total
  uint32_t* h_enc_meta = nullptr;  // pinned copy
  hipEvent_t enc_copied = nullptr;
  // host statistics: which pair of the context's span events this batch recorded (-1: none, e.g. nothing decodable)
  int span_idx = -1;
  uint64_t span_seq = 0;
  int issue_rc = 0;          // result of the worker's entropy stage + enqueue
  std::string issue_err;
  int state = 0;             // 0 free, 1 queued for the worker, 2 issued to the GPU (guarded by Worker::mu)
};

struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Slot*> q;
  bool stop = false;
  Ctx* ctx = nullptr;
  std::unique_ptr<ThreadPool> pool;
  unsigned host_threads = 1;
  // host statistics (ufd_host_stats_read): written by the worker thread only
  std::atomic<uint64_t> ns_busy{0}, ns_plan{0}, ns_copy{0}, batches{0}, launches{0};
};

// Device-side working set of one in-flight batch.  A handle owns kNumCtx of them and alternates
// batches between them: their kernels run on different HIP streams, so the latency-bound stages
// of one batch (small feature maps, NMS) overlap the bandwidth-bound stages of the other.
struct Ctx {
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // the handle's one copy stream (shared by the contexts): H2D of the next batch overlaps kernels
  float* d_arena = nullptr;
  float* d_input = nullptr;
  JpegFrameDesc* d_descs_buf[2] = {nullptr, nullptr};  // double-buffered: copy(i+1) runs beside kernels(i)
  int16_t* d_coef_buf[2] = {nullptr, nullptr};
  // device entropy decoding: JPEG bytes, scan layouts and restart intervals of the batch
  uint8_t* d_stage_buf[2] = {nullptr, nullptr};  // device image of Slot::h_stage (d_descs_buf points at its head)
  uint8_t* d_sync = nullptr;  // scratch of the self-synchronising entropy decoder
  const JpegFrameDesc* stem_descs = nullptr;  // non-null: the next forward reads the 4:2:0 sample planes (fused stem)
  SyncBuffers sync;
  uint32_t* d_status = nullptr;
  hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
  bool consumed_valid[2] = {false, false};
  int flip = 0;
  uint8_t* d_planes = nullptr;
  uint8_t* d_rgb = nullptr;
  float* d_scores = nullptr;
  float* d_boxes = nullptr;
  unsigned long long* d_keys = nullptr;
  uint32_t* d_counts = nullptr;
  uint32_t* d_ndet = nullptr;
  float4* d_spill = nullptr;
  unsigned long long* d_nms_mat = nullptr;  // suppression matrices of frames with many candidates
  uint32_t last_forward_count = 0;
  // host statistics: begin / end events of the last kSpanRing batches of this context (timing enabled).  Batch j uses pair
  // j % kSpanRing; at most UFD_MAX_SLOTS batches are in flight, so pair j - 1 is still intact when batch j is finished.
  static constexpr int kSpanRing = 2 * UFD_MAX_SLOTS;
  hipEvent_t ev_span[kSpanRing][2] = {};
  uint64_t span_issued = 0;    // batches that recorded a span (issue worker / API thread under the handle lock)
  uint64_t span_last_done = 0; // 1 + sequence number of the last batch folded into the sums below (shared_mu)
  uint64_t gpu_batches = 0;
  double gpu_span_ms = 0, gpu_gap_ms = 0;
  // N1 encoder scratch, sized for the largest frame an annotate batch of this context has had (regrown when a larger
  // one arrives), and the (quality, framing) set-ups seen: quantiser + marker segments, each with its own device header,
  // so that streams of one model that differ in quality or framing alternate without a stream drain
  EncBuffers enc;
  bool enc_ready = false;
  size_t enc_mcus = 0;  // MCUs per frame the scratch holds
  struct EncSetup {
    int quality = -1, multipart = -1;
    EncQuant q{};
    bool ifast = true;
    uint8_t* d_header = nullptr;
    uint32_t pre_len = 0, hdr_len = 0, dim_off = 0, post_len = 0;
    uint64_t last_use = 0;
  };
  static constexpr int kEncSetups = 4;
  EncSetup enc_setups[kEncSetups];
  int enc_cur = 0;  // set-up of the batch being issued
  uint64_t enc_seq = 0;
  uint32_t* d_enc_tables = nullptr;
  JpegFrameDesc* d_enc_descs = nullptr;  // descriptors of frames that did not come out of the decoder (debug taps)
  void* d_label_ops = nullptr;           // one drawing operation per detection of the batch
  int* d_glyphs = nullptr;               // label glyph atlas (glyph_atlas.inc)
  float* d_coverage = nullptr;
};
constexpr int kMaxCtx = 8;

constexpr uint32_t kDetCopy = 256;  // detections per frame copied back with the batch

struct TapsDev {
  int32_t* left = nullptr;
  int32_t* cnt = nullptr;
  float* w = nullptr;
  int stride = 0;
};

}  // namespace
}  // namespace ufd

using namespace ufd;

struct ufd_model {
  std::mutex mu;
  std::string err;
  ufd_config cfg{};
  int W = 0, H = 0, K = 0;
  uint32_t B = 0;
  uint32_t max_w = 0, max_h = 0;
  Ctx ctx[kMaxCtx];
  Worker workers[kMaxCtx];
  std::atomic<uint64_t> ns_wait{0}, waits{0};
  uint64_t stats_t0 = 0;  // now_ns() of the last ufd_host_stats_reset (or of ufd_create)
  int num_ctx = 3;  // measured: 2 -> 34.0 k, 3 -> 40-42 k, 4 -> 40-41 k frames/s; each has its own stream pair
  int next_ctx = 0;
  std::mutex shared_mu;  // profiling tables, resize-tap cache, Huffman table-set cache
  std::mutex err_mu;     // error string
  std::mutex copy_mu;    // enqueue order on the shared copy stream
  std::unique_ptr<ThreadPool> pool;
  unsigned host_threads = 1;

  // host placement: NUMA node of the GPU's PCIe root and the CPUs of it this process may use; the handle's issue
  // workers and pool threads are pinned to them (8 ranks on a two-socket box must not stage JPEG bytes across sockets)
  int numa_node = -1;
  std::vector<int> pin_cpus;
  std::string pci_bdf, cpu_list;

  // resident model
  size_t weight_img_floats = 0, priors_floats = 0;
  float* d_weights = nullptr;
  float* d_priors = nullptr;
  float* d_lut = nullptr;
  std::vector<Layer> layers;
  std::vector<Tensor> tensors;
  size_t arena_floats = 0;

  // frame staging (device)
  size_t coef_stride = 0, plane_stride = 0, rgb_stride = 0;

  // post
  size_t key_stride = 0;

  // device entropy decoding: table sets seen so far (append-only, shared by the contexts)
  // Huffman table sets seen so far (per-camera optimised tables make new ones).  A full cache evicts the set that has
  // gone unused the longest, provided no batch that could still be in flight or staged refers to it.
  static constexpr int kMaxLutSets = 64;
  std::vector<std::array<HuffLut, 4>> lut_sets;
  struct LutMeta {
    uint64_t hash = 0, last_use = 0;  // content hash; plan sequence number of the last batch that used the set
    uint32_t pins = 0;                // staged batches holding the set
    uint32_t gen = 0;                 // bumped when the slot gets another set (stale key-cache entries then miss)
  };
  std::vector<LutMeta> lut_meta;
  // Front cache of lut_sets keyed by what DETERMINES a table set -- the frame's DHT payload bytes + scan selectors
  // (GpuScanPlan::key_bytes): a camera stream repeats them in every frame, and a hit means the frame's lookup tables
  // are never built on the host at all (jpeg_plan_gpu_scan(build_luts = false), a quarter of the planning work).
  struct LutKey {
    uint64_t hash = 0;
    std::vector<uint8_t> bytes;
    int set = -1;
    uint32_t gen = 0;
  };
  std::vector<LutKey> lut_keys;  // <= 2 * kMaxLutSets entries, replaced round-robin
  size_t lut_key_next = 0;
  uint64_t plan_seq = 0;
  static constexpr size_t kMaxTapSets = 32;  // resize-tap tables kept (one per distinct source size)
  std::map<std::pair<int, int>, uint64_t> taps_used;
  SyncLutImage* d_sync_luts = nullptr;  // same table sets, with the state-only step tables
  size_t blob_stride = 0;   // bytes reserved per frame for JPEG bytes
  size_t scans_off = 0, ivs_off = 0, stage_cap = 0;  // layout of the staging block (descriptors at 0)
  hipStream_t copy_stream = nullptr;
  uint32_t iv_cap = 0;      // restart intervals per batch
  bool stem_fusable = false;          // layer 0 can run as k_stem_planes_mfma
  bool rfb_tail = false;              // the three dilated RFB convs + the summed 1x1 run as ONE launch (k_rfb_tail)
  bool plan_parallel = false;         // UFD_PLAN_PARALLEL=1 at ufd_create: header scan + staging copy on the pool (A/B knob)
  bool gpu_entropy_enabled = true;   // device entropy kernels for baseline single-scan streams
  std::vector<float*> tap_buf;        // UFD_FLAG_TAP_LAYERS: per tensor, a copy taken right after its producing launch

  Slot slots[UFD_MAX_SLOTS];
  uint32_t next_ticket = 1;

  std::map<std::pair<int, int>, std::pair<TapsDev, TapsDev>> taps;

  // profiling
  bool profile = false;
  uint32_t prof_every = 1, prof_batch = 0;
  std::vector<std::string> prof_names;
  std::vector<ufd_kernel_stat> prof_stats;
  std::vector<ProfEntry> prof_pending;
  std::vector<hipEvent_t> prof_free;

  int fail(int code, const std::string& msg) {
    std::lock_guard<std::mutex> lk(err_mu);
    err = msg;
    return code;
  }
};

struct DevicePlan {
  bool ok = false;         // every decodable frame of the batch can take the device decoder
  bool any_ok = false;
  uint32_t n_iv = 0;       // intervals in h_ivs
  size_t used_blob = 0, used_coef = 0;  // bytes of JPEG data in the packed blob; largest coefficient slab
  size_t blob_base = 0;    // offset of the packed JPEG bytes inside the staging block
  size_t stage_bytes = 0;  // bytes of the staging block to copy
  uint32_t max_nsub = 0, max_bpm = 1;
};


// A batch whose JPEG bytes, frame descriptors and scan plans are resident in HBM
// (ufd_stage_jpeg_batch): submitting it moves no input over PCIe.
struct ufd_staged {
  uint32_t count = 0;
  DevicePlan plan;
  std::vector<JpegFrameDesc> h_descs;
  std::vector<HuffScan> h_scans;  // (which Huffman table sets the batch keeps pinned)
  std::vector<int32_t> st;
  uint8_t* d_stage = nullptr;  // device image of the staging block
  uint8_t* d_blob = nullptr;
  JpegFrameDesc* d_descs = nullptr;
  HuffScan* d_scans = nullptr;
  HuffInterval* d_ivs = nullptr;
};


namespace {

#define HIPC(m, expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return (m)->fail(UFD_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));               \
  } while (0)

// ---------------------------------------------------------------- events behind copies
// ROCm 7.2's runtime keeps ~2 KB of host memory for every event recorded DIRECTLY behind an asynchronous copy-engine
// transfer on a stream and never gives it back: the host-bytes path grew 2.1 KB per batch, 12 GB per hour at 54 k frames/s
// (tools/soak.py; reproduced on the runtime alone by tools/ubench/leak_probe2.hip: copy + hipEventRecord grows whether the
// event is waited for by a stream or by the host, copy + hipStreamSynchronize does not, and neither does copy + any kernel
// + hipEventRecord).  So an empty kernel goes between a copy and the event that marks it.
__global__ void k_copy_fence() {}
hipError_t record_behind_copy(hipEvent_t ev, hipStream_t stream) {
  hipLaunchKernelGGL(k_copy_fence, dim3(1), dim3(64), 0, stream);
  return hipEventRecord(ev, stream);
}

// ---------------------------------------------------------------- profiling
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

struct ProfScope {
  ufd_model* m;
  ProfEntry pe;
  bool on;
  hipStream_t st;
  ProfScope(ufd_model* mm, const std::string& name, double bytes, double flops, hipStream_t stream = nullptr)
      : m(mm), on(mm->profile && tl_prof), st(stream ? stream : tl_cur->stream) {
    tl_launches++;
    if (!on) return;
    {
      std::lock_guard<std::mutex> lk(m->shared_mu);
      pe.name_id = prof_name_id(m, name);
      pe.e0 = prof_event(m);
      pe.e1 = prof_event(m);
    }
    pe.bytes = bytes;
    pe.flops = flops;
    (void)hipEventRecord(pe.e0, st);
  }
  // the launch this scope was opened for did not happen: no sample, the events go back to the pool
  void cancel() {
    if (!on) return;
    on = false;
    std::lock_guard<std::mutex> lk(m->shared_mu);
    m->prof_free.push_back(pe.e0);
    m->prof_free.push_back(pe.e1);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(pe.e1, st);
    std::lock_guard<std::mutex> lk(m->shared_mu);
    m->prof_pending.push_back(pe);
  }
};

// Host-side sections of the pipeline (header scan, staging copies, launch enqueue): wall time on the
// issuing thread, reported beside the kernels as "host_*" entries (launches = batches).
struct HostScope {
  ufd_model* m;
  const char* name;
  bool on;
  std::chrono::steady_clock::time_point t0;
  HostScope(ufd_model* mm, const char* n) : m(mm), name(n), on(mm->profile && tl_prof) {
    if (on) t0 = std::chrono::steady_clock::now();
  }
  ~HostScope() {
    if (!on) return;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::lock_guard<std::mutex> lk(m->shared_mu);
    auto& st = m->prof_stats[prof_name_id(m, name)];
    st.launches++;
    st.total_ms += ms;
  }
};

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

// ---------------------------------------------------------------- model construction
void gen_priors(int W, int H, std::vector<float>& out) {
  // upstream generate_priors: float64 arithmetic, cast to f32, clamp to [0, 1]
  out.clear();
  for (int idx = 0; idx < 4; idx++) {
    int fw = (W + kStrides[idx] - 1) / kStrides[idx], fh = (H + kStrides[idx] - 1) / kStrides[idx];
    double shrink_w = (double)W / fw, shrink_h = (double)H / fh;
    double scale_w = (double)W / shrink_w, scale_h = (double)H / shrink_h;
    for (int j = 0; j < fh; j++)
      for (int i = 0; i < fw; i++) {
        double xc = (i + 0.5) / scale_w, yc = (j + 0.5) / scale_h;
        for (int a = 0; a < kHeadAnchors[idx]; a++) {
          double v[4] = {xc, yc, kMinBoxes[idx][a] / W, kMinBoxes[idx][a] / H};
          for (double x : v) {
            float f = (float)x;
            out.push_back(f < 0.f ? 0.f : (f > 1.f ? 1.f : f));
          }
        }
      }
  }
}

// Liveness-based arena: every conv output gets [B][c][h][w]; buffers are recycled after their
// last reader unless UFD_FLAG_KEEP_LAYERS asks to keep all of them for ufd_debug_layer_output.
void plan_tensors(ufd_model* m, bool keep_all) {
  const uint32_t flags = m->cfg.flags;
  const ConvSpec* specs = conv_specs();
  m->layers.resize(kNumConv);
  m->tensors.clear();
  std::vector<int> tensor_of(kNumConv, -1);
  int cat_tensor = -1;
  for (int i = 0; i < kNumConv; i++) {
    Layer& L = m->layers[i];
    L.spec = specs[i];
    if (L.spec.src == -1) {
      L.ih = m->H, L.iw = m->W;
      L.in_tensor = -1;
    } else if (L.spec.src == -2) {
      L.ih = m->layers[kRfbCatA].oh, L.iw = m->layers[kRfbCatA].ow;
      L.in_tensor = cat_tensor;
    } else {
      L.ih = m->layers[L.spec.src].oh, L.iw = m->layers[L.spec.src].ow;
      L.in_tensor = tensor_of[L.spec.src];
    }
    L.oh = conv_out_dim(L.ih, L.spec);
    L.ow = conv_out_dim(L.iw, L.spec);
    if (L.spec.k == 1 && L.spec.groups == 1)
      L.kind = kKindPointwise;
    else if (L.spec.k == 3 && L.spec.groups == 1 && L.spec.cout <= 16 && L.spec.pad == L.spec.dil)
      L.kind = kKindConv3x3;
    else
      L.kind = kKindDirect;
    L.res_tensor = (i == kRfbShortcut) ? tensor_of[kRfbLinear] : -1;
    L.out_coff = 0;
    if (i == kRfbCatA || i == kRfbCatB || i == kRfbCatC) {
      if (cat_tensor < 0) {
        Tensor t;
        t.c = 48, t.h = L.oh, t.w = L.ow;
        m->tensors.push_back(t);
        cat_tensor = (int)m->tensors.size() - 1;
      }
      L.out_tensor = cat_tensor;
      L.out_coff = (i == kRfbCatA) ? 0 : (i == kRfbCatB ? 16 : 32);
    } else {
      Tensor t;
      t.c = L.spec.cout, t.h = L.oh, t.w = L.ow;
      m->tensors.push_back(t);
      L.out_tensor = (int)m->tensors.size() - 1;
    }
    tensor_of[i] = L.out_tensor;
    const double in_b = (double)L.spec.cin * L.ih * L.iw * 4, out_b = (double)L.spec.cout * L.oh * L.ow * 4;
    L.weight_bytes = (double)(conv_weight_floats(L.spec) + L.spec.cout) * 4;
    L.bytes_per_frame = in_b + out_b + (L.res_tensor >= 0 ? out_b : 0);
    L.flops_per_frame = 2.0 * L.oh * L.ow * L.spec.cout * (L.spec.cin / L.spec.groups) * L.spec.k * L.spec.k;
  }
  // fuse every depthwise 3x3 into the pointwise conv that consumes it (its only consumer)
  for (int i = 0; i + 1 < kNumConv; i++) {
    Layer& D = m->layers[i];
    Layer& P = m->layers[i + 1];
    if (D.spec.groups == 1 || D.spec.groups != D.spec.cin || D.spec.k != 3 || D.spec.pad != 1 || D.spec.dil != 1 ||
        !D.spec.relu)
      continue;
    if (P.kind != kKindPointwise || P.spec.src != i || P.res_tensor >= 0) continue;
    ConvArgs probe{};
    probe.ih = D.ih, probe.iw = D.iw, probe.oh = P.oh, probe.ow = P.ow, probe.cin = D.spec.cin;
    if (!dwpw_supported(probe, D.spec.stride)) continue;
    P.kind = kKindDwPw;
    P.fused_dw = i;
    D.kind = kKindFusedAway;
    D.materialize = keep_all;
    P.bytes_per_frame = (double)D.spec.cin * D.ih * D.iw * 4 + (double)P.spec.cout * P.oh * P.ow * 4;
    P.flops_per_frame += D.flops_per_frame;
    P.weight_bytes += D.weight_bytes;
  }
  // chain two dw->pw blocks into one launch where the tensor between them is the big one
  // (m1 -> m2: 32 channels at half the input resolution) and nothing else reads it
  if (!keep_all && !(flags & UFD_FLAG_NO_CHAIN)) {
    for (int i = 0; i < kNumConv; i++) {
      Layer& P2 = m->layers[i];
      if (P2.kind != kKindDwPw) continue;
      const Layer& D2 = m->layers[P2.fused_dw];
      const int p1 = D2.spec.src;
      if (p1 < 0 || m->layers[p1].kind != kKindDwPw || m->layers[p1].chained) continue;
      Layer& P1 = m->layers[p1];
      const Layer& D1 = m->layers[P1.fused_dw];
      bool only_reader = true;
      for (int j = 0; j < kNumConv; j++)
        if (j != P2.fused_dw && m->layers[j].spec.src == p1) only_reader = false;
      for (int h = 0; h < 4; h++)
        if (kHeadCls[h] == p1 || kHeadReg[h] == p1) only_reader = false;
      if (!only_reader || D1.spec.stride != 1 || D2.spec.stride != 2) continue;
      ConvArgs f{}, g{};
      f.cin = P1.spec.cin, f.cout = P1.spec.cout, f.ih = D1.ih, f.iw = D1.iw, f.oh = P1.oh, f.ow = P1.ow, f.relu = P1.spec.relu;
      g.cin = P2.spec.cin, g.cout = P2.spec.cout, g.ih = D2.ih, g.iw = D2.iw, g.oh = P2.oh, g.ow = P2.ow;
      if (!dwpw2_supported(f, g)) continue;
      P2.kind = kKindDwPw2;
      P2.chain_first = p1;
      P1.chained = true;
      P2.bytes_per_frame = (double)D1.spec.cin * D1.ih * D1.iw * 4 + (double)P2.spec.cout * P2.oh * P2.ow * 4;
      P2.flops_per_frame += P1.flops_per_frame;
      P2.weight_bytes += P1.weight_bytes;
    }
  }
  // out = relu(ConvLinear(cat) + shortcut(x)) as ONE 1x1 conv over the channels of both inputs
  // (weights side by side, biases summed): ConvLinear's output, written once and read back as the
  // residual, never exists.  fp32 rounding apart from the two-launch form (one fma chain, not two).
  if (!keep_all && !(flags & UFD_FLAG_NO_RFB_SUM)) {
    Layer& S = m->layers[kRfbShortcut];
    Layer& Lin = m->layers[kRfbLinear];
    if (S.kind == kKindPointwise && Lin.kind == kKindPointwise && S.res_tensor == tensor_of[kRfbLinear] && S.oh == Lin.oh &&
        S.ow == Lin.ow && S.spec.cout == Lin.spec.cout && (Lin.spec.cin & 1) == 0 && (S.spec.cin & 1) == 0) {
      bool only_reader = true;
      for (int j = 0; j < kNumConv; j++)
        if (j != kRfbShortcut && m->layers[j].spec.src == kRfbLinear) only_reader = false;
      if (only_reader) {
        S.sum_with = kRfbLinear;
        S.res_tensor = -1;
        Lin.chained = true;
        S.bytes_per_frame = ((double)Lin.spec.cin + S.spec.cin + S.spec.cout) * S.oh * S.ow * 4;
        S.flops_per_frame += Lin.flops_per_frame;
        S.weight_bytes += Lin.weight_bytes;
      }
    }
  }
  // RFB tail as ONE launch (k_rfb_tail, issued at the shortcut layer's turn): the three dilated 3x3 convs hand their
  // results to the summed 1x1 in registers, the 48-channel concat tensor never exists.  The dilated layers become
  // "chained" (no launch, no output of their own; ufd_debug_layer_output reports them absent in this plan).
  m->rfb_tail = false;
  if (!keep_all && !(flags & UFD_FLAG_NO_RFB_TAIL) && m->layers[kRfbShortcut].sum_with == kRfbLinear) {
    const int dil_layers[3] = {kRfbCatA, kRfbCatB, kRfbCatC};
    ConvArgs d3[3]{}, fin{};
    bool ok = true;
    for (int b = 0; b < 3; b++) {
      const Layer& D = m->layers[dil_layers[b]];
      ok = ok && D.kind == kKindConv3x3 && !D.chained;
      d3[b].k = D.spec.k, d3[b].stride = D.spec.stride, d3[b].dil = D.spec.dil, d3[b].pad = D.spec.pad;
      d3[b].cin = D.spec.cin, d3[b].cout = D.spec.cout, d3[b].relu = D.spec.relu;
      d3[b].ih = D.ih, d3[b].iw = D.iw, d3[b].oh = D.oh, d3[b].ow = D.ow;
    }
    const Layer& S = m->layers[kRfbShortcut];
    fin.k = 1, fin.cout = S.spec.cout, fin.cin = m->layers[kRfbLinear].spec.cin + S.spec.cin;
    fin.in2_ctotal = S.in_tensor >= 0 ? m->tensors[S.in_tensor].c : 0;
    fin.oh = S.oh, fin.ow = S.ow;
    if (ok && rfb_tail_supported(d3, fin)) {
      m->rfb_tail = true;
      Layer& S2 = m->layers[kRfbShortcut];
      for (int b = 0; b < 3; b++) {
        Layer& D = m->layers[dil_layers[b]];
        D.chained = true;
        S2.flops_per_frame += D.flops_per_frame;
        S2.weight_bytes += D.weight_bytes;
      }
    }
  }
  // The three RFB reduce convs (64 -> 8 each, same input) as ONE 64 -> 24 conv: one cout tile instead
  // of three, the input read once; the consumers read channel slices of the stacked tensor.
  if (!keep_all) {
    static const int kStack[3] = {13, 16, 19};
    Layer& A = m->layers[kStack[0]];
    int cout_sum = 0;
    bool ok = true;
    for (int k = 0; k < 3; k++) {
      const Layer& Bm = m->layers[kStack[k]];
      ok = ok && Bm.kind == kKindPointwise && Bm.in_tensor == A.in_tensor && Bm.spec.cin == A.spec.cin && Bm.oh == A.oh &&
           Bm.ow == A.ow && Bm.res_tensor < 0 && Bm.spec.relu == A.spec.relu && Bm.out_coff == 0 && !Bm.chained && Bm.sum_with < 0;
      for (int h = 0; h < 4; h++) ok = ok && kHeadCls[h] != kStack[k] && kHeadReg[h] != kStack[k];
      cout_sum += Bm.spec.cout;
    }
    if (ok && cout_sum <= 32) {
      Tensor t;
      t.c = cout_sum, t.h = A.oh, t.w = A.ow;
      m->tensors.push_back(t);
      const int stacked = (int)m->tensors.size() - 1;
      int coff = 0;
      for (int k = 0; k < 3; k++) {
        Layer& Bm = m->layers[kStack[k]];
        for (int j = 0; j < kNumConv; j++)
          if (m->layers[j].in_tensor == Bm.out_tensor && m->layers[j].spec.src == kStack[k]) m->layers[j].in_tensor = stacked, m->layers[j].in_coff = coff;
        Bm.tap_tensor = stacked, Bm.tap_coff = coff;
        coff += Bm.spec.cout;
        A.stack[k] = kStack[k];
        if (k > 0) {
          Bm.chained = true;
          A.bytes_per_frame += (double)Bm.spec.cout * Bm.oh * Bm.ow * 4;
          A.flops_per_frame += Bm.flops_per_frame;
          A.weight_bytes += Bm.weight_bytes;
        }
      }
      A.out_tensor = stacked;
    }
  }
  // merged launches: layers with identical shapes whose inputs are ready at the leader's turn
  for (int i = 0; i < kNumConv; i++) m->layers[i].leader = i, m->layers[i].group[0] = i, m->layers[i].group[1] = m->layers[i].group[2] = -1;
  {
    // (leader first: the launch is issued at the leader's turn, so a leader that is not the lowest
    // index -- the RFB's dilated convs wait for the b2 branch -- delays the others to its turn)
    static const int kGroups[][3] = {{13, 16, 19}, {14, 17, 20}, {22, 15, 18}, {26, 28, -1}, {36, 38, -1}, {44, 46, -1}, {50, 51, -1}};
    for (const auto& g : kGroups) {
      const Layer& A = m->layers[g[0]];
      bool ok = !A.chained && A.stack[0] < 0;
      for (int k = 1; k < 3 && g[k] >= 0; k++) {
        ok = ok && !m->layers[g[k]].chained;
        const Layer& Bm = m->layers[g[k]];
        ok = ok && Bm.kind == A.kind && Bm.spec.cin == A.spec.cin && Bm.ih == A.ih && Bm.iw == A.iw && Bm.oh == A.oh &&
             Bm.ow == A.ow && Bm.spec.k == A.spec.k && Bm.spec.stride == A.spec.stride &&
             (Bm.spec.cout + 31) / 32 == (A.spec.cout + 31) / 32 && Bm.res_tensor < 0 && A.res_tensor < 0;
        // dense 3x3 convs of stride 1 may differ in dilation (run-time dilation form of the row kernel)
        const bool dil_free = A.kind == kKindConv3x3 && A.spec.stride == 1 && A.ow % 4 == 0 && Bm.spec.pad == Bm.spec.dil &&
                              A.spec.pad == A.spec.dil && Bm.spec.dil <= 5 && A.spec.dil <= 5 && !keep_all;
        ok = ok && (Bm.spec.dil == A.spec.dil || dil_free);
        // one launch at the leader's turn: every member reads the same, already produced tensor, or
        // a tensor whose producing launch comes before that turn
        const int src_a = A.kind == kKindDwPw ? m->layers[A.fused_dw].in_tensor : A.in_tensor;
        const int src_b = Bm.kind == kKindDwPw ? m->layers[Bm.fused_dw].in_tensor : Bm.in_tensor;
        const int prod_b = Bm.spec.src;
        const bool produced_before = Bm.kind == kKindConv3x3 && prod_b >= 0 && prod_b < g[0] && m->layers[prod_b].leader < g[0] &&
                                     g[k] < g[0];
        ok = ok && (src_a == src_b || produced_before);
        if (A.kind == kKindDwPw)
          ok = ok && m->layers[Bm.fused_dw].spec.stride == m->layers[A.fused_dw].spec.stride &&
               m->layers[Bm.fused_dw].ih == m->layers[A.fused_dw].ih;
        if (A.kind == kKindConv3x3) ok = ok && Bm.spec.cout <= 16 && A.spec.cout <= 16;  // (row kernel or gather kernel: one cout tile)
      }
      if (!ok) continue;
      for (int k = 0; k < 3 && g[k] >= 0; k++) {
        m->layers[g[k]].leader = g[0];
        m->layers[g[0]].group[k] = g[k];
      }
    }
  }
  // Dual launches: a cls/reg head pair and the next backbone block both read the tensor produced just before them and
  // do not depend on each other -- one grid for both (k_dual_*), issued at the head pair's turn.
  if (!keep_all && !(flags & UFD_FLAG_NO_DUAL)) {
    static const int kDuals[][2] = {{kHeadCls[0], 30}, {kHeadCls[1], 40}, {kHeadCls[2], 47}};
    for (const auto& d : kDuals) {
      Layer& A = m->layers[d[0]];
      Layer& Bm = m->layers[d[1]];
      if (A.kind != kKindDwPw || A.leader != d[0] || A.chained) continue;
      if ((Bm.kind != kKindDwPw && Bm.kind != kKindPointwise) || Bm.leader != d[1] || Bm.group[1] >= 0 || Bm.chained ||
          Bm.stack[0] >= 0 || Bm.sum_with >= 0 || Bm.res_tensor >= 0 || d[1] <= d[0])
        continue;
      // everything B reads exists before A's turn
      const int src = Bm.kind == kKindDwPw ? m->layers[Bm.fused_dw].spec.src : Bm.spec.src;
      if (src < 0 || src >= d[0] || m->layers[src].chained) continue;
      const int src_turn = m->layers[src].ride >= 0 ? m->layers[src].ride : m->layers[src].leader;
      if (src_turn >= d[0]) continue;
      A.rider = d[1];
      Bm.ride = d[0];
    }
  }
  // where each layer's output can be read back in this plan (ufd_debug_layer_output)
  for (int i = 0; i < kNumConv; i++) {
    Layer& L = m->layers[i];
    if (L.tap_tensor >= 0) continue;  // slice of the stacked reduce tensor
    if ((L.kind == kKindFusedAway && !L.materialize) || L.chained) continue;
    L.tap_tensor = L.out_tensor, L.tap_coff = L.out_coff;
  }
  // liveness: first writer, last reader (head outputs live until the decode kernel)
  const int nt = (int)m->tensors.size();
  std::vector<int> first(nt, kNumConv), last(nt, -1);
  for (int i = 0; i < kNumConv; i++) {
    const Layer& L = m->layers[i];
    if (L.kind == kKindFusedAway && !L.materialize) continue;  // never written, never read
    if (L.chained) continue;                                     // computed inside a later launch
    const int lead = m->layers[L.leader].ride >= 0 ? std::min(L.leader, m->layers[L.leader].ride) : L.leader;
    first[L.out_tensor] = std::min(first[L.out_tensor], lead);  // a merged layer writes at its leader's turn, a rider at its host's
    last[L.out_tensor] = std::max(last[L.out_tensor], std::max(i, L.leader));
    const int when = std::max(i, L.leader);  // a merged member is read at its leader's turn
    int src_t = L.kind == kKindDwPw ? m->layers[L.fused_dw].in_tensor : L.in_tensor;
    if (L.kind == kKindDwPw2) src_t = m->layers[m->layers[L.chain_first].fused_dw].in_tensor;
    if (src_t >= 0) last[src_t] = std::max(last[src_t], when);
    if (L.in_tensor >= 0 && L.kind != kKindDwPw && L.kind != kKindDwPw2) last[L.in_tensor] = std::max(last[L.in_tensor], when);
    if (L.res_tensor >= 0) last[L.res_tensor] = std::max(last[L.res_tensor], when);
    if (L.sum_with >= 0 && m->layers[L.sum_with].in_tensor >= 0)
      last[m->layers[L.sum_with].in_tensor] = std::max(last[m->layers[L.sum_with].in_tensor], when);
  }
  if (m->rfb_tail)  // the fused launch reads the dilated convs' inputs at the shortcut layer's turn
    for (int j : {kRfbCatA, kRfbCatB, kRfbCatC})
      if (m->layers[j].in_tensor >= 0) last[m->layers[j].in_tensor] = std::max(last[m->layers[j].in_tensor], kRfbShortcut);
  for (int h = 0; h < 4; h++) {
    last[tensor_of[kHeadCls[h]]] = kNumConv;
    last[tensor_of[kHeadReg[h]]] = kNumConv;
  }
  struct Blk {
    size_t off, size;
  };
  std::vector<Blk> free_list;
  size_t top = 0;
  auto align = [](size_t v) { return (v + 63) & ~(size_t)63; };
  auto allocate = [&](int t) {
    const size_t need = align(m->tensors[t].per_frame() * m->B);
    size_t best = (size_t)-1;
    for (size_t i = 0; i < free_list.size(); i++)
      if (free_list[i].size >= need && (best == (size_t)-1 || free_list[i].size < free_list[best].size)) best = i;
    if (best != (size_t)-1) {
      m->tensors[t].off = free_list[best].off;
      if (free_list[best].size > need) {
        free_list[best].off += need;
        free_list[best].size -= need;
      } else {
        free_list.erase(free_list.begin() + best);
      }
    } else {
      m->tensors[t].off = top;
      top += need;
    }
  };
  std::vector<bool> allocated(nt, false);
  for (int i = 0; i < kNumConv; i++) {
    for (int j = i; j < kNumConv; j++) {  // every tensor first written at turn i (merged members included)
      const int tj = m->layers[j].out_tensor;
      if (first[tj] == i && !allocated[tj]) {
        allocate(tj);
        allocated[tj] = true;
      }
    }  // (a fused-away depthwise output has first == kNumConv: no storage)
    // a buffer is recycled only after the layer that reads it last has been issued, so a
    // layer's output never aliases its own inputs
    if (!keep_all)
      for (int u = 0; u < nt; u++)
        if (last[u] == i && allocated[u])  // (a tensor no launch writes -- the RFB concat under k_rfb_tail -- has no storage to give back)
          free_list.push_back({m->tensors[u].off, align(m->tensors[u].per_frame() * m->B)});
  }
  for (int u = 0; u < nt; u++) m->tensors[u].first = first[u], m->tensors[u].last = last[u], m->tensors[u].stored = allocated[u];
  m->arena_floats = top;
}

int upload_weights(ufd_model* m, const float* blob) {
  const ConvSpec* specs = conv_specs();
  std::vector<float> img;
  std::vector<size_t> w_off(kNumConv), b_off(kNumConv), dw_off(kNumConv, (size_t)-1), rows_off(kNumConv, (size_t)-1);
  const float* p = blob;
  for (int i = 0; i < kNumConv; i++) {
    const ConvSpec& s = specs[i];
    const size_t nw = conv_weight_floats(s);
    while (img.size() % 64) img.push_back(0.f);
    w_off[i] = img.size();
    const LayerKind kind = m->layers[i].kind;
    if (kind == kKindPointwise || kind == kKindDwPw || kind == kKindDwPw2) {
      const size_t np = pointwise_packed_floats(s.cin, s.cout);
      img.resize(img.size() + np);
      pack_pointwise_weights(p, s.cin, s.cout, img.data() + w_off[i]);
    } else if (kind == kKindConv3x3) {
      const size_t np = conv3x3_packed_floats(s.cin);
      img.resize(img.size() + np);
      pack_conv3x3_weights(p, s.cin, s.cout, img.data() + w_off[i]);
    } else {
      img.insert(img.end(), p, p + nw);
    }
    p += nw;
    while (img.size() % 64) img.push_back(0.f);
    b_off[i] = img.size();
    img.insert(img.end(), p, p + s.cout);
    if (kind == kKindConv3x3) {
      while (img.size() % 64) img.push_back(0.f);
      rows_off[i] = img.size();
      img.resize(img.size() + conv3x3_rows_packed_floats(s.cin));
      pack_conv3x3_rows_weights(p - nw, s.cin, s.cout, img.data() + rows_off[i]);
    }
    if (s.groups > 1 && s.k == 3) {  // depthwise: also the [c][12] image the fused kernel copies into LDS
      while (img.size() % 64) img.push_back(0.f);
      dw_off[i] = img.size();
      img.resize(img.size() + depthwise_packed_floats(s.cout));
      pack_depthwise_weights(p - nw, p, s.cout, img.data() + dw_off[i]);
    }
    p += s.cout;
  }
  // summed 1x1 pairs: weights of both convs side by side per output channel, biases added
  std::vector<size_t> sumw_off(kNumConv, (size_t)-1), sumb_off(kNumConv, (size_t)-1);
  {
    std::vector<const float*> wsrc(kNumConv), bsrc(kNumConv);
    const float* q = blob;
    for (int i = 0; i < kNumConv; i++) {
      wsrc[i] = q;
      q += conv_weight_floats(specs[i]);
      bsrc[i] = q;
      q += specs[i].cout;
    }
    for (int i = 0; i < kNumConv; i++) {
      if (m->layers[i].stack[0] != i) continue;
      const int ci = specs[i].cin;
      std::vector<float> wcat, bcat;
      for (int k : m->layers[i].stack) {
        if (k < 0) continue;
        wcat.insert(wcat.end(), wsrc[k], wsrc[k] + (size_t)specs[k].cout * ci);
        bcat.insert(bcat.end(), bsrc[k], bsrc[k] + specs[k].cout);
      }
      while (img.size() % 64) img.push_back(0.f);
      sumw_off[i] = img.size();
      img.resize(img.size() + pointwise_packed_floats(ci, (int)bcat.size()));
      pack_pointwise_weights(wcat.data(), ci, (int)bcat.size(), img.data() + sumw_off[i]);
      while (img.size() % 64) img.push_back(0.f);
      sumb_off[i] = img.size();
      img.insert(img.end(), bcat.begin(), bcat.end());
    }
    for (int i = 0; i < kNumConv; i++) {
      const int j = m->layers[i].sum_with;
      if (j < 0) continue;
      const int ca = specs[j].cin, cb = specs[i].cin, co = specs[i].cout;
      std::vector<float> wcat((size_t)co * (ca + cb)), bsum(co);
      for (int o = 0; o < co; o++) {
        std::memcpy(&wcat[(size_t)o * (ca + cb)], wsrc[j] + (size_t)o * ca, sizeof(float) * ca);
        std::memcpy(&wcat[(size_t)o * (ca + cb) + ca], wsrc[i] + (size_t)o * cb, sizeof(float) * cb);
        bsum[o] = bsrc[j][o] + bsrc[i][o];
      }
      while (img.size() % 64) img.push_back(0.f);
      sumw_off[i] = img.size();
      img.resize(img.size() + pointwise_packed_floats(ca + cb, co));
      pack_pointwise_weights(wcat.data(), ca + cb, co, img.data() + sumw_off[i]);
      while (img.size() % 64) img.push_back(0.f);
      sumb_off[i] = img.size();
      img.insert(img.end(), bsum.begin(), bsum.end());
    }
  }
  size_t tail_off = (size_t)-1;
  if (m->rfb_tail) {
    const float* q = blob;
    const float *w_lin = nullptr, *w_short = nullptr;
    for (int i = 0; i < kNumConv; i++) {
      if (i == kRfbLinear) w_lin = q;
      if (i == kRfbShortcut) w_short = q;
      q += conv_weight_floats(specs[i]) + specs[i].cout;
    }
    while (img.size() % 64) img.push_back(0.f);
    tail_off = img.size();
    img.resize(img.size() + rfb_tail_packed_floats());
    pack_rfb_tail_weights(w_lin, w_short, img.data() + tail_off);
  }
  m->weight_img_floats = img.size();
  HIPC(m, hipMalloc(&m->d_weights, img.size() * sizeof(float)));
  HIPC(m, hipMemcpy(m->d_weights, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
  for (int i = 0; i < kNumConv; i++) {
    m->layers[i].d_w = m->d_weights + w_off[i];
    m->layers[i].d_b = m->d_weights + b_off[i];
    if (dw_off[i] != (size_t)-1) m->layers[i].d_w_dwpack = m->d_weights + dw_off[i];
    if (rows_off[i] != (size_t)-1) m->layers[i].d_w_rows = m->d_weights + rows_off[i];
    if (sumw_off[i] != (size_t)-1) m->layers[i].d_w_sum = m->d_weights + sumw_off[i], m->layers[i].d_b_sum = m->d_weights + sumb_off[i];
  }
  if (tail_off != (size_t)-1) m->layers[kRfbShortcut].d_w_tail = m->d_weights + tail_off;
  return UFD_OK;
}

int alloc_slot(ufd_model* m, Slot& s) {
  if (s.h_descs) return UFD_OK;
  HIPC(m, hipHostMalloc(&s.h_stage, m->stage_cap, hipHostMallocDefault));
  s.h_descs = reinterpret_cast<JpegFrameDesc*>(s.h_stage);
  s.h_scans = reinterpret_cast<HuffScan*>(s.h_stage + m->scans_off);
  s.h_ivs = reinterpret_cast<HuffInterval*>(s.h_stage + m->ivs_off);
  HIPC(m, hipHostMalloc(&s.h_dets, sizeof(Det) * kDetCopy * m->B, hipHostMallocDefault));
  HIPC(m, hipMalloc(&s.d_dets, sizeof(Det) * m->K * m->B));
  HIPC(m, hipMemset(s.d_dets, 0, sizeof(Det) * m->K * m->B));
  // (decode status and detection counts side by side, as on the device: ONE copy brings both back)
  HIPC(m, hipHostMalloc(&s.h_gpu_status, sizeof(uint32_t) * 2 * m->B, hipHostMallocDefault));
  s.h_ndet = s.h_gpu_status + m->B;
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
float* tensor_ptr(ufd_model* m, int t) { return tl_cur->d_arena + m->tensors[t].off; }

// one conv layer for frames [f0, f0 + count) of the batch
// Kernel arguments of conv layer i for frames [f0, f0 + count)
ConvArgs layer_args(ufd_model* m, int i, uint32_t f0, uint32_t count, int* dw_stride) {
  const Layer& L = m->layers[i];
  auto in_ptr = [&](int t, int ih, int iw) -> const float* {
    if (t < 0) return tl_cur->d_input + (size_t)f0 * 3 * ih * iw;
    return tensor_ptr(m, t) + (size_t)f0 * m->tensors[t].per_frame();
  };
  ConvArgs a{};
  a.in = in_ptr(L.in_tensor, L.ih, L.iw);
  a.w = L.d_w;
  a.bias = L.d_b;
  a.out = L.chained ? nullptr : tensor_ptr(m, L.out_tensor) + (size_t)f0 * m->tensors[L.out_tensor].per_frame();
  a.res = L.res_tensor >= 0 ? in_ptr(L.res_tensor, 0, 0) : nullptr;
  a.B = (int)count;
  a.cin = L.spec.cin, a.cout = L.spec.cout;
  a.ih = L.ih, a.iw = L.iw, a.oh = L.oh, a.ow = L.ow;
  a.k = L.spec.k, a.stride = L.spec.stride, a.pad = L.spec.pad, a.dil = L.spec.dil;
  a.depthwise = L.spec.groups > 1;
  a.relu = L.spec.relu || i == kRfbShortcut;
  a.in_ctotal = L.in_tensor < 0 ? 3 : m->tensors[L.in_tensor].c;
  a.out_ctotal = m->tensors[L.out_tensor].c;
  a.out_coff = L.out_coff;
  *dw_stride = 1;
  if (L.in_coff) a.in += (size_t)L.in_coff * L.ih * L.iw;  // a channel slice of a stacked tensor
  if (L.stack[0] == i) {  // this launch computes the stacked output channels of all members
    a.cout = m->tensors[L.out_tensor].c;
    a.w = L.d_w_sum;
    a.bias = L.d_b_sum;
  }
  if (L.sum_with >= 0) {  // two summed 1x1 convs: first the other conv's input channels, then this layer's
    const Layer& O = m->layers[L.sum_with];
    a.in2 = a.in;
    a.in2_ctotal = a.in_ctotal;
    a.in = in_ptr(O.in_tensor, O.ih, O.iw);
    a.in_ctotal = O.in_tensor < 0 ? 3 : m->tensors[O.in_tensor].c;
    a.ksplit = O.spec.cin >> 1;
    a.cin = O.spec.cin + L.spec.cin;
    a.w = L.d_w_sum;
    a.bias = L.d_b_sum;
  }
  if (L.kind == kKindDwPw2) {  // second block of a chain: its input tensor does not exist
    const Layer& D = m->layers[L.fused_dw];
    a.in = nullptr;
    a.ih = D.ih, a.iw = D.iw;
    a.w2 = D.d_w_dwpack, a.bias2 = D.d_b;
    *dw_stride = D.spec.stride;
  }
  if (L.kind == kKindDwPw) {
    const Layer& D = m->layers[L.fused_dw];
    a.in = in_ptr(D.in_tensor, D.ih, D.iw);
    a.in_ctotal = D.in_tensor < 0 ? 3 : m->tensors[D.in_tensor].c;
    a.ih = D.ih, a.iw = D.iw;
    a.w2 = D.d_w_dwpack, a.bias2 = D.d_b;
    *dw_stride = D.spec.stride;
  }
  return a;
}

// Issues conv layer i -- together with the layers merged into its launch (Layer::group: cls + reg
// head pairs and the three RFB reduce convs share shapes and run as one launch, blockIdx.y
// selecting the member).  Non-leading members are skipped when their turn comes.
void tap_outputs(ufd_model* m, int i, uint32_t count, hipStream_t st);

void enqueue_layer_launch(ufd_model* m, int i, uint32_t f0, uint32_t count, hipStream_t st);

void enqueue_layer(ufd_model* m, int i, uint32_t count) {
  hipStream_t st = tl_cur->stream;
  enqueue_layer_launch(m, i, 0, count, st);
  if (!m->tap_buf.empty() && tl_cur == &m->ctx[0]) tap_outputs(m, i, count, st);
}

// UFD_FLAG_TAP_LAYERS: copies every tensor the launch issued at layer i's turn has just written
// (whole batch) to its tap buffer, before the arena recycles it.
void tap_outputs(ufd_model* m, int i, uint32_t count, hipStream_t st) {
  const Layer& L = m->layers[i];
  if ((L.kind == kKindFusedAway && !L.materialize) || L.chained || L.leader != i) return;
  int seen[3] = {-1, -1, -1}, n = 0;
  for (int j : L.group) {
    if (j < 0) continue;
    const int t = m->layers[j].out_tensor;
    if (t == seen[0] || t == seen[1]) continue;
    seen[n++] = t;
    (void)hipMemcpyAsync(m->tap_buf[t], tensor_ptr(m, t), sizeof(float) * m->tensors[t].per_frame() * count, hipMemcpyDeviceToDevice, st);
  }
}

void enqueue_layer_launch(ufd_model* m, int i, uint32_t f0, uint32_t count, hipStream_t st) {
  const Layer& L = m->layers[i];
  if (L.kind == kKindFusedAway && !L.materialize) return;
  if (L.chained) return;      // computed inside the kKindDwPw2 launch of the next block
  if (L.leader != i) return;  // issued with its group leader
  if (L.ride >= 0 && !tl_force_rider) return;  // issued in (or right behind) the launch of the layer it rides with
  if (i == 0 && tl_cur->stem_descs) {  // stem conv straight from the decoder's sample planes
    int st_ = 1;
    StemArgs sa;
    sa.a = layer_args(m, 0, f0, count, &st_);
    sa.a.w = L.d_w_rows;
    sa.descs = tl_cur->stem_descs + f0;
    sa.planes = tl_cur->d_planes + (size_t)f0 * m->plane_stride;
    sa.plane_stride = m->plane_stride;
    sa.lut = m->d_lut;
    ProfScope ps(m, std::string("stem_planes_mfma:") + L.spec.name,
                 (double)count * (1.5 * L.ih * L.iw + 4.0 * L.spec.cout * L.oh * L.ow) + L.weight_bytes, L.flops_per_frame * count, st);
    launch_stem_planes_mfma(sa, st);
    return;
  }
  if (i == kRfbShortcut && m->rfb_tail) {  // the three dilated convs + relu(ConvLinear(cat) + shortcut(x)) as one launch
    int st_ = 1;
    ConvArgs d3[3];
    const int dil_layers[3] = {kRfbCatA, kRfbCatB, kRfbCatC};
    std::string names;
    for (int b = 0; b < 3; b++) {
      d3[b] = layer_args(m, dil_layers[b], f0, count, &st_);
      d3[b].w = m->layers[dil_layers[b]].d_w_rows;
      names += std::string(m->layers[dil_layers[b]].spec.name) + "+";
    }
    ConvArgs fin = layer_args(m, i, f0, count, &st_);
    fin.w = L.d_w_tail;
    ProfScope ps(m, std::string("rfb_tail:") + names + L.spec.name,
                 L.bytes_per_frame * count + L.weight_bytes, L.flops_per_frame * count, st);
    launch_rfb_tail(d3, fin, st);
    return;
  }
  if (L.kind == kKindDwPw2) {
    int s1 = 1, s2 = 2;
    const Layer& F = m->layers[L.chain_first];
    const ConvArgs first = layer_args(m, L.chain_first, f0, count, &s1);
    const ConvArgs second = layer_args(m, i, f0, count, &s2);
    ProfScope ps(m, std::string("conv_dwpw2_mfma") + conv_dwpw2_instance(first, second) + ":" + F.spec.name + "+" + L.spec.name,
                 L.bytes_per_frame * count + L.weight_bytes,
                 L.flops_per_frame * count, st);
    launch_conv_dwpw2_mfma(first, second, st);
    return;
  }
  ConvArgs args[3];
  int n = 0, dw_stride = 1;
  std::string names;
  double bytes = 0, flops = 0;
  for (int j : L.group) {
    if (j < 0) continue;
    const Layer& M = m->layers[j];
    args[n++] = layer_args(m, j, f0, count, &dw_stride);
    names += (names.empty() ? "" : "+") + std::string(M.spec.name);
    bytes += M.bytes_per_frame * count + M.weight_bytes;
    flops += M.flops_per_frame * count;
  }
  ConvArgs& a = args[0];
  bool use_rows = false;
  const char* kind = "conv_direct_full";
  switch (L.kind) {
    case kKindPointwise: kind = "conv_pw_mfma"; break;
    case kKindDwPw: kind = dwpw_uses_coop(args, n) ? "conv_dwpw_coop" : "conv_dwpw_mfma"; break;
    case kKindDwPw2: break;  // issued above
    case kKindConv3x3:
      use_rows = true;
      for (int j = 0; j < n; j++) use_rows = use_rows && conv3x3_rows_supported(args[j]);
      kind = use_rows ? "conv3x3_rows_mfma" : "conv3x3_mfma";
      break;
    case kKindFusedAway: kind = "conv_direct_dw_debug"; break;
    case kKindDirect: kind = a.depthwise ? "conv_direct_dw" : "conv_direct_full"; break;
  }
  if (L.rider >= 0 && L.kind == kKindDwPw) {  // dual launch with the rider's conv, when that pair of instances exists
    const Layer& R = m->layers[L.rider];
    int r_stride = 1;
    const ConvArgs rb = layer_args(m, L.rider, f0, count, &r_stride);
    const int rb_stride = R.kind == kKindDwPw ? r_stride : 0;
    // (labelled with the device function and its template instance, like every other launch: "conv_dual_coop<1, 1, 2>")
    if (const char* label = conv_dual_instance(args, n, dw_stride, &rb, rb_stride)) {
      ProfScope ps(m, std::string(label) + ":" + names + "|" + R.spec.name, bytes + R.bytes_per_frame * count + R.weight_bytes,
                   flops + R.flops_per_frame * count, st);
      if (launch_conv_dual(args, n, dw_stride, &rb, rb_stride, st)) return;
      ps.cancel();
    }
  }
  const char* inst = L.kind == kKindPointwise ? conv_pointwise_instance(args, n)
                     : (L.kind == kKindDwPw ? conv_dwpw_instance(args, n, dw_stride) : (use_rows ? conv3x3_rows_instance(args, n) : ""));
  {
  ProfScope ps(m, std::string(kind) + inst + ":" + names, bytes, flops, st);
  switch (L.kind) {
    case kKindPointwise: launch_conv_pointwise_mfma(args, n, st); break;
    case kKindDwPw: launch_conv_dwpw_mfma(args, n, dw_stride, st); break;
    case kKindConv3x3:
      if (use_rows) {
        int k = 0;
        for (int j : L.group)
          if (j >= 0) args[k++].w = m->layers[j].d_w_rows;
        launch_conv3x3_rows_mfma(args, n, st);
      } else {
        launch_conv3x3_mfma(args, n, st);
      }
      break;
    default: launch_conv_direct(a, st); break;
  }
  }
  if (L.rider >= 0) {  // the pair is not compiled as one grid: the rider right behind its host, on its own
    tl_force_rider = true;
    enqueue_layer_launch(m, L.rider, f0, count, st);
    tl_force_rider = false;
  }
}

// [count][3][H][W] in d_input (or the sample planes, fused stem) -> every conv output the heads need.
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

void enqueue_forward(ufd_model* m, uint32_t count) {
  for (int i = 0; i < kNumConv; i++) enqueue_layer(m, i, count);
  tl_cur->last_forward_count = count;
}

void enqueue_heads(ufd_model* m, uint32_t count, bool raw_outputs = false) {
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
  launch_sort_nms(tl_cur->d_keys, m->key_stride, tl_cur->d_counts, tl_cur->d_boxes, m->K, m->cfg.max_iou, s.d_dets, m->K, tl_cur->d_ndet,
                  tl_cur->d_spill, tl_cur->d_nms_mat, count, tl_cur->stream);
}

int enqueue_results_copy(ufd_model* m, Slot& s, uint32_t count) {
  // [B decode statuses][B detection counts] are one allocation on both sides: one copy of B + count words (statuses past
  // `count` are stale and never read), or the counts alone when the host decoded the entropy stage
  if (s.gpu_entropy)
    HIPC(m, hipMemcpyAsync(s.h_gpu_status, tl_cur->d_status, sizeof(uint32_t) * ((size_t)m->B + count), hipMemcpyDeviceToHost, tl_cur->stream));
  else
    HIPC(m, hipMemcpyAsync(s.h_ndet, tl_cur->d_ndet, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, tl_cur->stream));
  HIPC(m, hipMemcpy2DAsync(s.h_dets, sizeof(Det) * kDetCopy, s.d_dets, sizeof(Det) * m->K, sizeof(Det) * kDetCopy, count,
                           hipMemcpyDeviceToHost, tl_cur->stream));
  span_end(s);
  HIPC(m, hipEventRecord(s.done, tl_cur->stream));
  s.ctx = tl_cur;
  return UFD_OK;
}

// N1: the finished streams of the slot's batch -> the caller's buffer.  Their sizes are known only now, so this is the
// second half of a two-step copy: one D2H of everything that fits, on the handle's copy stream.
int fetch_streams(ufd_model* m, Slot& s) {
  const ufd_annotate& a = s.annot_args;
  for (uint32_t i = 0; i < s.count; i++) a.jpeg_off[i] = 0, a.jpeg_len[i] = 0;
  if (!s.annot_ran) return UFD_OK;  // nothing decoded
  const uint32_t* len = s.h_enc_meta;
  const uint32_t* off = s.h_enc_meta + m->B;
  size_t fit = 0;  // bytes of the packed output that hold whole streams and fit the caller's buffer
  int rc = UFD_OK;
  for (uint32_t i = 0; i < s.count; i++) {
    // (the frame's final status as finish_slot merged it -- host parse, device entropy decoder, truncation -- whether or
    // not the caller passed a status array: a frame the device decoder flagged never hands out its stream)
    const bool failed = s.st[i] != UFD_OK && s.st[i] != UFD_E_TRUNCATED;
    if (failed || !len[i]) continue;
    if ((size_t)off[i] + len[i] > a.jpeg_cap) {
      if (s.st[i] == UFD_OK) s.st[i] = UFD_E_TRUNCATED;
      if (s.status && s.status[i] == UFD_OK) s.status[i] = UFD_E_TRUNCATED;
      if (!s.status && rc == UFD_OK) rc = UFD_E_TRUNCATED;
      continue;
    }
    a.jpeg_off[i] = off[i], a.jpeg_len[i] = len[i];
    fit = std::max(fit, (size_t)off[i] + len[i]);
  }
  if (fit) {
    {
      std::lock_guard<std::mutex> lk(m->copy_mu);
      HIPC(m, hipMemcpyAsync(a.jpeg_out, s.d_enc_out, fit, hipMemcpyDeviceToHost, m->copy_stream));
      HIPC(m, record_behind_copy(s.enc_copied, m->copy_stream));
    }
    HIPC(m, hipEventSynchronize(s.enc_copied));
  }
  return rc;
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
int finish_slot(ufd_model* m, Slot& s, bool locked = true) {
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

// Host entropy decode of `count` JPEGs into the slot, then enqueue the whole GPU pipeline.
int status_from_jpeg(int st) {
  return st == kJpegOk ? UFD_OK : (st == kJpegCorrupt ? UFD_E_DECODE : (st == kJpegUnsupported ? UFD_E_UNSUPPORTED : st));
}

// Index of this frame's Huffman table set in d_sync_luts (uploading it first if it is new), or -1.
int lut_set_for(ufd_model* m, const HuffLut (&luts)[4], uint64_t seq) {
  uint64_t h = 1469598103934665603ull;  // FNV-1a over the four tables: one compare per cached set instead of a 40 KB memcmp
  const uint8_t* bytes = reinterpret_cast<const uint8_t*>(luts);
  for (size_t i = 0; i < sizeof(HuffLut) * 4; i += 8) {
    uint64_t w;
    std::memcpy(&w, bytes + i, 8);
    h = (h ^ w) * 1099511628211ull;
  }
  std::lock_guard<std::mutex> lk(m->shared_mu);
  for (size_t i = 0; i < m->lut_sets.size(); i++)
    if (m->lut_meta[i].hash == h && !std::memcmp(m->lut_sets[i].data(), luts, sizeof(HuffLut) * 4)) {
      m->lut_meta[i].last_use = seq;
      return (int)i;
    }
  size_t idx = m->lut_sets.size();
  if ((int)idx >= ufd_model::kMaxLutSets) {
    // evict the least recently used set no in-flight batch can refer to: at most UFD_MAX_SLOTS batches are in flight, each
    // planned after the one before, so a set last used more than 2 * UFD_MAX_SLOTS plans ago is idle; staged batches pin theirs
    idx = SIZE_MAX;
    for (size_t i = 0; i < m->lut_sets.size(); i++) {
      const auto& q = m->lut_meta[i];
      if (q.pins || seq < q.last_use + 2 * UFD_MAX_SLOTS + 1) continue;
      if (idx == SIZE_MAX || q.last_use < m->lut_meta[idx].last_use) idx = i;
    }
    if (idx == SIZE_MAX) return -1;  // every set is busy: this batch decodes on the host workers
    // (batches enqueued long ago may still be running on the GPU: drain the contexts before their table goes away)
    for (int c = 0; c < m->num_ctx; c++)
      if (m->ctx[c].stream) (void)hipStreamSynchronize(m->ctx[c].stream);
  }
  std::array<HuffLut, 4> set;
  std::memcpy(set.data(), luts, sizeof(HuffLut) * 4);
  // rare (once per camera stream): blocking copy into the slot of the shared table array
  {
    std::unique_ptr<SyncLutImage> img(new SyncLutImage);
    build_sync_lut_image(set.data(), img.get());
    if (hipMemcpy(m->d_sync_luts + idx, img.get(), sizeof(SyncLutImage), hipMemcpyHostToDevice) != hipSuccess) return -1;
  }
  ufd_model::LutMeta meta;
  meta.hash = h, meta.last_use = seq;
  if (idx < m->lut_meta.size()) meta.gen = m->lut_meta[idx].gen + 1;
  if (idx == m->lut_sets.size()) {
    m->lut_sets.push_back(set);
    m->lut_meta.push_back(meta);
  } else {
    m->lut_sets[idx] = set;
    m->lut_meta[idx] = meta;
  }
  return (int)idx;
}

// The table set of a frame planned WITHOUT its lookup tables, by the key of its DHT bytes: index in d_sync_luts, or -1 when
// this key has not been seen (or its set has been evicted since).
int lut_set_by_key(ufd_model* m, const GpuScanPlan& p, uint64_t seq) {
  if (!p.key_hash) return -1;
  std::lock_guard<std::mutex> lk(m->shared_mu);
  for (const auto& k : m->lut_keys) {
    if (k.hash != p.key_hash || k.bytes.size() != p.key_len || std::memcmp(k.bytes.data(), p.key_bytes, p.key_len)) continue;
    if (k.set < 0 || (size_t)k.set >= m->lut_meta.size() || m->lut_meta[k.set].gen != k.gen) return -1;
    m->lut_meta[k.set].last_use = seq;
    return k.set;
  }
  return -1;
}

void remember_lut_key(ufd_model* m, const GpuScanPlan& p, int set) {
  if (!p.key_hash || set < 0) return;
  std::lock_guard<std::mutex> lk(m->shared_mu);
  ufd_model::LutKey k;
  k.hash = p.key_hash, k.bytes.assign(p.key_bytes, p.key_bytes + p.key_len), k.set = set, k.gen = m->lut_meta[set].gen;
  for (auto& e : m->lut_keys)
    if (e.hash == k.hash && e.bytes == k.bytes) {
      e = std::move(k);
      return;
    }
  if (m->lut_keys.size() < 2 * (size_t)ufd_model::kMaxLutSets) {
    m->lut_keys.push_back(std::move(k));
  } else {
    m->lut_keys[m->lut_key_next] = std::move(k);
    m->lut_key_next = (m->lut_key_next + 1) % m->lut_keys.size();
  }
}

// staged batches keep their table sets resident (ufd_stage_jpeg_batch / ufd_staged_free)
void pin_lut_sets(ufd_model* m, const HuffScan* scans, uint32_t count, int delta) {
  std::lock_guard<std::mutex> lk(m->shared_mu);
  for (uint32_t i = 0; i < count; i++) {
    if (!scans[i].nseg) continue;
    const size_t set = scans[i].lut_base / 4;
    if (set < m->lut_meta.size()) m->lut_meta[set].pins += delta;
  }
}

// Host half of the device entropy path: header / marker scan of every frame (no bit is decoded),
// scan layouts, table sets and intervals into the slot's pinned arrays, JPEG bytes into h_blob.
DevicePlan plan_device_entropy(ufd_model* m, Slot& s, const uint8_t* const* jpegs, const size_t* lens, uint32_t count) {
  DevicePlan p;
  HostScope hs(m, "host_plan");
  const uint64_t t_plan0 = now_ns();
  // The header scan is 4 us per frame and the staging copy 1-2 us (35 KB): a batch of 32 is 0.2 ms on the issuing worker
  // itself, deterministically.  Handing it to the pool (round 3) is faster on a quiet host -- 40-60 us -- but every
  // parallel_for wakes sleeping threads and waits for the LAST of them: on a host whose CPUs are busy elsewhere (eight
  // ranks and other tenants on one box) the same two calls took 240 + 140 us per batch, the context's stream sat idle
  // 480 us between batches and the frame rate fell from 53 k to 43 k (profiles/r4z/bench_driver_flags.json: `host`).  The
  // pool is used only when the batch's bytes make the copy worth it (large frames).
  size_t batch_bytes = 0;
  for (uint32_t i = 0; i < count; i++) batch_bytes += lens[i];
  const bool use_pool = m->plan_parallel || batch_bytes > ((size_t)8 << 20);
  auto for_each_frame = [&](const std::function<void(unsigned)>& fn) {
    if (use_pool) tl_pool->parallel_for(count, fn);
    else
      for (uint32_t i = 0; i < count; i++) fn(i);
  };
  for_each_frame([&](unsigned i) {
    JpegFrameDesc* d = &s.h_descs[i];
    // (no lookup tables yet: frames of a camera stream share their DHT bytes, found below by key)
    int st = (jpegs[i] && lens[i]) ? jpeg_plan_gpu_scan(jpegs[i], lens[i], d, &s.plans[i], /*build_luts=*/false) : kJpegCorrupt;
    if (st == kJpegOk && ((uint32_t)d->width > m->max_w || (uint32_t)d->height > m->max_h)) st = UFD_E_TOO_LARGE;
    if (st == kJpegOk && (lens[i] + 64 > m->blob_stride || d->coef_total > m->coef_stride)) st = kJpegNotEligible;
    s.st[i] = st;
  });
  uint64_t seq;
  {
    std::lock_guard<std::mutex> lk(m->shared_mu);
    seq = ++m->plan_seq;
  }
  uint32_t n_iv = 0;
  for (uint32_t i = 0; i < count; i++) {
    if (s.st[i] == kJpegNotEligible) return p;
    if (s.st[i] == kJpegOk) n_iv += s.plans[i].n_intervals;
  }
  if (n_iv > m->iv_cap) return p;
  // JPEG bytes packed back to back behind the interval table (16-byte aligned starts, 64 bytes of
  // slack behind every frame: the unstuff kernel reads whole 16-byte pieces)
  p.blob_base = (m->ivs_off + (size_t)n_iv * sizeof(HuffInterval) + 255) & ~(size_t)255;
  s.h_blob = s.h_stage + p.blob_base;
  size_t blob_fill = 0;
  uint32_t k = 0;
  for (uint32_t i = 0; i < count; i++) {
    std::memset(&s.h_scans[i], 0, sizeof(HuffScan));  // nseg = 0: the frame's workgroups exit at once
    if (s.st[i] != kJpegOk) continue;
    int set = lut_set_by_key(m, s.plans[i], seq);
    if (set < 0) {  // first frame with these tables (once per camera stream): build them, upload the set, remember the key
      JpegFrameDesc again;
      if (jpeg_plan_gpu_scan(jpegs[i], lens[i], &again, &s.plans[i], /*build_luts=*/true) != kJpegOk) return p;
      set = lut_set_for(m, s.plans[i].luts, seq);
      if (set < 0) return p;
      remember_lut_key(m, s.plans[i], set);
    }
    HuffScan sc = s.plans[i].scan;
    sc.lut_base = (uint32_t)set * 4;
    sc.seg_base = sc.nseg = sc.sub_bytes = sc.nsub = sc.pad = 0;
    sc.blob_off = (uint32_t)blob_fill;
    blob_fill += (lens[i] + 64 + 63) & ~(size_t)63;
    const uint32_t nseg = s.plans[i].n_intervals;
    {
      // subsequence slots: every segment starts on a subsequence boundary and keeps 32 bytes of
      // zero slack behind its data; longer streams get longer subsequences (<= kSyncMaxSub slots)
      if (nseg < 1 || nseg > (uint32_t)kSyncMaxSeg) return p;
      size_t padded = 0;
      for (uint32_t j = 0; j < nseg; j++) padded += (size_t)(s.plans[i].iv[j].end - s.plans[i].iv[j].begin) + 32;
      uint32_t sub = (uint32_t)((padded + (kSyncMaxSub - nseg) - 1) / (kSyncMaxSub - nseg));
      sub = std::max((sub + 3u) & ~3u, 64u);
      uint32_t first = 0;
      for (uint32_t j = 0; j < nseg; j++) {
        HuffInterval iv = s.plans[i].iv[j];
        iv.frame = i;
        iv.first_sub = first;
        first += (iv.end - iv.begin + 32 + sub - 1) / sub;
        s.h_ivs[k + j] = iv;
      }
      if (first > (uint32_t)kSyncMaxSub || (size_t)first * sub + 64 > m->blob_stride) return p;
      sc.seg_base = k, sc.nseg = nseg, sc.sub_bytes = sub, sc.nsub = first;
      k += nseg;
      p.max_nsub = std::max(p.max_nsub, first);
      p.max_bpm = std::max(p.max_bpm, sc.blocks_per_mcu);
    }
    s.h_scans[i] = sc;
  }
  if (p.blob_base + blob_fill > m->stage_cap) return p;
  const uint64_t t_copy0 = now_ns();
  for_each_frame([&](unsigned i) {
    if (s.st[i] == kJpegOk) std::memcpy(s.h_blob + s.h_scans[i].blob_off, jpegs[i], lens[i]);
  });
  if (tl_worker) {
    const uint64_t t_copy1 = now_ns();
    tl_worker->ns_plan.fetch_add(t_copy0 - t_plan0, std::memory_order_relaxed);
    tl_worker->ns_copy.fetch_add(t_copy1 - t_copy0, std::memory_order_relaxed);
  }
  p.used_blob = blob_fill;
  p.stage_bytes = p.blob_base + blob_fill;
  for (uint32_t i = 0; i < count; i++) {
    if (s.st[i] == kJpegOk) {
      p.any_ok = true;
      p.used_coef = std::max(p.used_coef, (size_t)s.h_descs[i].coef_total);
    } else {
      std::memset(&s.h_descs[i], 0, sizeof(JpegFrameDesc));
    }
    s.st[i] = status_from_jpeg(s.st[i]);
  }
  p.ok = true;
  p.n_iv = k;
  return p;
}

// Device half: zeroes the coefficient slabs and decodes into them on the context's stream.  All
// pointers are device memory (the context's double buffers, or a staged batch).
int enqueue_device_entropy(ufd_model* m, Ctx& c, const DevicePlan& p, uint32_t count, const uint8_t* d_blob,
                           const JpegFrameDesc* d_descs, const HuffScan* d_scans, const HuffInterval* d_ivs, int16_t* d_coef) {
  // (the slabs, the DC side array and d_status are cleared by the chain's first launch, beside the unstuffing: a frame
  // the decoder flags corrupt then reconstructs from zeros, not from what an earlier batch left there)
  {
    std::unique_ptr<ProfScope> scope;
    const double bytes = (double)p.used_blob;
    const HuffStageHook hook = [&](const char* kernel, bool begin) {
      if (begin) scope.reset(new ProfScope(m, kernel, bytes, 0));
      else scope.reset();
    };
    launch_huffman_sync(d_blob, d_scans, d_ivs, count, p.max_nsub, p.max_bpm, m->d_sync_luts, d_descs, d_coef, m->coef_stride,
                        c.sync, c.d_status, c.stream, &hook, p.used_coef);
  }
  return UFD_OK;
}

// Stage 1 of row A1 for `count` JPEGs: leaves quantised coefficient slabs in d_coef_buf[*buf] and
// frame descriptors in d_descs_buf[*buf], ordered before later work on the context's stream.
//   device path: header/marker scan on host workers, JPEG bytes H2D, entropy decoding kernels
//   host path:   Huffman decoding on host workers, coefficient slabs H2D
int entropy_stage(ufd_model* m, Slot& s, const uint8_t* const* jpegs, const size_t* lens, uint32_t count, int* buf_out,
                  bool* any_ok_out) {
  Ctx& c = *tl_cur;
  if (m->gpu_entropy_enabled) {
    const DevicePlan p = plan_device_entropy(m, s, jpegs, lens, count);
    if (p.ok) {
      s.gpu_entropy = true;
      s.coef_zigzag = true;
      *any_ok_out = p.any_ok;
      if (!p.any_ok) return UFD_OK;
      const int buf = c.flip;
      c.flip ^= 1;
      *buf_out = buf;
      {
        // the handle's one copy stream carries the copies of all contexts: its enqueue order is
        // serialised here (descriptors + scan plans + intervals + JPEG bytes in ONE contiguous copy)
        std::lock_guard<std::mutex> lk(m->copy_mu);
        if (c.consumed_valid[buf]) HIPC(m, hipStreamWaitEvent(c.copy_stream, c.ev_consumed[buf], 0));
        ProfScope ps(m, "h2d_jpeg", (double)p.stage_bytes, 0, c.copy_stream);
        HIPC(m, hipMemcpyAsync(c.d_stage_buf[buf], s.h_stage, p.stage_bytes, hipMemcpyHostToDevice, c.copy_stream));
        HIPC(m, record_behind_copy(c.ev_copied[buf], c.copy_stream));
      }
      HIPC(m, hipStreamWaitEvent(c.stream, c.ev_copied[buf], 0));
      span_begin(s);
      uint8_t* ds = c.d_stage_buf[buf];
      return enqueue_device_entropy(m, c, p, count, ds + p.blob_base, c.d_descs_buf[buf], reinterpret_cast<const HuffScan*>(ds + m->scans_off),
                                    reinterpret_cast<const HuffInterval*>(ds + m->ivs_off), c.d_coef_buf[buf]);
    }
  }
  // ---- host entropy decoding
  if (!s.h_coef)  // pinned coefficient slabs: only handles / batches that decode on the host need them
    HIPC(m, hipHostMalloc(&s.h_coef, sizeof(int16_t) * m->coef_stride * m->B, hipHostMallocDefault));
  s.gpu_entropy = false;
  s.coef_zigzag = false;
  tl_pool->parallel_for(count, [&](unsigned i) {
    JpegFrameDesc* d = &s.h_descs[i];
    int st = (jpegs[i] && lens[i]) ? jpeg_decode_coefficients(jpegs[i], lens[i], d, s.h_coef + (size_t)i * m->coef_stride,
                                                               m->coef_stride)
                                   : kJpegCorrupt;
    if (st == kJpegOk && ((uint32_t)d->width > m->max_w || (uint32_t)d->height > m->max_h)) st = UFD_E_TOO_LARGE;
    if (st != kJpegOk) std::memset(d, 0, sizeof(*d));  // total_blocks = 0, width = 0: every kernel skips the frame
    s.st[i] = status_from_jpeg(st);
  });
  bool any_ok = false;
  size_t used = 0;
  for (uint32_t i = 0; i < count; i++) {
    if (s.st[i] == UFD_OK) any_ok = true;
    used = std::max(used, (size_t)s.h_descs[i].coef_total);
  }
  *any_ok_out = any_ok;
  if (!any_ok) return UFD_OK;
  const int buf = c.flip;
  c.flip ^= 1;
  *buf_out = buf;
  // copy stream: wait until the kernels of two batches ago have consumed this buffer
  std::lock_guard<std::mutex> copy_lk(m->copy_mu);
  if (c.consumed_valid[buf]) HIPC(m, hipStreamWaitEvent(c.copy_stream, c.ev_consumed[buf], 0));
  HIPC(m, hipMemcpyAsync(c.d_descs_buf[buf], s.h_descs, sizeof(JpegFrameDesc) * count, hipMemcpyHostToDevice, c.copy_stream));
  {
    ProfScope ps(m, "h2d_coef", 0, 0, c.copy_stream);
    // frames are equally sized in a stream: copy the used prefix of every slab in one 2-D copy
    HIPC(m, hipMemcpy2DAsync(c.d_coef_buf[buf], m->coef_stride * 2, s.h_coef, m->coef_stride * 2, used * 2, count,
                             hipMemcpyHostToDevice, c.copy_stream));
  }
  HIPC(m, record_behind_copy(c.ev_copied[buf], c.copy_stream));
  HIPC(m, hipStreamWaitEvent(c.stream, c.ev_copied[buf], 0));
  span_begin(s);
  return UFD_OK;
}

// ---------------------------------------------------------------- N1: rectangles + JPEG re-encode (inferer.rs:38-40)
size_t enc_frame_bound(uint32_t w, uint32_t h) {
  const size_t mcus = (size_t)((w + 15) / 16) * ((h + 15) / 16);
  return 2 * enc_stream_bound(mcus) + 1024;  // every entropy-coded byte stuffed + header, EOI, framing
}

// Encoder scratch of a context for frames of up to mw x mh (grown on demand: a camera stream has one frame size, so
// this happens on its first annotate batch) and the set-up of the requested (quality, framing).
int ensure_encoder(ufd_model* m, Ctx& c, uint32_t quality, bool multipart, uint32_t mw, uint32_t mh) {
  if (quality < 1 || quality > 100) return m->fail(UFD_E_ARG, "quality must be in 1..100");
  EncBuffers& e = c.enc;
  if (!c.enc_ready) {  // fixed-size pieces, once
    HIPC(m, hipMalloc(&e.total_bits, sizeof(uint32_t) * m->B));
    HIPC(m, hipMalloc(&c.d_enc_tables, sizeof(uint32_t) * 2 * 272));
    HIPC(m, hipMalloc(&c.d_enc_descs, sizeof(JpegFrameDesc) * m->B));
    HIPC(m, hipMalloc(&c.d_label_ops, label_ops_bytes(m->B, (uint32_t)m->K)));
    {
      const int* g;
      const float* cov;
      size_t ng, nc;
      label_atlas(&g, &ng, &cov, &nc);
      HIPC(m, hipMalloc(&c.d_glyphs, ng * sizeof(int)));
      HIPC(m, hipMalloc(&c.d_coverage, nc * sizeof(float)));
      HIPC(m, hipMemcpy(c.d_glyphs, g, ng * sizeof(int), hipMemcpyHostToDevice));
      HIPC(m, hipMemcpy(c.d_coverage, cov, nc * sizeof(float), hipMemcpyHostToDevice));
    }
    uint32_t tabs[2 * 272];
    enc_make_code_tables(tabs);
    HIPC(m, hipMemcpy(c.d_enc_tables, tabs, sizeof(tabs), hipMemcpyHostToDevice));
    e.tables = c.d_enc_tables;
    c.enc_ready = true;
  }
  const size_t mcus = (size_t)((mw + 15) / 16) * ((mh + 15) / 16);
  if (mcus > c.enc_mcus) {
    // earlier encodes of this context may still use the old scratch: drain, free, allocate the larger set.  A failed
    // allocation leaves enc_mcus = 0 and null pointers behind (nothing leaks, the next batch tries again).
    HIPC(m, hipStreamSynchronize(c.stream));
    (void)hipFree(e.planes), (void)hipFree(e.coef), (void)hipFree(e.bits), (void)hipFree(e.words), (void)hipFree(e.chunk_ff);
    e.planes = nullptr, e.coef = nullptr, e.bits = nullptr, e.words = nullptr, e.chunk_ff = nullptr;
    c.enc_mcus = 0;
    const size_t sb = enc_stream_bound(mcus);
    e.coef_stride = mcus * 6 * 64;
    e.blk_stride = mcus * 6;
    e.word_stride = (((sb + 3) / 4 + 3) & ~(size_t)3) + 8;  // whole 16-byte groups + the padding the scan zeroes
    e.chunk_stride = (sb + 4095) / 4096 + 1;
    e.plane_stride = mcus * 384;  // 256 luma + 2 x 64 chroma samples per MCU
    HIPC(m, hipMalloc(&e.planes, e.plane_stride * m->B));
    HIPC(m, hipMalloc(&e.coef, sizeof(int16_t) * e.coef_stride * m->B));
    HIPC(m, hipMalloc(&e.bits, sizeof(uint32_t) * e.blk_stride * m->B));
    HIPC(m, hipMalloc(&e.words, sizeof(uint32_t) * e.word_stride * m->B));
    HIPC(m, hipMalloc(&e.chunk_ff, sizeof(uint32_t) * e.chunk_stride * m->B));
    c.enc_mcus = mcus;
  }
  // the (quality, framing) set-up: a cached one, a free entry (its fresh device header has no reader yet: no drain), or
  // the least recently used entry, rewritten behind the stream's earlier encodes
  int pick = -1, lru = 0;
  for (int i = 0; i < Ctx::kEncSetups; i++) {
    const Ctx::EncSetup& q = c.enc_setups[i];
    if (q.quality == (int)quality && q.multipart == (int)multipart) pick = i;
    if (q.last_use < c.enc_setups[lru].last_use) lru = i;
  }
  if (pick < 0) {
    pick = lru;
    Ctx::EncSetup& q = c.enc_setups[pick];
    uint8_t ql[64], qc[64], hdr[1024];
    enc_quant_tables((int)quality, ql, qc);
    q.ifast = quality < 96;  // turbojpeg.c setCompDefaults: JDCT_ISLOW from quality 96 on, JDCT_FASTEST below
    enc_make_quant(ql, qc, q.ifast, &q.q);
    const size_t n = enc_make_header(ql, qc, multipart, hdr, &q.pre_len, &q.hdr_len, &q.dim_off, &q.post_len);
    if (q.d_header) HIPC(m, hipStreamSynchronize(c.stream));  // an evicted set-up: earlier encodes still read its header
    else HIPC(m, hipMalloc(&q.d_header, 1024));
    q.quality = -1;
    HIPC(m, hipMemcpy(q.d_header, hdr, n, hipMemcpyHostToDevice));
    q.quality = (int)quality, q.multipart = (int)multipart;
  }
  Ctx::EncSetup& q = c.enc_setups[pick];
  q.last_use = ++c.enc_seq;
  c.enc_cur = pick;
  e.header = q.d_header;
  e.pre_len = q.pre_len, e.hdr_len = q.hdr_len, e.dim_off = q.dim_off, e.post_len = q.post_len;
  return UFD_OK;
}

// Output of a slot's batch for frames of up to mw x mh (grown on demand; the slot is ours and its previous batch has
// been waited for, so nothing on the GPU still refers to the old buffer).
int ensure_slot_encoder(ufd_model* m, Slot& s, uint32_t mw, uint32_t mh) {
  if (!s.d_enc_meta) {
    HIPC(m, hipMalloc(&s.d_enc_meta, sizeof(uint32_t) * (2 * m->B + 1)));
    HIPC(m, hipHostMalloc(&s.h_enc_meta, sizeof(uint32_t) * (2 * m->B + 1), hipHostMallocDefault));
    HIPC(m, hipEventCreateWithFlags(&s.enc_copied, hipEventDisableTiming));
  }
  const size_t need = enc_frame_bound(mw, mh) * m->B;
  if (need > s.enc_out_cap) {
    (void)hipFree(s.d_enc_out);
    s.d_enc_out = nullptr, s.enc_out_cap = 0;
    HIPC(m, hipMalloc(&s.d_enc_out, need));
    s.enc_out_cap = need;
  }
  return UFD_OK;
}

// Rectangles of the slot's detections into the context's RGB frames, then the encoder; lengths / offsets of the
// finished streams follow the detections to the host.  On the context's stream, behind the NMS.
int enqueue_annotate(ufd_model* m, Slot& s, const JpegFrameDesc* d_descs, uint32_t mw, uint32_t mh, uint32_t count) {
  Ctx& c = *tl_cur;
  int rc = ensure_encoder(m, c, s.annot_args.quality, (s.annot_args.flags & UFD_ANNOT_MULTIPART) != 0, mw, mh);
  if (rc) return rc;
  rc = ensure_slot_encoder(m, s, mw, mh);
  if (rc) return rc;
  {
    ProfScope ps(m, "draw_labels", 0, 0);
    launch_draw_labels(d_descs, s.d_dets, (uint32_t)m->K, c.d_ndet, (uint32_t)m->K, c.d_label_ops, c.d_glyphs, c.d_coverage,
                       !(s.annot_args.flags & UFD_ANNOT_NO_TEXT), c.d_rgb, m->rgb_stride, mw, mh, s.annot_args.label_width,
                       s.annot_args.label_height, count, c.stream);
  }
  EncBuffers e = c.enc;
  e.out = s.d_enc_out;
  e.out_len = s.d_enc_meta, e.out_off = s.d_enc_meta + m->B, e.out_total = s.d_enc_meta + 2 * m->B;
  {
    std::unique_ptr<ProfScope> scope;
    const double bytes = (double)count * mw * mh * 3.0;
    const EncStageHook hook = [&](const char* stage, bool begin) {
      if (begin) scope.reset(new ProfScope(m, stage, bytes, 0));
      else scope.reset();
    };
    launch_jpeg_encode(d_descs, c.d_rgb, m->rgb_stride, mw, mh, count, c.enc_setups[c.enc_cur].q, c.enc_setups[c.enc_cur].ifast, e, c.stream,
                       &hook);
  }
  HIPC(m, hipMemcpyAsync(s.h_enc_meta, s.d_enc_meta, sizeof(uint32_t) * (2 * m->B + 1), hipMemcpyDeviceToHost, c.stream));
  s.annot_ran = true;
  return UFD_OK;
}

int run_decoded(ufd_model* m, Slot& s, uint32_t count, bool any_ok, const JpegFrameDesc* d_descs, int16_t* d_coef, int buf);

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
  s.coef_zigzag = true;
  int buf = 0;
  if (g.plan.any_ok) {
    buf = c.flip;
    c.flip ^= 1;
    // the slab is written on the context's own stream: ordered behind its previous readers; a
    // host-path batch that reuses it later waits for ev_consumed as usual
    span_begin(s);
    rc = enqueue_device_entropy(m, c, g.plan, count, g.d_blob, g.d_descs, g.d_scans, g.d_ivs, c.d_coef_buf[buf]);
    if (rc) return rc;
  }
  return run_decoded(m, s, count, g.plan.any_ok, g.d_descs, c.d_coef_buf[buf], buf);
}

// Every decodable frame of the batch is 3-component YCbCr 4:2:0 (what the fancy-upsampling fast paths take).
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
      // 4:2:0 frames at the model size: the stem conv reads the sample planes itself (no f32 input
      // tensor); UFD_NO_STEM_FUSE=1 at ufd_create keeps the two-kernel path
      if (all_420 && m->stem_fusable) {
        tl_cur->stem_descs = d_descs;
        fused_stem = true;
      } else {
        const double bytes = (double)count * (m->W * m->H * 1.5 + m->W * m->H * 12.0);
        ProfScope ps(m, all_420 ? "upsample_norm_420" : "upsample_norm", bytes, 0);
        if (all_420)
          launch_upsample_norm_420(d_descs, tl_cur->d_planes, m->plane_stride, m->d_lut, tl_cur->d_input, m->W, m->H, count, tl_cur->stream);
        else
          launch_upsample_norm(d_descs, tl_cur->d_planes, m->plane_stride, m->d_lut, tl_cur->d_input, m->W, m->H, count, tl_cur->stream);
        HIPC(m, hipEventRecord(tl_cur->ev_consumed[buf], tl_cur->stream));
        tl_cur->consumed_valid[buf] = true;
      }
    } else {
      enqueue_upsample_rgb(m, s, d_descs, mw, mh, count);
      HIPC(m, hipEventRecord(tl_cur->ev_consumed[buf], tl_cur->stream));
      tl_cur->consumed_valid[buf] = true;
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
    enqueue_forward(m, count);
    if (fused_stem) {  // the stem was the last reader of the descriptors / planes of this buffer
      tl_cur->stem_descs = nullptr;
      HIPC(m, hipEventRecord(tl_cur->ev_consumed[buf], tl_cur->stream));
      tl_cur->consumed_valid[buf] = true;
    }
    enqueue_heads(m, count);
    enqueue_nms(m, s, count);
    if (s.annot) {
      rc = enqueue_annotate(m, s, d_descs, mw, mh, count);
      if (rc) return rc;
      // the encoder was the last reader of this buffer's descriptors
      HIPC(m, hipEventRecord(tl_cur->ev_consumed[buf], tl_cur->stream));
      tl_cur->consumed_valid[buf] = true;
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

// ---------------------------------------------------------------- host placement
// "0-3,8,10-11" -> cpu ids
std::vector<int> parse_cpu_list(const std::string& txt) {
  std::vector<int> out;
  size_t i = 0;
  while (i < txt.size()) {
    while (i < txt.size() && !std::isdigit((unsigned char)txt[i])) i++;
    if (i >= txt.size()) break;
    int a = 0;
    while (i < txt.size() && std::isdigit((unsigned char)txt[i])) a = a * 10 + (txt[i++] - '0');
    int b = a;
    if (i < txt.size() && txt[i] == '-') {
      i++;
      b = 0;
      while (i < txt.size() && std::isdigit((unsigned char)txt[i])) b = b * 10 + (txt[i++] - '0');
    }
    for (int c = a; c <= b && out.size() < 4096; c++) out.push_back(c);
  }
  return out;
}

std::string read_first_line(const std::string& path) {
  std::ifstream f(path);
  std::string line;
  if (f) std::getline(f, line);
  return line;
}

// NUMA node of the device (/sys/bus/pci/devices/<bdf>/numa_node) and the CPUs of that node inside this process's
// affinity mask.  Nothing is pinned when the node is unknown (-1: one socket, or a VM that hides the topology), when
// the mask and the node do not intersect, or with UFD_FLAG_NO_NUMA_PIN.
void resolve_placement(ufd_model* m) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), m->cfg.device_id) != hipSuccess) return;
  for (char* p = bdf; *p; p++) *p = (char)std::tolower((unsigned char)*p);
  m->pci_bdf = bdf;
  const std::string node_txt = read_first_line(std::string("/sys/bus/pci/devices/") + bdf + "/numa_node");
  if (node_txt.empty()) return;
  m->numa_node = std::atoi(node_txt.c_str());
  if (m->numa_node < 0 || (m->cfg.flags & UFD_FLAG_NO_NUMA_PIN)) return;
  const std::vector<int> node_cpus =
      parse_cpu_list(read_first_line("/sys/devices/system/node/node" + std::to_string(m->numa_node) + "/cpulist"));
  cpu_set_t cur;
  CPU_ZERO(&cur);
  if (sched_getaffinity(0, sizeof(cur), &cur) != 0) return;
  for (int c : node_cpus)
    if (c < CPU_SETSIZE && CPU_ISSET(c, &cur)) m->pin_cpus.push_back(c);
  // compact "a-b,c" form for reports
  std::string txt;
  for (size_t i = 0; i < m->pin_cpus.size();) {
    size_t j = i;
    while (j + 1 < m->pin_cpus.size() && m->pin_cpus[j + 1] == m->pin_cpus[j] + 1) j++;
    txt += (txt.empty() ? "" : ",") + std::to_string(m->pin_cpus[i]) + (j > i ? "-" + std::to_string(m->pin_cpus[j]) : "");
    i = j + 1;
  }
  m->cpu_list = txt;
}

// Calling thread -> the handle's CPUs (no-op when nothing was resolved).
void pin_this_thread(const ufd_model* m) {
  if (m->pin_cpus.empty()) return;
  cpu_set_t set;
  CPU_ZERO(&set);
  for (int c : m->pin_cpus) CPU_SET(c, &set);
  (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
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

std::string default_weights_path(int variant) {
  // dirs::cache_dir()/infercam_onnx/ultraface-RFB-{640,320}.onnx (nn.rs:144-156)
  const char* xdg = std::getenv("XDG_CACHE_HOME");
  std::string base;
  if (xdg && *xdg) {
    base = xdg;
  } else {
    const char* home = std::getenv("HOME");
    base = std::string(home ? home : ".") + "/.cache";
  }
  return base + "/infercam_onnx/ultraface-RFB-" + std::to_string(variant) + ".onnx";
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
  if (m->copy_stream) (void)hipStreamSynchronize(m->copy_stream);
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
      if (c.ev_copied[i]) (void)hipEventDestroy(c.ev_copied[i]);
      if (c.ev_consumed[i]) (void)hipEventDestroy(c.ev_consumed[i]);
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
  for (auto& pe : m->prof_pending) m->prof_free.push_back(pe.e0), m->prof_free.push_back(pe.e1);
  for (auto e : m->prof_free) (void)hipEventDestroy(e);
  for (Ctx& c : m->ctx) {
    if (c.stream) (void)hipStreamDestroy(c.stream);
  }
  if (m->copy_stream) (void)hipStreamDestroy(m->copy_stream);
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
  if (const char* e = std::getenv("UFD_PLAN_PARALLEL")) m->plan_parallel = std::atoi(e) != 0;
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
  // More than four live HSA queues cost throughput (host pipeline): if the process raised the
  // runtime's cap, stay at two contexts = four streams (34 k instead of 30 k frames/s at GPU_MAX_HW_QUEUES=8).
  if (const char* q = std::getenv("GPU_MAX_HW_QUEUES"))
    if (std::atoi(q) > 4) m->num_ctx = 2;
  for (int ci = 0; ci < m->num_ctx; ci++) {
    Ctx& c = m->ctx[ci];
    HIPB(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    // (3 compute streams + 1 copy stream = the runtime's 4 hardware queues: a copy stream per
    // context made 6 streams share 4 queues, and the host-boundary rate fell to 0.69 of the staged one)
    if (!m->copy_stream) HIPB(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
    c.copy_stream = m->copy_stream;
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
    // worst-case slab: every component at full resolution, padded to 16-pixel MCUs
    const size_t pw = (m->max_w + 15) / 16 * 16, ph = (m->max_h + 15) / 16 * 16;
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
      HIPB(hipEventCreateWithFlags(&c.ev_copied[i], hipEventDisableTiming));
      HIPB(hipEventCreateWithFlags(&c.ev_consumed[i], hipEventDisableTiming));
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
    HIPB(hipMalloc(&c.d_nms_mat, nms_matrix_bytes((uint32_t)B)));
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

template <typename F>
int guarded(ufd_model* m, F&& f) {
  if (!m) return UFD_E_ARG;
  std::lock_guard<std::mutex> lk(m->mu);
  try {
    if (hipSetDevice(m->cfg.device_id) != hipSuccess) return m->fail(UFD_E_DEVICE, "hipSetDevice failed");
    tl_cur = &m->ctx[0];  // synchronous calls and taps run on context 0 from the calling thread
    tl_pool = m->pool.get();
    tl_prof = true;
    return f();
  } catch (const std::exception& e) {
    return m->fail(UFD_E_DEVICE, std::string("exception: ") + e.what());
  } catch (...) {
    return m->fail(UFD_E_DEVICE, "unknown exception");
  }
}

}  // namespace

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

int ufd_annotate_parity(uint32_t annot_flags) {
  return (annot_flags & UFD_ANNOT_NO_TEXT) ? UFD_PARITY_EXACT : UFD_PARITY_LABELS_UNPINNED;
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

size_t ufd_encode_bound(uint32_t w, uint32_t h) { return enc_frame_bound(w, h) + 64; }

void* ufd_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
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

// ---------------------------------------------------------------- stage taps
// N1 stages on a caller-provided RGB frame (context 0, synchronous)
static int upload_plain_frame(ufd_model* m, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t pitch) {
  int rc = upload_rgb(m, rgb, w, h, pitch, 1);
  if (rc) return rc;
  JpegFrameDesc d;
  std::memset(&d, 0, sizeof(d));
  d.width = (int32_t)w, d.height = (int32_t)h;
  HIPC(m, hipMemcpyAsync(tl_cur->d_enc_descs, &d, sizeof(d), hipMemcpyHostToDevice, tl_cur->stream));
  return UFD_OK;
}

int ufd_debug_draw_labels(ufd_model* m, uint8_t* rgb, uint32_t w, uint32_t h, uint32_t pitch, const ufd_det* dets, uint32_t n,
                          float label_width, float label_height, uint32_t text) {
  return guarded(m, [&]() -> int {
    drain_worker0(m);
    if (!rgb || (!dets && n)) return m->fail(UFD_E_ARG, "null argument");
    if (n > (uint32_t)m->K) return m->fail(UFD_E_TOO_LARGE, "more detections than priors");
    Slot* s = find_free_slot(m);
    if (!s) return m->fail(UFD_E_STATE, "all slots busy");
    int rc = alloc_slot(m, *s);
    if (rc) return rc;
    if (!w || !h || w > m->max_w || h > m->max_h) return m->fail(UFD_E_TOO_LARGE, "frame larger than max_src_width/height");
    rc = ensure_encoder(m, *tl_cur, 95, false, w, h);
    if (rc) return rc;
    rc = upload_plain_frame(m, rgb, w, h, pitch);
    if (rc) return rc;
    if (n) HIPC(m, hipMemcpyAsync(s->d_dets, dets, sizeof(Det) * n, hipMemcpyHostToDevice, tl_cur->stream));
    HIPC(m, hipMemcpyAsync(tl_cur->d_ndet, &n, sizeof(uint32_t), hipMemcpyHostToDevice, tl_cur->stream));
    launch_draw_labels(tl_cur->d_enc_descs, s->d_dets, (uint32_t)m->K, tl_cur->d_ndet, std::max(n, 1u), tl_cur->d_label_ops,
                       tl_cur->d_glyphs, tl_cur->d_coverage, text != 0, tl_cur->d_rgb, m->rgb_stride, w, h, label_width, label_height,
                       1, tl_cur->stream);
    HIPC(m, hipMemcpy2DAsync(rgb, pitch, tl_cur->d_rgb, (size_t)w * 3, (size_t)w * 3, h, hipMemcpyDeviceToHost, tl_cur->stream));
    HIPC(m, hipStreamSynchronize(tl_cur->stream));
    return UFD_OK;
  });
}

int ufd_debug_encode_rgb(ufd_model* m, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t pitch, uint32_t quality,
                         uint32_t flags, uint8_t* out, size_t cap, size_t* len) {
  return guarded(m, [&]() -> int {
    drain_worker0(m);
    if (!rgb || !out || !len) return m->fail(UFD_E_ARG, "null argument");
    Slot* s = find_free_slot(m);
    if (!s) return m->fail(UFD_E_STATE, "all slots busy");
    int rc = alloc_slot(m, *s);
    if (rc) return rc;
    if (!w || !h || w > m->max_w || h > m->max_h) return m->fail(UFD_E_TOO_LARGE, "frame larger than max_src_width/height");
    rc = ensure_encoder(m, *tl_cur, quality, (flags & UFD_ANNOT_MULTIPART) != 0, w, h);
    if (rc) return rc;
    rc = ensure_slot_encoder(m, *s, w, h);
    if (rc) return rc;
    rc = upload_plain_frame(m, rgb, w, h, pitch);
    if (rc) return rc;
    EncBuffers e = tl_cur->enc;
    e.out = s->d_enc_out;
    e.out_len = s->d_enc_meta, e.out_off = s->d_enc_meta + m->B, e.out_total = s->d_enc_meta + 2 * m->B;
    launch_jpeg_encode(tl_cur->d_enc_descs, tl_cur->d_rgb, m->rgb_stride, w, h, 1, tl_cur->enc_setups[tl_cur->enc_cur].q,
                       tl_cur->enc_setups[tl_cur->enc_cur].ifast, e, tl_cur->stream);
    HIPC(m, hipMemcpyAsync(s->h_enc_meta, s->d_enc_meta, sizeof(uint32_t) * (2 * m->B + 1), hipMemcpyDeviceToHost, tl_cur->stream));
    HIPC(m, hipStreamSynchronize(tl_cur->stream));
    *len = s->h_enc_meta[0];
    if (*len > cap) return m->fail(UFD_E_TRUNCATED, "encoded stream larger than the output buffer");
    HIPC(m, hipMemcpy(out, s->d_enc_out + s->h_enc_meta[m->B], *len, hipMemcpyDeviceToHost));
    return UFD_OK;
  });
}

int ufd_debug_decode_jpeg(ufd_model* m, const uint8_t* jpeg, size_t len, uint8_t* rgb, size_t cap_bytes, uint32_t* w,
                          uint32_t* h) {
  return guarded(m, [&]() -> int {
    drain_worker0(m);
    if (!jpeg || !len || !rgb) return m->fail(UFD_E_ARG, "null argument");
    Slot* s = find_free_slot(m);
    if (!s) return m->fail(UFD_E_STATE, "all slots busy");
    int rc = alloc_slot(m, *s);
    if (rc) return rc;
    // same stage-1 code as the batch path (device entropy decoding when the stream is eligible)
    int buf = 0;
    bool any_ok = false;
    rc = entropy_stage(m, *s, &jpeg, &len, 1, &buf, &any_ok);
    if (rc) return rc;
    if (s->st[0] == UFD_E_DECODE) return m->fail(UFD_E_DECODE, "corrupt JPEG");
    if (s->st[0] == UFD_E_TOO_LARGE) return m->fail(UFD_E_TOO_LARGE, "frame larger than max_src_width/height");
    if (s->st[0] != UFD_OK) return m->fail(UFD_E_UNSUPPORTED, "unsupported JPEG feature");
    const JpegFrameDesc* d = &s->h_descs[0];
    if (w) *w = d->width;
    if (h) *h = d->height;
    const size_t bytes = (size_t)d->width * d->height * 3;
    if (cap_bytes < bytes) return m->fail(UFD_E_ARG, "rgb buffer too small");
    launch_idct(tl_cur->d_descs_buf[buf], tl_cur->d_coef_buf[buf], m->coef_stride, tl_cur->d_planes, m->plane_stride,
                d->total_blocks, 1, s->coef_zigzag, tl_cur->stream, s->gpu_entropy ? tl_cur->sync.dc : nullptr,
                tl_cur->sync.dc_stride);
    launch_upsample_rgb(tl_cur->d_descs_buf[buf], tl_cur->d_planes, m->plane_stride, tl_cur->d_rgb, m->rgb_stride, d->width,
                        d->height, 1, tl_cur->stream);
    HIPC(m, hipEventRecord(tl_cur->ev_consumed[buf], tl_cur->stream));
    tl_cur->consumed_valid[buf] = true;
    if (s->gpu_entropy)
      HIPC(m, hipMemcpyAsync(s->h_gpu_status, tl_cur->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, tl_cur->stream));
    HIPC(m, hipMemcpyAsync(rgb, tl_cur->d_rgb, bytes, hipMemcpyDeviceToHost, tl_cur->stream));
    HIPC(m, hipStreamSynchronize(tl_cur->stream));
    if (s->gpu_entropy && s->h_gpu_status[0]) return m->fail(UFD_E_DECODE, "corrupt JPEG");
    return UFD_OK;
  });
}

int ufd_debug_preproc_rgb(ufd_model* m, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t pitch, float* out_nchw) {
  return guarded(m, [&]() -> int {
    drain_worker0(m);
    if (!out_nchw) return m->fail(UFD_E_ARG, "null argument");
    int rc = upload_rgb(m, rgb, w, h, pitch, 1);
    if (rc) return rc;
    if ((int)w == m->W && (int)h == m->H) {
      launch_norm_only(tl_cur->d_rgb, w, h, w * 3, m->rgb_stride, m->d_lut, tl_cur->d_input, 1, tl_cur->stream);
    } else {
      ResizeTaps v, hz;
      rc = get_taps(m, w, h, &v, &hz);
      if (rc) return rc;
      launch_resize_norm(tl_cur->d_rgb, w, h, w * 3, m->rgb_stride, v, hz, m->d_lut, tl_cur->d_input, m->W, m->H, 1, tl_cur->stream);
    }
    HIPC(m, hipMemcpyAsync(out_nchw, tl_cur->d_input, sizeof(float) * 3 * m->W * m->H, hipMemcpyDeviceToHost, tl_cur->stream));
    HIPC(m, hipStreamSynchronize(tl_cur->stream));
    return UFD_OK;
  });
}

int ufd_debug_forward(ufd_model* m, const float* input_nchw, uint32_t count, float* scores, float* boxes) {
  return guarded(m, [&]() -> int {
    drain_worker0(m);
    if (!input_nchw || !scores || !boxes) return m->fail(UFD_E_ARG, "null argument");
    if (count < 1 || count > m->B) return m->fail(UFD_E_TOO_LARGE, "count must be in 1..max_batch");
    const size_t in_floats = (size_t)count * 3 * m->W * m->H;
    HIPC(m, hipMemcpyAsync(tl_cur->d_input, input_nchw, in_floats * sizeof(float), hipMemcpyHostToDevice, tl_cur->stream));
    enqueue_forward(m, count);
    enqueue_heads(m, count, /*raw_outputs=*/true);
    // (no k_sort_nms follows on this tap: put the candidate counters back to zero here)
    HIPC(m, hipMemsetAsync(tl_cur->d_counts, 0, sizeof(uint32_t) * count, tl_cur->stream));
    HIPC(m, hipMemcpyAsync(scores, tl_cur->d_scores, sizeof(float) * 2 * m->K * count, hipMemcpyDeviceToHost, tl_cur->stream));
    HIPC(m, hipMemcpyAsync(boxes, tl_cur->d_boxes, sizeof(float) * 4 * m->K * count, hipMemcpyDeviceToHost, tl_cur->stream));
    HIPC(m, hipStreamSynchronize(tl_cur->stream));
    prof_flush(m);
    return UFD_OK;
  });
}

int ufd_debug_layer_output(ufd_model* m, uint32_t layer, uint32_t frame, float* out, size_t cap_floats, size_t* floats) {
  return guarded(m, [&]() -> int {
    drain_worker0(m);
    const bool taps = !m->tap_buf.empty();
    if (!(m->cfg.flags & UFD_FLAG_KEEP_LAYERS) && !taps) return m->fail(UFD_E_STATE, "needs UFD_FLAG_KEEP_LAYERS or UFD_FLAG_TAP_LAYERS");
    if (layer >= (uint32_t)kNumConv || frame >= tl_cur->last_forward_count) return m->fail(UFD_E_ARG, "layer/frame out of range");
    const Layer& L = m->layers[layer];
    if (L.tap_tensor < 0) return m->fail(UFD_E_STATE, "this layer's output never exists in the issued plan (fused into the next launch)");
    const Tensor& t = m->tensors[L.tap_tensor];
    const size_t plane = (size_t)L.oh * L.ow, nf = (size_t)L.spec.cout * plane;
    if (floats) *floats = nf;
    if (!out || cap_floats < nf) return m->fail(UFD_E_ARG, "output buffer too small");
    const float* base = taps ? m->tap_buf[L.tap_tensor] : tensor_ptr(m, L.tap_tensor);
    const float* src = base + ((size_t)frame * t.c + L.tap_coff) * plane;
    HIPC(m, hipMemcpyAsync(out, src, nf * sizeof(float), hipMemcpyDeviceToHost, tl_cur->stream));
    HIPC(m, hipStreamSynchronize(tl_cur->stream));
    return UFD_OK;
  });
}

int ufd_debug_postproc(ufd_model* m, const float* scores, const float* boxes, uint32_t count, ufd_det* out, uint32_t cap,
                       uint32_t* n) {
  return guarded(m, [&]() -> int {
    drain_worker0(m);
    int rc = check_outputs(m, out, cap, n);
    if (rc) return rc;
    if (!scores || !boxes) return m->fail(UFD_E_ARG, "null argument");
    if (count < 1 || count > m->B) return m->fail(UFD_E_TOO_LARGE, "count must be in 1..max_batch");
    Slot* s = find_free_slot(m);
    if (!s) return m->fail(UFD_E_STATE, "all slots busy");
    rc = alloc_slot(m, *s);
    if (rc) return rc;
    HIPC(m, hipMemcpyAsync(tl_cur->d_scores, scores, sizeof(float) * 2 * m->K * count, hipMemcpyHostToDevice, tl_cur->stream));
    HIPC(m, hipMemcpyAsync(tl_cur->d_boxes, boxes, sizeof(float) * 4 * m->K * count, hipMemcpyHostToDevice, tl_cur->stream));
    launch_threshold(tl_cur->d_scores, m->K, count, m->cfg.min_confidence, tl_cur->d_keys, m->key_stride, tl_cur->d_counts, tl_cur->stream);
    enqueue_nms(m, *s, count);
    s->count = count, s->cap = cap, s->out = out, s->n = n, s->status = nullptr;
    s->gpu_entropy = false;
    std::fill(s->st.begin(), s->st.begin() + count, UFD_OK);
    rc = enqueue_results_copy(m, *s, count);
    if (rc) return rc;
    return finish_slot(m, *s);
  });
}

int ufd_debug_jpeg_coefficients(const uint8_t* jpeg, size_t len, int16_t* coef, size_t cap_i16, uint32_t* n_i16,
                                uint32_t* w, uint32_t* h) {
  try {
    if (!jpeg || !len) return UFD_E_ARG;
    JpegFrameDesc d;
    int st = jpeg_parse_header(jpeg, len, &d);
    if (st != kJpegOk) return st == kJpegCorrupt ? UFD_E_DECODE : UFD_E_UNSUPPORTED;
    if (w) *w = d.width;
    if (h) *h = d.height;
    if (n_i16) *n_i16 = d.coef_total;
    if (!coef) return UFD_OK;
    if (cap_i16 < d.coef_total) return UFD_E_ARG;
    st = jpeg_decode_coefficients(jpeg, len, &d, coef, cap_i16);
    return st == kJpegOk ? UFD_OK : (st == kJpegCorrupt ? UFD_E_DECODE : UFD_E_UNSUPPORTED);
  } catch (...) {
    return UFD_E_DEVICE;
  }
}

int ufd_debug_plan(uint32_t variant, uint32_t max_batch, uint32_t flags, ufd_plan_layer* layers, uint32_t layer_cap, uint32_t* n_layers,
                   ufd_plan_tensor* tensors, uint32_t tensor_cap, uint32_t* n_tensors, uint64_t* arena_floats) {
  if ((variant != 640 && variant != 320) || !max_batch || !n_layers || !n_tensors) return UFD_E_ARG;
  try {
    std::unique_ptr<ufd_model> m(new ufd_model());
    m->cfg.variant = variant, m->cfg.flags = flags, m->cfg.max_batch = max_batch;
    m->W = variant == 640 ? 640 : 320, m->H = variant == 640 ? 480 : 240;
    m->B = max_batch;
    plan_tensors(m.get(), (flags & UFD_FLAG_KEEP_LAYERS) != 0);
    *n_layers = (uint32_t)m->layers.size(), *n_tensors = (uint32_t)m->tensors.size();
    if (arena_floats) *arena_floats = m->arena_floats;
    for (uint32_t i = 0; i < *n_layers && i < layer_cap && layers; i++) {
      const Layer& L = m->layers[i];
      ufd_plan_layer& o = layers[i];
      std::memset(&o, 0, sizeof(o));
      std::snprintf(o.name, sizeof(o.name), "%s", L.spec.name);
      o.kind = (int32_t)L.kind, o.leader = L.leader, o.ride = L.ride, o.chain_first = L.chain_first, o.fused_dw = L.fused_dw;
      o.chained = L.chained ? 1 : 0, o.materialize = L.materialize ? 1 : 0;
      o.in_tensor = L.in_tensor, o.out_tensor = L.out_tensor, o.out_coff = L.out_coff, o.tap_tensor = L.tap_tensor;
      // issues a launch of its own at its turn: not computed inside another launch, not a non-leading member, not a rider
      o.launches = !(L.kind == kKindFusedAway && !L.materialize) && !L.chained && L.leader == (int)i && L.ride < 0;
      if ((int)i == kRfbShortcut && m->rfb_tail) o.rfb_tail = 1;
    }
    for (uint32_t t = 0; t < *n_tensors && t < tensor_cap && tensors; t++) {
      const Tensor& T = m->tensors[t];
      tensors[t] = ufd_plan_tensor{(uint64_t)T.off, (uint64_t)T.per_frame() * max_batch, T.c, T.h, T.w, T.first, T.last, T.stored ? 1 : 0};
    }
    return UFD_OK;
  } catch (...) {
    return UFD_E_DEVICE;
  }
}

int ufd_debug_load_onnx(const char* path, uint32_t variant, float* weights, size_t weights_cap, float* priors,
                        size_t priors_cap, uint32_t* priors_found, char* err, size_t err_cap) {
  try {
    if (!path || !weights || (variant != 640 && variant != 320)) return UFD_E_ARG;
    std::vector<float> blob, pri;
    std::string why;
    const int W = variant == 640 ? 640 : 320, H = variant == 640 ? 480 : 240;
    if (!load_ultraface_onnx(path, W, H, &blob, &pri, &why)) {
      if (err && err_cap) std::snprintf(err, err_cap, "%s", why.c_str());
      return UFD_E_WEIGHTS;
    }
    if (weights_cap < blob.size()) return UFD_E_ARG;
    std::memcpy(weights, blob.data(), blob.size() * sizeof(float));
    if (priors_found) *priors_found = pri.empty() ? 0 : 1;
    if (!pri.empty() && priors) {
      if (priors_cap < pri.size()) return UFD_E_ARG;
      std::memcpy(priors, pri.data(), pri.size() * sizeof(float));
    }
    return UFD_OK;
  } catch (...) {
    return UFD_E_DEVICE;
  }
}

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

// ---------------------------------------------------------------- hooks of replicas.cpp (not part of the C ABI)
namespace ufd {
int create_handle(const ufd_config* cfg, ufd_model** out) { return create(cfg, out); }
void set_create_error(const std::string& msg) { g_create_error = msg; }
std::string get_create_error() { return g_create_error; }
void weight_buffers(ufd_model* m, float** d_weights, size_t* weight_floats, float** d_priors, size_t* prior_floats) {
  *d_weights = m->d_weights, *weight_floats = m->weight_img_floats;
  *d_priors = m->d_priors, *prior_floats = m->priors_floats;
}
bool load_weights_once(const ufd_config* cfg, std::vector<float>* blob, std::vector<float>* priors, std::string* why) {
  const int W = cfg->variant == 640 ? 640 : 320, H = cfg->variant == 640 ? 480 : 240;
  if (cfg->weights) {
    if (cfg->weights_floats != total_weight_floats()) {
      *why = "weights blob must hold " + std::to_string(total_weight_floats()) + " floats";
      return false;
    }
    blob->assign(cfg->weights, cfg->weights + cfg->weights_floats);
    if (cfg->priors) priors->assign(cfg->priors, cfg->priors + cfg->priors_floats);
  } else {
    const std::string path = cfg->weights_path ? cfg->weights_path : default_weights_path(cfg->variant);
    if (!load_ultraface_onnx(path, W, H, blob, priors, why)) {
      *why = "cannot load " + path + ": " + *why;
      return false;
    }
  }
  if (priors->empty()) gen_priors(W, H, *priors);
  return true;
}
}  // namespace ufd

