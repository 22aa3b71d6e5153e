// model_types.hpp -- the handle's state (plan, contexts, slots, workers) and the small helpers every translation unit of the
// library's host side shares: model.cpp (C ABI, per-batch pipeline, entropy / encoder / post stages) and plan.cpp (the
// network's launch plan, weight packing, the launches of the 52 convolutions).  Library-internal, not part of the C ABI.
#pragma once
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

extern thread_local std::string g_create_error;

struct Ctx;
// Per-thread view of the handle: API threads use context 0 and the handle's pool; each context's
// worker thread uses its own context and pool.
extern thread_local Ctx* tl_cur;
extern thread_local ThreadPool* tl_pool;
extern thread_local bool tl_prof;  // record kernel events for the batch being issued by this thread
extern thread_local bool tl_force_rider;  // enqueue_layer_launch: issue a riding layer on its own (its host could not take it)
struct Worker;
extern thread_local Worker* tl_worker;  // issue worker this thread is (host statistics go to it), or null on API threads
extern thread_local uint64_t tl_launches;     // launches + copies enqueued by this thread (ProfScope counts them)

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
  uint32_t* h_status_dev = nullptr;  // h_gpu_status / h_dets as device addresses (k_results_out)
  Det* h_dets_dev = nullptr;
  bool small_batch = false;         // this batch: staging block in and results out by kernels on the context's stream
  bool results_in_nms = false;      // ... and the results were written by k_sort_nms itself (no k_results_out launch)
  uint8_t* h_stage_dev = nullptr;   // h_stage as a device address (pinned host memory the GPU reads directly: k_stage_in)
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
  bool annot_fetched = false;  // the finished streams were written to the caller's pinned buffer by the batch's own chain (k_fetch_streams)
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
  uint64_t seq = 0;          // submission order over the whole handle (0: a synchronous call) -- pipeline_gate.cpp
  bool gate_published = false;
};

// Cross-context order of the GPU-filling stretch (pipeline_gate.cpp).
constexpr int kGateDefaultLayer = 8;  // m4.pw, the end of the m3->m4 launch: measured, tools/ab/r6_gate.py
struct PipelineGate {
  static constexpr int kRing = 64;  // > batches in flight (slots) by a wide margin
  int layer = -1;                   // < 0: off
  hipEvent_t ev[kRing] = {};
  std::atomic<uint64_t> published[kRing];
  std::atomic<uint64_t> waits{0}, timeouts{0};
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
  float* d_arena = nullptr;
  float* d_input = nullptr;
  JpegFrameDesc* d_descs_buf[2] = {nullptr, nullptr};  // double-buffered: copy(i+1) runs beside kernels(i)
  int16_t* d_coef_buf[2] = {nullptr, nullptr};
  // device entropy decoding: JPEG bytes, scan layouts and restart intervals of the batch
  uint8_t* d_stage_buf[2] = {nullptr, nullptr};  // device image of Slot::h_stage (d_descs_buf points at its head)
  uint8_t* d_sync = nullptr;  // scratch of the self-synchronising entropy decoder
  const JpegFrameDesc* stem_descs = nullptr;  // non-null: the next forward reads the 4:2:0 / 4:2:2 sample planes (fused stem)
  SyncBuffers sync;
  uint32_t* d_status = nullptr;
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
  int num_ctx = 4;  // one stream each = the runtime's four hardware queues (round 4: 3 contexts + a copy stream 55.5 k, 4 contexts and no copy stream 57.3 k frames/s)
  int next_ctx = 0;
  std::mutex shared_mu;  // profiling tables, resize-tap cache, Huffman table-set cache
  std::mutex err_mu;     // error string
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
  uint32_t iv_cap = 0;      // restart intervals per batch
  bool stem_fusable = false;          // layer 0 can run as k_stem_planes_mfma
  bool rfb_tail = false;              // the three dilated RFB convs + the summed 1x1 run as ONE launch (k_rfb_tail)
  bool plan_parallel = false;         // UFD_PLAN_PARALLEL=1 at ufd_create: header scan + staging copy on the pool (A/B knob, experiments build)
  uint32_t force_sub_floor = 0;       // UFD_FLAG_SUBSEQ_32 / _64: the entropy decoder's subsequence size forced (0 = the host plan's choice)
  bool gpu_entropy_enabled = true;   // device entropy kernels for baseline single-scan streams
  std::vector<float*> tap_buf;        // UFD_FLAG_TAP_LAYERS: per tensor, a copy taken right after its producing launch

  Slot slots[UFD_MAX_SLOTS];
  uint32_t next_ticket = 1;
  uint64_t next_seq = 1;   // Slot::seq of the next asynchronous batch
  PipelineGate gate;

  std::map<std::pair<int, int>, std::pair<TapsDev, TapsDev>> taps;

  // profiling
  bool profile = false;
  uint32_t prof_every = 1, prof_batch = 0;
  std::vector<std::string> prof_names;
  std::vector<ufd_kernel_stat> prof_stats;
  std::vector<LaunchShape> prof_shapes;  // by name id: shape of the label's last launch (ufd_profile_shapes)
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




#define HIPC(m, expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return (m)->fail(UFD_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));               \
  } while (0)


namespace ufd {
// ---------------------------------------------------------------- profiling (model.cpp)
int prof_name_id(ufd_model* m, const std::string& name);
hipEvent_t prof_event(ufd_model* m);

struct ProfScope {
  ufd_model* m;
  ProfEntry pe;
  bool on;
  hipStream_t st;
  ProfScope(ufd_model* mm, const std::string& name, double bytes, double flops, hipStream_t stream = nullptr)
      : m(mm), on(mm->profile && tl_prof), st(stream ? stream : tl_cur->stream) {
    tl_launches++;
    if (!on) return;
    tl_launch_shape.launches = 0;
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
    if (tl_launch_shape.launches) {  // shape of the (last) launch inside this scope: ufd_profile_shapes
      if (m->prof_shapes.size() <= (size_t)pe.name_id) m->prof_shapes.resize(pe.name_id + 1);
      m->prof_shapes[pe.name_id] = tl_launch_shape;
    }
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

// ---------------------------------------------------------------- plan.cpp
void gen_priors(int W, int H, std::vector<float>& out);
// Liveness-based arena + fusion decisions of the 52 convolutions (no HIP call: ufd_debug_plan runs it without a GPU)
void plan_tensors(ufd_model* m, bool keep_all);
int upload_weights(ufd_model* m, const float* blob);
float* tensor_ptr(ufd_model* m, int t);
ConvArgs layer_args(ufd_model* m, int i, uint32_t f0, uint32_t count, int* dw_stride);
void enqueue_layer(ufd_model* m, int i, uint32_t count);
void enqueue_layer_launch(ufd_model* m, int i, uint32_t f0, uint32_t count, hipStream_t st);
}  // namespace ufd
