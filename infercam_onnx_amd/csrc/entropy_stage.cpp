// entropy_stage.cpp -- the first stage of a batch on the issuing worker: header + marker scan of the JPEGs, their Huffman
// table sets, the staging block on its way into HBM and the device entropy chain's launches (huffman_kernels.hip) -- or,
// for progressive / multi-scan files and UFD_FLAG_HOST_ENTROPY, the host workers' Huffman decode and the coefficient
// slabs' copy.  Row A1 of SURVEY 8 (turbojpeg::decompress_image, inferer.rs:35) up to the quantised coefficients.
// (Moved out of model.cpp in round 6; state and helpers: model_types.hpp, what the files share: model_parts.hpp.)
#include "model_types.hpp"
#include "model_internal.hpp"
#include "model_parts.hpp"

namespace ufd {
namespace {
// NO COPY STREAM (round 4).  The runtime gives a process four hardware queues; a fifth stream shares one of them and its
// work is serialised with a context's kernels.  With a copy stream three compute contexts were the most that paid; without
// it there are four, and the pipeline runs 3.7 % (640x480, batch 32) to 20 % (UltraFace-320) faster.  So the staging block
// of a batch (descriptors, scan plans, intervals, JPEG bytes: 60 KB for one 640x480 frame, 1.2 MB for 32) is fetched from
// the slot's pinned host memory by a kernel ON THE CONTEXT'S STREAM: for a lone frame no transfer set-up, no fence launch
// and no event between two streams in front of the first decoder kernel (a tenth of its latency); for full batches the
// fetch is serial with the context's chain (idle gap per batch 18 -> 40 us) while the other three contexts compute.
// The way back likewise: for a batch of a few frames one launch writes the statuses, the detection counts and the
// detections each frame HAS (not 256 rows per frame) into the slot's pinned result arrays instead of two transfers, and an
// annotate batch's finished streams are written to the caller's buffer by a launch at the end of the batch's own chain
// (k_fetch_streams) when that buffer is pinned host memory (ufd_host_alloc / ufd_model_host_alloc) -- nothing is left to
// copy in ufd_wait.
constexpr size_t kStageInMaxBytes = 256 * 1024;
__global__ __launch_bounds__(256) void k_stage_in(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}
}  // namespace

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
  // Shortest subsequence: 64 bytes when the batch fills the GPU with lanes anyway (one lane per subsequence and block of
  // the MCU: 32 frames of 640x480 are 100 k lanes), 32 for a frame or a few at a time -- the launches of the chain last as
  // long as their slowest lane walks, and half the symbols per lane is 167 -> 129 us for ONE 640x480 frame (24 bytes gain
  // nothing more, 16 leave the true chain unspeculated in most frames: k_huff_resolve then decodes on the spot, 1 ms).
  static const size_t small_bytes = experiment_env("UFD_SUB_SMALL_BYTES") ? (size_t)std::atol(experiment_env("UFD_SUB_SMALL_BYTES")) : 200u * 1024;
  // (UFD_FLAG_SUBSEQ_32 / _64: the parity test that shows the floor does not change a pixel forces either)
  const uint32_t sub_floor = m->force_sub_floor ? m->force_sub_floor : (batch_bytes <= small_bytes ? 32u : 64u);
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
      sub = std::max((sub + 3u) & ~3u, sub_floor);
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
    // Timing experiments only (UFD_REPEAT_ENTROPY=n: the decoder chain n more times, same results): what the stage costs the
    // LOADED pipeline is the frame rate it takes away when it runs twice (docs/EXPERIMENTS.md, round 5).
    static const int repeat = experiment_env("UFD_REPEAT_ENTROPY") ? std::atoi(experiment_env("UFD_REPEAT_ENTROPY")) : 0;
    for (int r = 0; r < repeat; r++)
      launch_huffman_sync(d_blob, d_scans, d_ivs, count, p.max_nsub, p.max_bpm, m->d_sync_luts, d_descs, d_coef, m->coef_stride,
                          c.sync, c.d_status, c.stream, nullptr, p.used_coef);
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
  s.small_batch = false;
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
      // (stream order protects the buffer: its last readers were kernels of an earlier batch on this stream)
      s.small_batch = p.stage_bytes <= kStageInMaxBytes && s.h_stage_dev;
      {
        ProfScope ps(m, "h2d_jpeg", (double)p.stage_bytes, 0, c.stream);
        if (s.h_stage_dev) {
          const uint32_t n16 = (uint32_t)((p.stage_bytes + 15) / 16);
          ufd_launch(k_stage_in, dim3(std::min(256u, (n16 + 255) / 256)), dim3(256), 0, c.stream,
                             reinterpret_cast<const uint4*>(s.h_stage_dev), reinterpret_cast<uint4*>(c.d_stage_buf[buf]), n16);
        } else {
          HIPC(m, hipMemcpyAsync(c.d_stage_buf[buf], s.h_stage, p.stage_bytes, hipMemcpyHostToDevice, c.stream));
        }
      }
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
  // (on the context's own stream: the buffer's last readers were kernels of an earlier batch on it; the other contexts
  // compute while these 29 MB per batch of 32 cross PCIe)
  HIPC(m, hipMemcpyAsync(c.d_descs_buf[buf], s.h_descs, sizeof(JpegFrameDesc) * count, hipMemcpyHostToDevice, c.stream));
  {
    ProfScope ps(m, "h2d_coef", 0, 0, c.stream);
    // frames are equally sized in a stream: copy the used prefix of every slab in one 2-D copy
    HIPC(m, hipMemcpy2DAsync(c.d_coef_buf[buf], m->coef_stride * 2, s.h_coef, m->coef_stride * 2, used * 2, count,
                             hipMemcpyHostToDevice, c.stream));
    // (span_begin records an event: never directly behind an asynchronous copy -- ROCm 7.2's runtime keeps ~2 KB of host memory
    // per such event, record_behind_copy above)
    launch_copy_fence(c.stream);
  }
  span_begin(s);
  return UFD_OK;
}

}  // namespace ufd
