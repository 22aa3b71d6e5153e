// annotate.cpp -- N1: the rest of the Inferer::run iteration behind NMS (inferer.rs:38-46) for a decoded batch --
// the context's encoder scratch and (quality, framing) set-ups, rectangles + labels + baseline JPEG re-encode on the batch's
// own stream, and the finished streams' way into the caller's buffer.  Split from model.cpp in round 5.
#include "model_types.hpp"
#include "model_internal.hpp"
#include "model_parts.hpp"

using namespace ufd;

namespace {
// packed streams of an annotate batch (16-byte aligned pieces, *total bytes in all) -> the caller's pinned buffer of `cap` bytes
__global__ __launch_bounds__(256) void k_fetch_streams(const uint4* __restrict__ src, const uint32_t* __restrict__ total, uint4* __restrict__ dst,
                                                       uint32_t cap16) {
  const uint32_t n16 = min((*total + 15u) >> 4, cap16);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

namespace ufd {

// N1: the finished streams of the slot's batch -> the caller's buffer: which frames hand theirs out, and where they lie.
// The bytes are there already when the buffer is pinned (k_fetch_streams at the end of the batch's chain); otherwise
// this is the second half of a two-step copy, one D2H of everything that fits.
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
  if (fit && !s.annot_fetched) {
    // (the caller's buffer is not pinned host memory: a copy on the batch's own stream, behind whatever that context has
    // queued since -- the price of pageable output)
    HIPC(m, hipMemcpyAsync(a.jpeg_out, s.d_enc_out, fit, hipMemcpyDeviceToHost, s.ctx->stream));
    HIPC(m, record_behind_copy(s.enc_copied, s.ctx->stream));
    HIPC(m, hipEventSynchronize(s.enc_copied));
  } else if (fit > (a.jpeg_cap & ~(size_t)15)) {
    // k_fetch_streams writes whole 16-byte pieces only: a buffer whose size is no multiple of 16 and whose last stream ends in
    // the ragged tail gets those 1..15 bytes (the EOI marker or the multipart trailer) by a copy of their own.
    const size_t done = a.jpeg_cap & ~(size_t)15;
    HIPC(m, hipMemcpyAsync(a.jpeg_out + done, s.d_enc_out + done, fit - done, hipMemcpyDeviceToHost, s.ctx->stream));
    HIPC(m, record_behind_copy(s.enc_copied, s.ctx->stream));
    HIPC(m, hipEventSynchronize(s.enc_copied));
  }
  return rc;
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
  s.annot_fetched = false;
  if (s.annot_args.jpeg_out && s.annot_args.jpeg_cap >= 16) {
    hipPointerAttribute_t at;
    std::memset(&at, 0, sizeof(at));
    // (pinned host memory this handle's GPU can write: allocated portable -- ufd_host_alloc, ufd_model_host_alloc -- or on this device)
    if (hipPointerGetAttributes(&at, s.annot_args.jpeg_out) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer &&
        (reinterpret_cast<uintptr_t>(at.devicePointer) & 15) == 0 &&
        ((at.allocationFlags & hipHostMallocPortable) || at.device == m->cfg.device_id)) {
      const uint32_t cap16 = (uint32_t)std::min<size_t>(s.annot_args.jpeg_cap >> 4, 0xFFFFFFFFu);
      ProfScope ps(m, "d2h_streams", 0, 0);
      ufd_launch(k_fetch_streams, dim3(256), dim3(256), 0, c.stream, reinterpret_cast<const uint4*>(s.d_enc_out),
                         s.d_enc_meta + 2 * m->B, static_cast<uint4*>(at.devicePointer), cap16);
      s.annot_fetched = true;
    } else {
      (void)hipGetLastError();  // (pageable memory: hipPointerGetAttributes reports an error the next call must not see)
    }
  }
  return UFD_OK;
}

}  // namespace ufd

extern "C" {

size_t ufd_encode_bound(uint32_t w, uint32_t h) { return ufd::enc_frame_bound(w, h) + 64; }

int ufd_annotate_parity(uint32_t annot_flags) {
  return (annot_flags & UFD_ANNOT_NO_TEXT) ? UFD_PARITY_EXACT : UFD_PARITY_LABELS_UNPINNED;
}

}  // extern "C"
