// debug_taps.cpp -- the stage taps of the C ABI (ufd_debug_*): A1, A2-A4, A6, A7-A10 and N1 one stage at a time, on context 0
// from the calling thread, so that the parity tests can hold every stage against the oracle through the boundary itself.
// No reference counterpart; nothing here is on the product path.  (Moved out of model.cpp in round 6.)
#include "model_types.hpp"
#include "model_internal.hpp"
#include "model_parts.hpp"

using namespace ufd;

extern "C" {

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
    s->gpu_entropy = false, s->small_batch = false;
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

}  // extern "C"
