// model_parts.hpp -- what model.cpp, stats.cpp and annotate.cpp need from one another (library-internal, not part of the C
// ABI; include behind model_types.hpp, whose types the declarations use).
#pragma once
#include "model_types.hpp"

namespace ufd {
// stats.cpp: kernel profiling (events on the library's streams), device-time spans per context, host statistics
void prof_flush(ufd_model* m);
void span_begin(Slot& s);
void span_end(Slot& s);
void span_fold(ufd_model* m, Slot& s);
// annotate.cpp: N1 (inferer.rs:38-46) -- encoder scratch, rectangles + labels + re-encode of a decoded batch, the streams' way out
size_t enc_frame_bound(uint32_t w, uint32_t h);
int ensure_encoder(ufd_model* m, Ctx& c, uint32_t quality, bool multipart, uint32_t mw, uint32_t mh);
int ensure_slot_encoder(ufd_model* m, Slot& s, uint32_t mw, uint32_t mh);
int enqueue_annotate(ufd_model* m, Slot& s, const JpegFrameDesc* d_descs, uint32_t mw, uint32_t mh, uint32_t count);
int fetch_streams(ufd_model* m, Slot& s);
// model.cpp: the per-batch pipeline's pieces the stage taps (debug_taps.cpp) drive one at a time
int alloc_slot(ufd_model* m, Slot& s);
int get_taps(ufd_model* m, int sw, int sh, ResizeTaps* vert, ResizeTaps* horz);
void enqueue_forward(ufd_model* m, uint32_t count, Slot* s = nullptr);
void enqueue_heads(ufd_model* m, uint32_t count, bool raw_outputs = false);
void enqueue_nms(ufd_model* m, Slot& s, uint32_t count);
int enqueue_results_copy(ufd_model* m, Slot& s, uint32_t count);
int finish_slot(ufd_model* m, Slot& s, bool locked = true);
Slot* find_free_slot(ufd_model* m);
int check_outputs(ufd_model* m, const void* out, uint32_t cap, const void* n);
void drain_worker0(ufd_model* m);  // synchronous entry points run on context 0: wait until its worker has nothing queued
int upload_rgb(ufd_model* m, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t pitch, uint32_t count);
int run_rgb_on_device(ufd_model* m, Slot& s, uint32_t w, uint32_t h, uint32_t count);
int run_decoded(ufd_model* m, Slot& s, uint32_t count, bool any_ok, const JpegFrameDesc* d_descs, int16_t* d_coef, int buf);
// placement.cpp
void resolve_placement(ufd_model* m);
void pin_this_thread(const ufd_model* m);
// entropy_stage.cpp: what a batch does before its first reconstruction kernel
int status_from_jpeg(int st);
void pin_lut_sets(ufd_model* m, const HuffScan* scans, uint32_t count, int delta);
DevicePlan plan_device_entropy(ufd_model* m, Slot& s, const uint8_t* const* jpegs, const size_t* lens, uint32_t count);
int enqueue_device_entropy(ufd_model* m, Ctx& c, const DevicePlan& p, uint32_t count, const uint8_t* d_blob, const JpegFrameDesc* d_descs,
                           const HuffScan* d_scans, const HuffInterval* d_ivs, int16_t* d_coef);
int entropy_stage(ufd_model* m, Slot& s, const uint8_t* const* jpegs, const size_t* lens, uint32_t count, int* buf_out, bool* any_ok_out);
void launch_copy_fence(hipStream_t stream);  // model.cpp: the empty kernel that goes between a copy and an event (record_behind_copy)
// pipeline_gate.cpp: batch n + 1 starts its network behind batch n's GPU-filling stretch
int gate_init(ufd_model* m);
void gate_destroy(ufd_model* m);
void gate_wait_for_previous(ufd_model* m, const Slot& s, hipStream_t st);
void gate_pass(ufd_model* m, Slot& s, int layer, hipStream_t st);
void gate_close(ufd_model* m, Slot& s);
// model.cpp: an event that marks a copy (an empty kernel between the two: ROCm 7.2 keeps ~2 KB per event recorded directly behind a copy)
hipError_t record_behind_copy(hipEvent_t ev, hipStream_t stream);

// Synchronous entry points and stage taps: handle lock, device selection, context 0, no exception across the ABI.
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
}  // namespace ufd
