/*
 * ufd.h -- C ABI of libufacehip.so: the MI355X (gfx950) implementation of infer_server's
 * per-frame face-detection hot path
 *     JPEG decode -> Triangle resize + normalize -> UltraFace-RFB forward -> threshold + NMS.
 *
 * Every entry point is what a Rust `extern "C"` block in the reference's infer_server crate
 * would bind to replace one reference interface (cited per function; INTEGRATION.md shows the
 * binding).  Plain pointers and sizes only; the library never throws or aborts across this
 * boundary -- every failure is a negative status code plus ufd_last_error().
 *
 * Threading: a handle may be used from any thread; calls on one handle are serialised by an
 * internal lock (the reference has exactly one Inferer task, infer_server.rs:48-50).  Use one
 * handle per GPU.
 *
 * Ownership: input buffers are borrowed for the duration of the call (async form: until
 * ufd_wait returns); outputs are caller-allocated.
 */
#ifndef UFD_H
#define UFD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UFD_ABI_VERSION 1

/* ---- status codes (0 = ok; the reference uses anyhow::Result / panics, inferer.rs:23-37) ---- */
#define UFD_OK 0
#define UFD_E_ARG (-1)         /* bad argument */
#define UFD_E_DECODE (-2)      /* corrupt JPEG: frame skipped (reference: expect() panic, inferer.rs:35-36) */
#define UFD_E_UNSUPPORTED (-3) /* JPEG feature outside the decoder (arithmetic coding, 12-bit, CMYK ...) */
#define UFD_E_TRUNCATED (-4)   /* more detections than `cap`: `cap` written, *n = true count */
#define UFD_E_DEVICE (-5)      /* HIP runtime error / no gfx950 device */
#define UFD_E_WEIGHTS (-6)     /* weight file missing or not an UltraFace-RFB graph */
#define UFD_E_STATE (-7)       /* async misuse (wait without submit, slot busy, ...) */
#define UFD_E_TOO_LARGE (-8)   /* frame or batch exceeds the limits given at ufd_create */

/* Bbox + confidence: `(Bbox, f32)` with `Bbox = [f32; 4]` (infer_server/src/nn.rs:12,25).
 * Relative coordinates [x_top_left, y_top_left, x_bottom_right, y_bottom_right]. */
typedef struct ufd_det {
  float x_tl, y_tl, x_br, y_br, conf;
} ufd_det;

typedef struct ufd_model ufd_model; /* opaque: UltrafaceModel (nn.rs:45-51) resident on one GPU */

/* Arguments of UltrafaceModel::new(variant, max_iou, min_confidence) (nn.rs:55) plus placement. */
typedef struct ufd_config {
  uint32_t struct_size;    /* = sizeof(ufd_config) */
  uint32_t variant;        /* 640 -> UltrafaceVariant::W640H480, 320 -> W320H240 (nn.rs:29-42) */
  float max_iou;           /* reference passes 0.5 (inferer.rs:23) */
  float min_confidence;    /* reference passes 0.5 */
  int32_t device_id;       /* HIP device ordinal */
  uint32_t max_batch;      /* frames per batched call (1..1024; every activation tensor of a batch must stay below 4 GiB:
                              at most 873 for UltraFace-640, else UFD_E_TOO_LARGE) */
  uint32_t max_src_width;  /* largest decoded frame accepted; 0 -> 1920 */
  uint32_t max_src_height; /* 0 -> 1088 */
  uint32_t host_threads;   /* host entropy-decode workers; 0 -> min(32, hardware threads) */
  uint32_t flags;          /* UFD_FLAG_* */
  /* Weights: exactly one source.
   * (a) weights_path: an UltraFace-RFB .onnx (NULL and no blob -> the reference's cache path
   *     $XDG_CACHE_HOME|~/.cache /infercam_onnx/ultraface-RFB-{640,320}.onnx, nn.rs:144-156);
   * (b) weights/weights_floats: packed f32 blob, for each of the 52 convs w[cout][cin/g][k][k]
   *     then b[cout], BatchNorm folded (273 888 floats); priors (K*4, cx cy w h) optional. */
  const char* weights_path;
  const float* weights;
  size_t weights_floats;
  const float* priors;
  size_t priors_floats;
} ufd_config;

#define UFD_FLAG_KEEP_LAYERS 1u /* keep every conv output resident for ufd_debug_layer_output */
#define UFD_FLAG_PROFILE 2u     /* record HIP events around every kernel (ufd_profile_read) */
/* Entropy (Huffman) stage of the JPEG decode.  By default baseline single-scan streams, with or
 * without restart markers, are decoded by the GPU kernels (self-synchronising parallel decoder:
 * the host only scans headers and markers); progressive and multi-scan files are decoded by the
 * handle's host worker threads. */
#define UFD_FLAG_DEVICE_ENTROPY 4u /* accepted for compatibility: the GPU entropy kernels are the default */
#define UFD_FLAG_HOST_ENTROPY 8u   /* never use the GPU entropy kernels */
/* Parity-test instruments (no effect on results beyond what each says): */
#define UFD_FLAG_TAP_LAYERS 16u    /* the issued (fused) plan, plus a copy of every tensor it writes for ufd_debug_layer_output */
#define UFD_FLAG_NO_CHAIN 32u      /* m1->m2 / m3->m4 as two launches each instead of the chained kernel */
#define UFD_FLAG_NO_RFB_SUM 64u    /* ConvLinear and shortcut of the RFB block as two convs instead of one summed conv */
#define UFD_FLAG_NO_STEM_FUSE 128u /* upsample/colour/normalise kernel + stem conv instead of the stem reading the sample planes */
#define UFD_FLAG_NO_DUAL 512u      /* head pairs and the backbone block beside them as separate launches instead of one grid */
#define UFD_FLAG_NO_RFB_TAIL 2048u /* the RFB's dilated convs and its summed 1x1 as two launches (concat tensor in memory) instead of k_rfb_tail */
#define UFD_FLAG_SUBSEQ_32 4096u   /* device entropy decoder: 32-byte subsequences whatever the batch holds (the host plan picks 32 for a */
#define UFD_FLAG_SUBSEQ_64 8192u   /* frame or a few, 64 otherwise; same coefficients either way) ... or 64-byte ones */
#define UFD_FLAG_NO_GATE 32768u    /* no cross-context order of the network's GPU-filling stretch (csrc/pipeline_gate.cpp); same results */
#define UFD_FLAG_TEST_DUPLICATE_DEVICES 16384u /* ufd_create_replicas only: let one device be listed twice (a one-GPU box rehearsing n = 2) */
/* Host placement: by default the handle's issue workers and pool threads are pinned to the CPUs of the NUMA node the GPU
 * hangs off (/sys/bus/pci/devices/<bdf>/numa_node), inside the process's affinity mask -- eight handles on a two-socket
 * box then stage their JPEG bytes and issue their launches from the socket next to their GPU (ufd_model_placement
 * reports what was resolved).  Nothing is pinned when the node is unknown or with this flag. */
#define UFD_FLAG_NO_NUMA_PIN 256u
/* ufd_wait with other batches of the handle still in flight sleeps between polls of the batch's event (a pipelined
 * caller loses nothing by waking a few tens of microseconds late, and a spinning waiter would hold a CPU of the shared
 * host); with this flag it always spins in the runtime.  A lone batch (the latency form) always spins. */
#define UFD_FLAG_SPIN_WAIT 1024u

/* UltrafaceModel::new (nn.rs:55-67) + get_model (nn.rs:143-175): load + pack weights into HBM. */
int ufd_create(const ufd_config* cfg, ufd_model** out);
/* (Since round 6 a handle's streams are created at the highest stream priority, which by itself makes the order described here
 * irrelevant -- 65.5 k frames/s either way; this call remains for hosts that want the order right as well.)
 * For processes that hold ANOTHER copy of the HIP runtime (a Python host with torch: its wheels bundle one): call this before
 * that runtime first touches the device.  A handle's four streams are this runtime's four hardware queues; when the other
 * runtime opens its queue FIRST -- one 32-byte copy by torch is enough -- every handle created afterwards runs 21 % slower
 * for the life of the process (65.4 k -> 51.5 k frames/s at batch 32), while the same activity after this call, or after a
 * handle exists, costs nothing (round 6: tools/ab/r6_torch_queue.py, profiles/r6e/torch_queue*.txt).  Opens four streams on
 * the device, runs an empty kernel on each, closes them.  (The other runtime must have been INITIALISED before -- for torch:
 * torch.cuda.set_device(), which opens no queue --; initialised behind this call it no longer finds the device.)  No
 * reference counterpart (a Rust host links one runtime). */
int ufd_prime_device(int32_t device_id);
void ufd_destroy(ufd_model* m);
/* Multi-GPU start-up in ONE process -- the reference server is one process whose tasks share one model
 * (infer_server.rs:39-68; the single Inferer is spawned at :48-50).  Streams shard one-per-GPU (independent: run(&self)
 * is pure, nn.rs:178-186), so a host keeps one handle per GPU and N Inferer tasks.  This call
 *   - reads the weight source of `cfg` ONCE (get_model, nn.rs:143-175: one parse of the .onnx, not one per GPU),
 *   - creates the handle of device_ids[0] from it,
 *   - creates the other handles empty and fills them by the path's ONLY collective: ncclBroadcast (RCCL over xGMI,
 *     communicators from ncclCommInitAll) of the packed weight image and the priors from device_ids[0].
 * cfg->device_id is ignored; out[n] receives the handles in device_ids order (destroy each with ufd_destroy).  RCCL is
 * loaded on first use (dlopen): the single-GPU entry points do not depend on it.  On failure out[] is all NULL and
 * ufd_last_error(NULL) says why: UFD_E_ARG (n == 0, n > UFD_MAX_REPLICAS, an id out of range or listed twice),
 * UFD_E_WEIGHTS, UFD_E_DEVICE (no GPU, librccl missing, a RCCL or HIP error). */
#define UFD_MAX_REPLICAS 64
int ufd_create_replicas(const ufd_config* cfg, const int32_t* device_ids, uint32_t n, ufd_model** out);
/* Message for the last failing call on this handle (or on creation when m == NULL). */
const char* ufd_last_error(const ufd_model* m);
/* UltrafaceVariant::width_height (nn.rs:36-41) and K (number of priors) of the loaded model. */
int ufd_model_info(const ufd_model* m, uint32_t* width, uint32_t* height, uint32_t* num_priors);
/* Where the handle lives: HIP device, PCI address ("0000:63:00.0"), NUMA node of that PCI device (-1: unknown) and the
 * CPUs its host threads are pinned to (count, and a "0-15,128-143" list; 0 / "" when nothing was pinned).  Any output may be NULL. */
int ufd_model_placement(const ufd_model* m, int32_t* device_id, int32_t* numa_node, uint32_t* pinned_cpus, char* pci_bdf,
                        size_t pci_cap, char* cpu_list, size_t cpu_cap);
/* Limits given at ufd_create (defaults resolved): frames per batch, largest decoded frame. */
int ufd_model_limits(const ufd_model* m, uint32_t* max_batch, uint32_t* max_src_width, uint32_t* max_src_height);

/* InferModel::run(&self, input: &RgbImage) -> Result<Vec<(Bbox, f32)>> (nn.rs:24-26,178-186).
 * rgb: interleaved RGB8, `pitch` bytes per row (>= 3*w), any w x h up to the create-time limit.
 * out[cap] receives detections in descending confidence; *n = number found. */
int ufd_infer_rgb(ufd_model* m, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t pitch, ufd_det* out,
                  uint32_t cap, uint32_t* n);

/* Inferer::run lines inferer.rs:35-37: turbojpeg::decompress_image(jpeg) then infer_faces(&image).
 * img_w/img_h (optional) receive the JPEG's own frame size. */
int ufd_infer_jpeg(ufd_model* m, const uint8_t* jpeg, size_t len, ufd_det* out, uint32_t cap, uint32_t* n,
                   uint32_t* img_w, uint32_t* img_h);

/* The same step for `count` (<= max_batch) independent frames in one pass (one GPU worker
 * replacing the single Inferer task, inferer.rs:29-50).  out: count*cap entries, frame i at
 * out + i*cap; n[count]; status[count] per-frame status (a corrupt frame is skipped, the
 * others still run).  Returns UFD_OK if the batch ran (inspect status[]), <0 if nothing ran. */
int ufd_infer_jpeg_batch(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, uint32_t count,
                         ufd_det* out, uint32_t cap, uint32_t* n, int32_t* status);

/* Same for `count` equally sized RGB frames laid out back to back (frame stride = h*pitch). */
int ufd_infer_rgb_batch(ufd_model* m, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t pitch, uint32_t count,
                        ufd_det* out, uint32_t cap, uint32_t* n);

/* Asynchronous form (the 10-slot StaticImage ring of lib.rs:32-37 becomes `slots` in-flight
 * batches): ufd_submit_jpeg_batch only QUEUES the batch for one of the handle's four issue workers and returns a
 * ticket -- the worker scans the JPEG headers, copies the bytes into its pinned staging block and enqueues the GPU work
 * on its context's stream later, so nothing of the inputs has been read when the call returns; ufd_wait blocks until
 * that batch is done and fills the outputs given at submit.  Input and output buffers must stay valid until then.
 * The handle runs four device contexts in rotation (one stream each: the runtime's four hardware queues): keep six to
 * eight batches in flight (six is what bench.py uses) for full throughput; one at a time is the lowest-latency form. */
#define UFD_MAX_SLOTS 8
int ufd_submit_jpeg_batch(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, uint32_t count,
                          ufd_det* out, uint32_t cap, uint32_t* n, int32_t* status, uint32_t* ticket);
int ufd_wait(ufd_model* m, uint32_t ticket);

/* Device-resident input.  ufd_stage_jpeg_batch parses the headers of `count` JPEGs on the host
 * and places their bytes, frame descriptors and scan plans in HBM (blocking; UFD_E_STATE with
 * UFD_FLAG_HOST_ENTROPY; UFD_E_UNSUPPORTED if a frame cannot take the device entropy decoder).
 * ufd_submit_staged then runs the whole path -- entropy decode, IDCT, upsampling, colour,
 * normalisation, the network, NMS -- from those HBM buffers: no input crosses PCIe, only the
 * detections come back.  Same ticket / ufd_wait protocol and outputs as ufd_submit_jpeg_batch
 * (the reference has no such call: its ring slots are host Vec<u8>, lib.rs:32; this is the form a
 * capture card or NIC writing straight into GPU memory would use; bench.py times the host-bytes form above by
 * default and reports this one beside it, `--input hbm` / `config.hbm_resident_fps`).
 * A staged batch may be submitted any number of times; free it after its last ufd_wait. */
typedef struct ufd_staged ufd_staged;
int ufd_stage_jpeg_batch(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, uint32_t count, ufd_staged** staged);
int ufd_submit_staged(ufd_model* m, const ufd_staged* staged, ufd_det* out, uint32_t cap, uint32_t* n, int32_t* status,
                      uint32_t* ticket);
void ufd_staged_free(ufd_model* m, ufd_staged* staged);

/* ---- N1: the rest of the Inferer::run iteration (inferer.rs:38-46) on the frame that is already decoded in HBM ----
 * draw_bboxes_on_image(image, boxes, width, height) (inferer.rs:58-92): one hollow rectangle per detection at
 * bbox * (label_width, label_height) with the reference's `as i32` / `as u32` casts, colour (0, 255, 0), clipped
 * to the decoded frame.  NOTE: label_width / label_height are the slot's StaticImage.0 / .1, which the router stamps
 * with 1280 x 720 whatever the JPEG's own size is (router.rs:66-67).  A rectangle narrower or lower than one pixel
 * (imageproc's Rect::of_size would assert and the reference task panic) is skipped, label included.  Then the
 * confidence label of that detection, draw_text(.., x as i32, y as i32, Scale 16, DejaVuSansMono, "{:.2}%")
 * (inferer.rs:80-88): glyph coverage from a table restated from rusttype 0.9.3 / ab_glyph_rasterizer (parity with the
 * crates unpinned: DESIGN.md section 2), blended with imageproc's f32 weighted_sum; detections are drawn in order, so
 * later rectangles and labels go over earlier ones exactly as in the reference.  UFD_ANNOT_NO_TEXT: rectangles only.
 * turbojpeg::compress_image(&frame, quality, Subsamp::Sub2x2) (inferer.rs:39, quality 95): baseline 4:2:0 stream
 * byte-identical to libjpeg-turbo's (fast integer DCT below quality 96 as tjCompress2 selects, accurate from 96).
 * UFD_ANNOT_MULTIPART wraps every stream as as_jpeg_stream_item does (lib.rs:48-57).
 *
 * The streams of a batch are written back to back into jpeg_out (each starts on a 16-byte boundary):
 * frame i = jpeg_out[jpeg_off[i] .. jpeg_off[i] + jpeg_len[i]); a frame that failed to decode has length 0; a frame
 * whose stream does not fit in jpeg_cap gets status UFD_E_TRUNCATED (its detections are still valid) and length 0.
 * ufd_encode_bound(w, h) bytes per frame always suffice.  Detections, n[] and status[] as ufd_submit_jpeg_batch. */
#define UFD_ANNOT_MULTIPART 1u
#define UFD_ANNOT_NO_TEXT 2u
/* What is pinned about the annotated pixels.  Rectangles (first-party arithmetic, inferer.rs:66-76),
 * the blend (imageproc weighted_sum) and the encoder (byte-identical to libjpeg-turbo) are pinned; the label's GLYPH
 * COVERAGE values are restated from rusttype 0.9.3 / ab_glyph_rasterizer and could not be checked against the crates
 * (FreeType agrees on ink boxes within a pixel and on ink within 12 %: tests).  UFD_ANNOT_NO_TEXT is the exact mode. */
#define UFD_PARITY_EXACT 0
#define UFD_PARITY_LABELS_UNPINNED 1
int ufd_annotate_parity(uint32_t annot_flags);
typedef struct ufd_annotate {
  uint32_t struct_size;   /* = sizeof(ufd_annotate) */
  float label_width;      /* inferer.rs:32-33: recv_ref.0 / .1 (1280 / 720 in the reference's router) */
  float label_height;
  uint32_t quality;       /* inferer.rs:39 passes 95 */
  uint32_t flags;         /* UFD_ANNOT_* */
  uint8_t* jpeg_out;      /* output for the whole batch; into pinned memory (ufd_host_alloc) the batch's own chain writes the
                           * streams, nothing is left to copy in ufd_wait; ordinary memory is filled by a copy there */
  size_t jpeg_cap;
  size_t* jpeg_off;       /* [count] */
  size_t* jpeg_len;       /* [count] */
} ufd_annotate;
int ufd_submit_annotate_batch(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, uint32_t count,
                              const ufd_annotate* annot, ufd_det* out, uint32_t cap, uint32_t* n, int32_t* status,
                              uint32_t* ticket);
/* submit + wait */
int ufd_annotate_jpeg_batch(ufd_model* m, const uint8_t* const* jpegs, const size_t* lens, uint32_t count,
                            const ufd_annotate* annot, ufd_det* out, uint32_t cap, uint32_t* n, int32_t* status);
/* Worst-case bytes of one finished stream of a w x h frame (header, byte stuffing and framing included). */
size_t ufd_encode_bound(uint32_t w, uint32_t h);
/* Page-locked host memory for jpeg_out (and for input rings): the D2H copy then runs at PCIe speed. */
void* ufd_host_alloc(size_t bytes);
void ufd_host_free(void* p);
/* The same, allocated with the handle's GPU current (on a multi-GPU box a bare allocation would open a context on device 0
 * from every rank) and visible to every device; free with ufd_host_free. */
void* ufd_model_host_alloc(ufd_model* m, size_t bytes);

/* ---- N4: multi-stream batching scheduler ----
 * The FrameRouter -> INFER_IMAGES_CHANNEL -> Inferer leg of the reference (router.rs:64-71, lib.rs:32-37,
 * inferer.rs:29-50) for MANY camera streams per GPU.  The reference serialises every stream through one task and one
 * fixed model (UltraFace-320, inferer.rs:23); here every stream is bound to a variant (320 / 640) and to what it wants
 * back (detections, or detections + the annotated JPEG of N1), frames are copied into the stream's own ring
 * (drop-on-full exactly like `infer_tx.try_send_ref()`, router.rs:65: a full ring drops the NEW frame), and a
 * dispatcher thread forms batches per (variant, output kind):
 *   - fairness: one frame per stream per pass, round-robin, starting after the stream served last -- a 30 fps camera
 *     cannot starve a 1 fps one;
 *   - a batch leaves when it is full, when its oldest frame has waited max_wait_us, or at once when the model has
 *     nothing in flight (a lone frame never waits for company);
 *   - up to max_inflight batches per model in flight (the handle overlaps them on its four device contexts).
 * Results are delivered through on_result, per frame, in dispatch order per stream, on the completion thread of the
 * stream's replica: with several replicas on_result is called from several threads at once (never for one stream). */
#define UFD_E_FULL (-9) /* ufd_sched_push: the stream's ring is full, the frame was dropped (router.rs:65) */
#define UFD_SCHED_NO_WAIT 0xFFFFFFFFu
typedef struct ufd_sched ufd_sched;
typedef struct ufd_frame_result {
  uint64_t stream_id;  /* as given in ufd_stream_config (e.g. lib.rs:39-46 hashed(name)) */
  uint64_t tag;        /* as given to ufd_sched_push */
  int32_t status;      /* UFD_OK, UFD_E_DECODE / UFD_E_UNSUPPORTED / UFD_E_TOO_LARGE (frame skipped), UFD_E_TRUNCATED */
  uint32_t variant;    /* 320 / 640 */
  uint32_t n;          /* detections found; min(n, det_cap) entries in dets */
  uint32_t batch_fill; /* frames in the batch this one travelled in */
  const ufd_det* dets; /* valid during the callback only */
  const uint8_t* jpeg; /* annotated stream (NULL for a detections-only stream or a skipped frame); callback only */
  size_t jpeg_len;
  double queue_ms;     /* push -> batch dispatched */
  double total_ms;     /* push -> this callback */
  uint32_t replica;    /* replica of the variant (= position in ufd_sched_config.models_*) the frame ran on */
} ufd_frame_result;
typedef void (*ufd_result_fn)(void* user, const ufd_frame_result* result);
#define UFD_SCHED_PLACE_ROUND_ROBIN 0u  /* stream i (in order of arrival, per variant) -> replica i mod n */
#define UFD_SCHED_PLACE_LEAST_LOADED 1u /* -> the replica with the fewest live streams */
typedef struct ufd_sched_config {
  uint32_t struct_size;      /* = sizeof(ufd_sched_config) */
  ufd_model* model_320;      /* handles the scheduler submits to (not owned; either may be NULL): the one-GPU shorthand */
  ufd_model* model_640;
  /* The N-GPU form -- ONE scheduler over the handles ufd_create_replicas returns, one per GPU: a stream is placed on a
   * replica when it is added and lives there; batches are formed per replica, each replica keeps up to max_inflight of
   * them in flight and has its own completion thread.  A variant takes either its model_* or its models_* (not both). */
  ufd_model* const* models_320;
  uint32_t n_320;
  ufd_model* const* models_640;
  uint32_t n_640;
  uint32_t placement;        /* UFD_SCHED_PLACE_* */
  uint32_t ring_slots;       /* frames a stream may have queued; 0 -> 10 (INFER_IMAGES_CHANNEL, lib.rs:37) */
  uint32_t max_wait_us;      /* 0 -> 2000; UFD_SCHED_NO_WAIT: a batch leaves as soon as the model has a slot for it */
  uint32_t max_inflight;     /* per replica; 0 -> 6 */
  uint32_t det_cap;          /* 0 -> 256 */
  uint32_t jpeg_bytes_per_frame; /* output reserved per annotated frame; 0 -> 524288 (larger streams: UFD_E_TRUNCATED) */
  ufd_result_fn on_result;
  void* user;
} ufd_sched_config;
typedef struct ufd_stream_config {
  uint32_t struct_size;  /* = sizeof(ufd_stream_config) */
  uint64_t stream_id;
  uint32_t variant;      /* 320 / 640 */
  uint32_t annotate;     /* 0: detections only; 1: + annotated JPEG (ufd_submit_annotate_batch) */
  float label_width;     /* annotate: as ufd_annotate */
  float label_height;
  uint32_t quality;      /* annotate; 0 -> 95 */
  uint32_t flags;        /* annotate: UFD_ANNOT_* */
  uint32_t replica;      /* 0: placed by the scheduler (ufd_sched_config.placement); r + 1: replica r of the variant */
} ufd_stream_config;
typedef struct ufd_sched_stats {
  uint64_t pushed, dropped, delivered, batches, frames_in_batches;
  uint64_t sent_full, sent_deadline, sent_idle; /* why batches left */
} ufd_sched_stats;
int ufd_sched_create(const ufd_sched_config* cfg, ufd_sched** out);
/* Delivers everything queued, then stops the threads. */
void ufd_sched_destroy(ufd_sched* s);
/* *stream receives the stream's handle.  A removed stream still delivers what it had queued; once that is out its ring
 * slots are freed and its table entry is reused by a later ufd_sched_add_stream (cameras that reconnect do not grow the
 * scheduler); the old handle then fails with UFD_E_STATE / UFD_E_ARG instead of reaching the new stream. */
int ufd_sched_add_stream(ufd_sched* s, const ufd_stream_config* cfg, uint32_t* stream);
int ufd_sched_remove_stream(ufd_sched* s, uint32_t stream);
/* Table entries in use (live + draining streams) and ever allocated: flat under add / remove churn. */
int ufd_sched_debug_table(ufd_sched* s, uint32_t* live, uint32_t* allocated);
/* router.rs:64-71: copies the JPEG into a free slot of the stream's ring (the caller's buffer is free on return);
 * UFD_E_FULL when there is none. */
int ufd_sched_push(ufd_sched* s, uint32_t stream, const uint8_t* jpeg, size_t len, uint64_t tag);
/* The same for `count` frames of one stream in one call (a router thread that drained several frames off its socket):
 * they are taken in order while the ring has free slots, *accepted (optional) = how many were queued, the rest are
 * dropped exactly as single pushes would be (UFD_E_FULL unless all were accepted).  tags may be NULL (all 0).
 * ufd_sched_stats counts every frame a call did not take as `dropped` (and `pushed`) -- per ATTEMPT: a caller that
 * offers the unaccepted tail again (bench.py --one-process does) counts such a frame once per call, so `accepted` and
 * `delivered` are the numbers to balance, not `pushed`. */
int ufd_sched_push_batch(ufd_sched* s, uint32_t stream, const uint8_t* const* jpegs, const size_t* lens, const uint64_t* tags,
                         uint32_t count, uint32_t* accepted);
/* Blocks until every frame pushed before the call has been delivered. */
int ufd_sched_flush(ufd_sched* s);
int ufd_sched_get_stats(ufd_sched* s, ufd_sched_stats* out);
/* Where a stream lives, and what every replica of a variant has delivered (*n = replicas of the variant; min(*n, cap)
 * entries written). */
typedef struct ufd_sched_replica_stats {
  uint32_t replica, streams, inflight, pad;
  uint64_t batches, frames; /* delivered */
  uint64_t detections;      /* found in the delivered frames */
} ufd_sched_replica_stats;
int ufd_sched_stream_replica(ufd_sched* s, uint32_t stream, uint32_t* replica);
int ufd_sched_get_replica_stats(ufd_sched* s, uint32_t variant, ufd_sched_replica_stats* out, uint32_t cap, uint32_t* n);
/* The batching rule alone (no GPU, no threads): queued[i] frames wait in stream i (streams of one batch class);
 * starting after stream `last`, one frame per stream per pass until max_batch: take[i] = frames taken from stream i.
 * Returns the batch size. */
uint32_t ufd_sched_debug_plan(const uint32_t* queued, uint32_t n_streams, uint32_t last, uint32_t max_batch, uint32_t* take);

/* ---- stage taps (parity tests call the path stage by stage through these) ---- */
/* N1 stages alone: the rectangles (and, text != 0, labels) of `n` detections on an RGB8 frame (in place), and the encoder
 * on an RGB8 frame. */
int ufd_debug_draw_labels(ufd_model* m, uint8_t* rgb, uint32_t w, uint32_t h, uint32_t pitch, const ufd_det* dets, uint32_t n,
                          float label_width, float label_height, uint32_t text);
int ufd_debug_encode_rgb(ufd_model* m, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t pitch, uint32_t quality,
                         uint32_t flags, uint8_t* out, size_t cap, size_t* len);
/* A1 only: decode on the GPU and copy the interleaved RGB8 frame back (cap_bytes >= h*w*3). */
int ufd_debug_decode_jpeg(ufd_model* m, const uint8_t* jpeg, size_t len, uint8_t* rgb, size_t cap_bytes,
                          uint32_t* w, uint32_t* h);
/* A2-A4 only: resize + normalize on the GPU; out = [3][H][W] f32 of the model size. */
int ufd_debug_preproc_rgb(ufd_model* m, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t pitch, float* out_nchw);
/* A6 only: `count` pre-normalised inputs [count][3][H][W] -> scores [count][K][2], boxes [count][K][4]. */
int ufd_debug_forward(ufd_model* m, const float* input_nchw, uint32_t count, float* scores, float* boxes);
/* Output of conv `layer` (0..51, post activation; 24 = RFB block output) for frame `frame` of the
 * last forward.  UFD_FLAG_KEEP_LAYERS: the unfused plan, every layer available.  UFD_FLAG_TAP_LAYERS:
 * the plan the product issues; layers whose output is fused into the next launch (depthwise
 * convs, m1.pw, m3.pw, rfb.linear) return UFD_E_STATE.  *floats = cout*oh*ow. */
int ufd_debug_layer_output(ufd_model* m, uint32_t layer, uint32_t frame, float* out, size_t cap_floats,
                           size_t* floats);
/* A7-A10 only: threshold + sort + NMS on caller-provided raw outputs of `count` frames. */
int ufd_debug_postproc(ufd_model* m, const float* scores, const float* boxes, uint32_t count, ufd_det* out,
                       uint32_t cap, uint32_t* n);

/* Host half of A1 alone (no GPU needed): marker parse + Huffman decode to quantised DCT
 * coefficients, int16 natural order, laid out [component][block_row][block_col][64] with block
 * counts padded to whole MCUs.  *n_i16 = total int16 count. */
int ufd_debug_jpeg_coefficients(const uint8_t* jpeg, size_t len, int16_t* coef, size_t cap_i16, uint32_t* n_i16,
                                uint32_t* w, uint32_t* h);

/* The launch plan of the network alone (no GPU needed): what ufd_create would issue for `variant` at `max_batch` under
 * `flags` -- per conv layer of SURVEY 8.1 how it runs (kind: 0 pointwise, 1 dw->pw, 2 second block of a chained pair,
 * 3 depthwise computed inside the next launch, 4 dense 3x3, 5 direct), whether it issues a launch of its own at its turn,
 * rides in another layer's grid (`ride`), is computed inside a later launch (`chained`), which tensors it reads and writes;
 * per activation tensor its place in the arena (float offset and size for the whole batch), the turns it is first written
 * and last read (52 = until the head decode) and whether it has storage at all.  *n_layers / *n_tensors receive the full
 * counts; at most the caps are written.  The CPU test suite checks on this that no two tensors share arena bytes while
 * both are live, for every combination of the plan flags. */
typedef struct ufd_plan_layer {
  char name[24];
  int32_t kind, leader, ride, chain_first, fused_dw, chained, materialize, launches, rfb_tail;
  int32_t in_tensor, out_tensor, out_coff, tap_tensor;
} ufd_plan_layer;
typedef struct ufd_plan_tensor {
  uint64_t off_floats, size_floats;
  int32_t c, h, w, first, last, stored;
} ufd_plan_tensor;
int ufd_debug_plan(uint32_t variant, uint32_t max_batch, uint32_t flags, ufd_plan_layer* layers, uint32_t layer_cap, uint32_t* n_layers,
                   ufd_plan_tensor* tensors, uint32_t tensor_cap, uint32_t* n_tensors, uint64_t* arena_floats);

/* get_model alone (nn.rs:143-175; no GPU needed): parse an UltraFace-RFB .onnx into the packed
 * blob (273 888 floats) and, if the graph embeds them, the K*4 priors (*priors_found = 1). */
int ufd_debug_load_onnx(const char* path, uint32_t variant, float* weights, size_t weights_cap, float* priors,
                        size_t priors_cap, uint32_t* priors_found, char* err, size_t err_cap);

/* ---- measurement (bench.py roofline): per-kernel device time from HIP events on the handle's
 * stream; needs UFD_FLAG_PROFILE.  ufd_profile_reset() zeroes the accumulators. */
typedef struct ufd_kernel_stat {
  char name[48];
  uint64_t launches;
  double total_ms;
  double bytes; /* algorithmic bytes moved by those launches (inputs + outputs + weights) */
  double flops; /* algorithmic FLOPs of those launches */
} ufd_kernel_stat;
int ufd_profile_reset(ufd_model* m);
/* Record events only for every `every_n`-th batch (default 1 = all): keeps the event overhead
 * (~100 timestamp packets per batch) out of a timed run while still sampling it. */
int ufd_profile_sampling(ufd_model* m, uint32_t every_n);
int ufd_profile_read(ufd_model* m, ufd_kernel_stat* stats, uint32_t cap, uint32_t* n);
/* How each profiled launch sits on the GPU (DESIGN.md's kernel table, tools/design_table.py): the shape of the label's
 * last launch as the library issued it, the kernel's registers per lane and LDS per workgroup (static + dynamic), and the
 * workgroups one compute unit holds at once by the runtime's own occupancy query -- so
 *   slots = compute_units * resident_per_cu,  rounds = workgroups / slots. */
typedef struct ufd_launch_shape {
  char name[48];
  uint32_t workgroups, threads, lds_bytes, registers, resident_per_cu, compute_units;
} ufd_launch_shape;
int ufd_profile_shapes(ufd_model* m, ufd_launch_shape* shapes, uint32_t cap, uint32_t* n);

/* ---- measurement (bench.py `host` object): what the HOST side of the asynchronous pipeline costs, always on (a handful
 * of clock reads and two event records per batch; no UFD_FLAG_PROFILE needed).  The reference has one blocking task
 * (inferer.rs:29-50); here each of the handle's contexts has an issue worker, and SURVEY 8(e) names the host as the
 * limiter of eight GPUs on one box -- these counters say how far from that a run is.
 *   per batch (sums over the batches issued since the last reset, milliseconds of wall time on the issuing worker):
 *     plan_ms   header + marker scan of the batch's JPEGs, Huffman table-set lookup, scan layouts
 *     copy_ms   the JPEG bytes into the pinned staging block
 *     issue_ms  every hipMemcpyAsync / kernel launch / event of the batch (the worker's time minus the two above)
 *   wait_ms     time callers spent blocked inside ufd_wait (worker not done issuing, or the GPU not done)
 *   per context c < num_ctx:
 *     worker_busy_ms[c]  time the context's issue worker was working (share = / wall_ms)
 *     gpu_span_ms[c]     device time from the first kernel of a batch to its result copy, summed (HIP events)
 *     gpu_gap_ms[c]      device time the context's stream sat between the end of one batch and the first kernel of the
 *                        next (nothing issued yet, or its H2D not there yet), summed over gpu_batches[c] - 1 gaps */
#define UFD_MAX_CTX 8
typedef struct ufd_host_stats {
  uint32_t struct_size; /* = sizeof(ufd_host_stats) */
  uint32_t num_ctx;
  uint64_t batches, launches; /* batches issued; kernel launches + copies enqueued for them */
  double wall_ms;             /* since the last reset */
  double plan_ms, copy_ms, issue_ms;
  uint64_t waits;
  double wait_ms;
  double worker_busy_ms[UFD_MAX_CTX];
  uint64_t gpu_batches[UFD_MAX_CTX];
  double gpu_span_ms[UFD_MAX_CTX];
  double gpu_gap_ms[UFD_MAX_CTX];
} ufd_host_stats;
int ufd_host_stats_reset(ufd_model* m);
int ufd_host_stats_read(ufd_model* m, ufd_host_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* UFD_H */
