/*
 * ufd_oracle.h -- CPU oracle for the infer_server face-detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * per-frame algorithm (JPEG decode -> Triangle resize -> normalize -> UltraFace-RFB
 * forward -> softmax/prior decode -> strict-> threshold -> stable sort -> greedy NMS).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product library (infercam_onnx_amd/csrc) never includes, links or calls anything
 * in this directory.
 *
 * PARITY STATUS: "parity unpinned" at the box level against the reference itself.
 * The reference (Rust: tract-onnx 0.19.2, image 0.24.5, turbojpeg 0.5.2) cannot be
 * built or run in this environment (no cargo/rustc, crates not vendored, no model
 * file), and its only pinned results are eight face counts
 * that need the real .onnx.  What pins this oracle instead:
 *   - JPEG decode: bit-exact against libjpeg-turbo 3.1.x (via PIL) on committed
 *     fixtures (tests/golden/jpeg_*.npz) -- same library, same defaults as
 *     turbojpeg::decompress_image (tjDecompress2 flags=0: ISLOW IDCT, fancy upsampling).
 *   - Resize/normalize: restated from image 0.24.5 sample.rs semantics,
 *     checked against an independent numpy twin + analytic tap tables, and within one grey level of Pillow's BILINEAR.
 *   - CNN: checked against torch.nn.functional.conv2d (independent implementation).
 *   - Threshold / sort / NMS / IoU: first-party reference code, restated line by line
 *     from infer_server/src/nn.rs:109-140,198-260; known-answer cases in tests.
 */
#ifndef UFD_ORACLE_H
#define UFD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (oracle-local) ---- */
#define UFO_OK 0
#define UFO_E_DECODE (-1)      /* corrupt / truncated stream */
#define UFO_E_UNSUPPORTED (-2) /* arithmetic coding, 12-bit, CMYK, ... */
#define UFO_E_ARG (-3)

/* ---- A1: JPEG decode (follows turbojpeg::decompress_image, inferer.rs:35) ---- */
typedef struct {
  int width, height, ncomp;
  int progressive;        /* SOF2 */
  int hsamp[4], vsamp[4]; /* sampling factors per component */
  int restart_interval;   /* DRI, in MCUs (0 = none) */
} ufo_jpeg_info;

int ufo_jpeg_probe(const uint8_t* data, size_t len, ufo_jpeg_info* info);
/* rgb: height*width*3 bytes, interleaved, pitch = 3*width */
int ufo_jpeg_decode_rgb(const uint8_t* data, size_t len, uint8_t* rgb, int width, int height);

/* entropy-decoded quantised coefficients, int16 natural order, [comp][block_row][block_col][64]
 * with MCU-padded block counts (for checking the product's host Huffman stage on a CPU box).
 * coef may be NULL to query *n_i16. */
int ufo_jpeg_coefficients(const uint8_t* data, size_t len, int16_t* coef, size_t cap_i16, size_t* n_i16);

/* ---- A2/A3: image::imageops::resize(.., FilterType::Triangle)  (nn.rs:74-80) ---- */
/* src: sh x sw x 3 u8 (pitch 3*sw), dst: dh x dw x 3 u8 */
int ufo_resize_triangle_rgb(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);
/* window of output index o along an axis S -> D as the resampler uses it (cap >= ceil(2*max(S/D,1)) + 3 floats) */
int ufo_axis_taps(int S, int D, int o, int* left, int* n, float* w, int cap);

/* ---- A4: normalize closure (nn.rs:82-93): HWC u8 -> NCHW f32 [3][h][w] ---- */
void ufo_normalize_nchw(const uint8_t* rgb, int w, int h, float* out);

/* ---- A5/A6: UltraFace-RFB forward (nn.rs:181; topology SURVEY 8.1) ---- */
#define UFO_NUM_CONV 52
typedef struct {
  int cin, cout, k, stride, pad, dil, groups, relu;
} ufo_conv_spec;
const ufo_conv_spec* ufo_conv_specs(void); /* UFO_NUM_CONV entries */

/* number of priors for a model input size (4420 for 320x240, 17640 for 640x480) */
int ufo_num_priors(int width, int height);
/* upstream prior generator (float64 -> f32, clamped to [0,1]); out: K*4 */
void ufo_gen_priors(int width, int height, float* out);
/* total float count of the packed weight blob: for each conv, w[cout][cin/g][k][k] then b[cout] */
size_t ufo_weight_floats(void);

/* input: [3][H][W] f32 normalized; weights: packed blob; priors: K*4 (cx,cy,w,h)
 * scores: K*2 (softmax), boxes: K*4 (corner form, relative). */
int ufo_ultraface_forward(const float* input, int width, int height, const float* weights,
                          const float* priors, float* scores, float* boxes);
/* same, but also returns every conv layer's post-activation output (for per-layer parity).
 * layer_out[i] must hold cout_i*h_i*w_i floats or be NULL. */
int ufo_ultraface_forward_layers(const float* input, int width, int height, const float* weights,
                                 const float* priors, float* scores, float* boxes,
                                 float* const* layer_out);
/* spatial size of conv layer i's output for a given model input size */
void ufo_layer_out_hw(int layer, int width, int height, int* oh, int* ow);

/* ---- A7-A10: postproc (nn.rs:109-140) + NMS (nn.rs:198-224) + iou/bbox_area (nn.rs:227-260) ---- */
typedef struct {
  float x_tl, y_tl, x_br, y_br, conf;
} ufo_det;
float ufo_bbox_area(const float* bbox);
float ufo_iou(const float* a, const float* b);
/* returns number of detections (may exceed cap; only cap are written) */
int ufo_postproc(const float* scores, const float* boxes, int K, float min_confidence,
                 float max_iou, ufo_det* out, int cap);

/* ---- InferModel::run (nn.rs:178-186): preproc -> forward -> postproc ---- */
int ufo_infer_rgb(const uint8_t* rgb, int w, int h, int model_w, int model_h, const float* weights,
                  const float* priors, float min_confidence, float max_iou, ufo_det* out, int cap);
/* Inferer::run steps inferer.rs:35-37: decompress_image -> infer_faces */
int ufo_infer_jpeg(const uint8_t* jpeg, size_t len, int model_w, int model_h, const float* weights,
                   const float* priors, float min_confidence, float max_iou, ufo_det* out, int cap);
/* bench.py cpu_baseline, all-cores leg: up to `total` frames (cycling over the n_frames given) on `threads`
 * host threads, each running ufo_infer_jpeg, no frame started after `budget_s` seconds (<= 0: no limit);
 * returns the number of frames done. */
int ufo_infer_jpeg_mt(const uint8_t* const* jpegs, const size_t* lens, int n_frames, int total, double budget_s, int threads,
                      int model_w, int model_h, const float* weights, const float* priors, float min_confidence, float max_iou, int cap,
                      long* dets_total);

/* ---- N1: what Inferer::run does with the detections (inferer.rs:38-40, lib.rs:48-57) ---- */
/* inferer.rs:66-76: rectangle of one detection on a frame labelled label_w x label_h (the slot's width / height,
 * router.rs:66-67); returns 0 when Rect::of_size would assert (zero width or height). Inclusive pixel bounds. */
int ufo_rect_of_det(const ufo_det* d, float label_w, float label_h, int64_t* left, int64_t* top, int64_t* right, int64_t* bottom);
/* draw_hollow_rect of every detection, colour (0, 255, 0), clipped to the w x h frame (text: not restated) */
void ufo_draw_hollow_rects(uint8_t* rgb, int w, int h, const ufo_det* dets, int n, float label_w, float label_h);
/* the label of a detection: format!("{:.2}%", confidence * 100.0) as indices into "0123456789.%"; returns its length */
int ufo_label_chars(float confidence, uint8_t chars[8]);
/* draw_bboxes_on_image (inferer.rs:58-92): per detection, in order, the rectangle then the label (glyph_atlas.inc) */
void ufo_draw_labels(uint8_t* rgb, int w, int h, const ufo_det* dets, int n, float label_w, float label_h);
/* jpeg_set_quality(quality, TRUE) table, natural order */
void ufo_jpeg_quant_table(int quality, int chroma, uint8_t out[64]);
size_t ufo_jpeg_encode_bound(int w, int h);
/* turbojpeg::compress_image(&frame, quality, Subsamp::Sub2x2) (inferer.rs:39). dct: 0 = JDCT_ISLOW, 1 = JDCT_IFAST,
 * -1 = what tjCompress2 picks (IFAST below quality 96). */
int ufo_jpeg_encode_rgb(const uint8_t* rgb, int w, int h, int quality, int dct, uint8_t* out, size_t cap, size_t* len);
/* the quantised coefficients behind that stream: [mcu][Y00 Y01 Y10 Y11 Cb Cr][64, natural order] */
int ufo_jpeg_encode_coefficients(const uint8_t* rgb, int w, int h, int quality, int dct, int16_t* coef);
/* as_jpeg_stream_item (lib.rs:48-57): returns the framed length; writes when cap suffices */
size_t ufo_stream_item(const uint8_t* jpeg, size_t len, uint8_t* out, size_t cap);
/* inferer.rs:35-40: decode -> infer -> rectangles + labels -> encode; returns the detection count or < 0 */
int ufo_annotate_encode_jpeg(const uint8_t* jpeg, size_t len, int model_w, int model_h, const float* weights,
                             const float* priors, float min_confidence, float max_iou, float label_w, float label_h, int quality,
                             ufo_det* dets, int cap, uint8_t* out, size_t out_cap, size_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
