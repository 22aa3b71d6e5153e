#define _POSIX_C_SOURCE 200809L /* clock_gettime: time budget of the all-cores baseline leg */
/*
 * postproc_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see ufd_oracle.h) for rows A7-A10
 * and the run()/decode glue.  This part of the path is first-party reference code and is
 * restated line by line:
 *   UltrafaceModel::postproc          infer_server/src/nn.rs:109-140
 *   non_maximum_suppression           infer_server/src/nn.rs:198-224
 *   iou                               infer_server/src/nn.rs:227-243
 *   bbox_area                         infer_server/src/nn.rs:251-260
 *   InferModel::run                   infer_server/src/nn.rs:178-186
 *   Inferer::run decode -> infer      infer_server/src/inferer.rs:35-37
 * f32 arithmetic with separate multiply/add/divide in the reference's operation order
 * (build with -ffp-contract=off).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ufd_oracle.h"

#define UFO_EPS 1.0e-7f /* nn.rs:18 */

/* nn.rs:251-260 (the reference names bbox[3]-bbox[1] "width"; only the product matters) */
float ufo_bbox_area(const float* bbox) {
  float width = bbox[3] - bbox[1];
  float height = bbox[2] - bbox[0];
  if (width < 0.0f || height < 0.0f) return 0.0f;
  return width * height;
}

/* nn.rs:227-243 */
float ufo_iou(const float* a, const float* b) {
  float o[4];
  o[0] = fmaxf(a[0], b[0]);
  o[1] = fmaxf(a[1], b[1]);
  o[2] = fminf(a[2], b[2]);
  o[3] = fminf(a[3], b[3]);
  float overlap = ufo_bbox_area(o);
  return overlap / (ufo_bbox_area(a) + ufo_bbox_area(b) - overlap + UFO_EPS);
}

typedef struct {
  float conf;
  int idx;
} cand_t;

static int cand_cmp(const void* pa, const void* pb) {
  const cand_t* a = (const cand_t*)pa;
  const cand_t* b = (const cand_t*)pb;
  /* ascending confidence; equal confidences keep prior-index order (Rust sort_by is stable) */
  if (a->conf < b->conf) return -1;
  if (a->conf > b->conf) return 1;
  return (a->idx > b->idx) - (a->idx < b->idx);
}

int ufo_postproc(const float* scores, const float* boxes, int K, float min_confidence, float max_iou, ufo_det* out,
                 int cap) {
  /* nn.rs:124-130: keep iff conf > min_confidence (strict; NaN fails the comparison) */
  cand_t* c = (cand_t*)malloc((size_t)(K > 0 ? K : 1) * sizeof(cand_t));
  int* sel = (int*)malloc((size_t)(K > 0 ? K : 1) * sizeof(int));
  if (!c || !sel) {
    free(c);
    free(sel);
    return UFO_E_ARG;
  }
  int n = 0;
  for (int k = 0; k < K; k++) {
    float conf = scores[2 * k + 1];
    if (conf > min_confidence) {
      c[n].conf = conf;
      c[n].idx = k;
      n++;
    }
  }
  /* nn.rs:134 */
  qsort(c, (size_t)n, sizeof(cand_t), cand_cmp);
  /* nn.rs:198-224: pop from the back, reject iff iou > max_iou (strict) with any selected box */
  int nsel = 0;
  while (n > 0) {
    cand_t cur = c[--n];
    const float* bb = boxes + 4 * (size_t)cur.idx;
    int keep = 1;
    for (int j = 0; j < nsel; j++) {
      if (ufo_iou(bb, boxes + 4 * (size_t)sel[j]) > max_iou) {
        keep = 0;
        break;
      }
    }
    if (keep) {
      if (nsel < cap && out) {
        out[nsel].x_tl = bb[0];
        out[nsel].y_tl = bb[1];
        out[nsel].x_br = bb[2];
        out[nsel].y_br = bb[3];
        out[nsel].conf = cur.conf;
      }
      sel[nsel++] = cur.idx;
    }
  }
  free(c);
  free(sel);
  return nsel;
}

int ufo_infer_rgb(const uint8_t* rgb, int w, int h, int model_w, int model_h, const float* weights, const float* priors,
                  float min_confidence, float max_iou, ufo_det* out, int cap) {
  int K = ufo_num_priors(model_w, model_h);
  uint8_t* resized = (uint8_t*)malloc((size_t)model_w * model_h * 3);
  float* input = (float*)malloc((size_t)model_w * model_h * 3 * sizeof(float));
  float* scores = (float*)malloc((size_t)K * 2 * sizeof(float));
  float* boxes = (float*)malloc((size_t)K * 4 * sizeof(float));
  int rc = UFO_E_ARG;
  if (resized && input && scores && boxes) {
    rc = ufo_resize_triangle_rgb(rgb, w, h, resized, model_w, model_h);
    if (rc == UFO_OK) {
      ufo_normalize_nchw(resized, model_w, model_h, input);
      rc = ufo_ultraface_forward(input, model_w, model_h, weights, priors, scores, boxes);
    }
    if (rc == UFO_OK) rc = ufo_postproc(scores, boxes, K, min_confidence, max_iou, out, cap);
  }
  free(resized);
  free(input);
  free(scores);
  free(boxes);
  return rc;
}

int ufo_infer_jpeg(const uint8_t* jpeg, size_t len, int model_w, int model_h, const float* weights, const float* priors,
                   float min_confidence, float max_iou, ufo_det* out, int cap) {
  ufo_jpeg_info info;
  int rc = ufo_jpeg_probe(jpeg, len, &info);
  if (rc) return rc;
  uint8_t* rgb = (uint8_t*)malloc((size_t)info.width * info.height * 3);
  if (!rgb) return UFO_E_ARG;
  rc = ufo_jpeg_decode_rgb(jpeg, len, rgb, info.width, info.height);
  if (rc == UFO_OK)
    rc = ufo_infer_rgb(rgb, info.width, info.height, model_w, model_h, weights, priors, min_confidence, max_iou, out,
                       cap);
  free(rgb);
  return rc;
}

/* ---- bench.py's cpu_baseline leg: the same per-frame path on `threads` host threads (frames are
 * independent, SURVEY 8d "(b) one worker per core").  Each worker takes the next frame index from
 * a shared counter and runs ufo_infer_jpeg on it until `total` frames are taken or `budget_s` seconds
 * have passed (a frame in progress is finished); returns frames done, *dets_total = detections. */
#include <pthread.h>
#include <stdatomic.h>
#include <time.h>

static double ufo_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
  const uint8_t* const* jpegs;
  const size_t* lens;
  int n_frames, total;
  double deadline; /* CLOCK_MONOTONIC seconds after which no new frame is started */
  int model_w, model_h;
  const float *weights, *priors;
  float min_confidence, max_iou;
  int cap;
  atomic_int next, done;
  atomic_long dets;
} ufo_mt_job;

static void* ufo_mt_worker(void* p) {
  ufo_mt_job* j = (ufo_mt_job*)p;
  ufo_det* out = (ufo_det*)malloc(sizeof(ufo_det) * (size_t)j->cap);
  if (!out) return NULL;
  for (;;) {
    int i = atomic_fetch_add(&j->next, 1);
    if (i >= j->total || ufo_now() >= j->deadline) break;
    int f = i % j->n_frames;
    int n = ufo_infer_jpeg(j->jpegs[f], j->lens[f], j->model_w, j->model_h, j->weights, j->priors, j->min_confidence,
                           j->max_iou, out, j->cap);
    if (n >= 0) {
      atomic_fetch_add(&j->dets, n);
      atomic_fetch_add(&j->done, 1);
    }
  }
  free(out);
  return NULL;
}

int ufo_infer_jpeg_mt(const uint8_t* const* jpegs, const size_t* lens, int n_frames, int total, double budget_s, int threads,
                      int model_w, int model_h, const float* weights, const float* priors, float min_confidence, float max_iou, int cap,
                      long* dets_total) {
  if (n_frames < 1 || total < 1 || threads < 1 || threads > 1024) return UFO_E_ARG;
  ufo_mt_job j;
  j.deadline = ufo_now() + (budget_s > 0 ? budget_s : 1e9);
  j.jpegs = jpegs, j.lens = lens, j.n_frames = n_frames, j.total = total, j.model_w = model_w, j.model_h = model_h;
  j.weights = weights, j.priors = priors, j.min_confidence = min_confidence, j.max_iou = max_iou, j.cap = cap;
  atomic_init(&j.next, 0);
  atomic_init(&j.done, 0);
  atomic_init(&j.dets, 0);
  pthread_t th[1024];
  int started = 0;
  for (int t = 0; t < threads; t++)
    if (pthread_create(&th[started], NULL, ufo_mt_worker, &j) == 0) started++;
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
  if (dets_total) *dets_total = atomic_load(&j.dets);
  return atomic_load(&j.done);
}
