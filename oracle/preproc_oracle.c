/*
 * preproc_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see ufd_oracle.h) for rows A2-A4:
 *   UltrafaceModel::preproc                         (infer_server/src/nn.rs:70-94)
 *     image::imageops::resize(input, W, H, FilterType::Triangle)   nn.rs:74-80
 *     (v as f32 / 255.0 - mean[c]) / std[c] -> NCHW f32            nn.rs:82-93
 *
 * The resampler lives in the `image` crate 0.24.5 (Cargo.lock:946-961), which is not under
 * /root/reference; this restates its src/imageops/sample.rs algorithm:
 * vertical pass over the full source width into an unrounded f32 intermediate, then a
 * horizontal pass, clamp to [0,255], round-half-away (f32::round), cast to u8.  Windows are
 * truncated and renormalised at the borders.  All arithmetic is f32 with separate multiply
 * and add (Rust never contracts to fma) -- build with -ffp-contract=off.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ufd_oracle.h"

static inline float tri_kernel(float x) {
  float a = fabsf(x);
  return a < 1.0f ? 1.0f - a : 0.0f;
}

typedef struct {
  int left, n;
  float* w;
} taps_t;

/* weights for output index o along an axis of source length S, destination length D */
static int axis_taps(int S, int D, int o, taps_t* t, float* wbuf) {
  float ratio = (float)S / (float)D;
  float sratio = ratio < 1.0f ? 1.0f : ratio;
  float support = 1.0f * sratio;
  float in = ((float)o + 0.5f) * ratio;
  long left = (long)floorf(in - support);
  if (left < 0) left = 0;
  if (left > (long)S - 1) left = (long)S - 1;
  long right = (long)ceilf(in + support);
  if (right < left + 1) right = left + 1;
  if (right > (long)S) right = (long)S;
  in = in - 0.5f;
  float sum = 0.0f;
  int n = 0;
  for (long i = left; i < right; i++) {
    float w = tri_kernel(((float)i - in) / sratio);
    wbuf[n++] = w;
    sum += w;
  }
  for (int i = 0; i < n; i++) wbuf[i] /= sum;
  t->left = (int)left;
  t->n = n;
  t->w = wbuf;
  return n;
}

static int max_taps(int S, int D) {
  float ratio = (float)S / (float)D;
  float sratio = ratio < 1.0f ? 1.0f : ratio;
  return (int)ceilf(2.0f * sratio) + 3;
}

/* The window of output index o (tests check it against an exact-rational derivation): first source
 * index, tap count and the normalised f32 weights exactly as the two passes below use them. */
int ufo_axis_taps(int S, int D, int o, int* left, int* n, float* w, int cap) {
  if (S <= 0 || D <= 0 || o < 0 || o >= D || !left || !n || !w || cap < max_taps(S, D)) return UFO_E_ARG;
  taps_t t;
  axis_taps(S, D, o, &t, w);
  *left = t.left;
  *n = t.n;
  return UFO_OK;
}

int ufo_resize_triangle_rgb(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
  if (!src || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return UFO_E_ARG;
  if (sw == dw && sh == dh) { /* image 0.24.5 resize(): same dimensions -> plain copy */
    memcpy(dst, src, (size_t)sw * sh * 3);
    return UFO_OK;
  }
  /* pass 1: vertical, f32 intermediate [dh][sw][3] (the Rgba32F alpha lane is never read back) */
  float* tmp = (float*)malloc((size_t)dh * sw * 3 * sizeof(float));
  float* wbuf = (float*)malloc((size_t)(max_taps(sh, dh) > max_taps(sw, dw) ? max_taps(sh, dh) : max_taps(sw, dw)) *
                               sizeof(float));
  if (!tmp || !wbuf) {
    free(tmp);
    free(wbuf);
    return UFO_E_ARG;
  }
  taps_t t;
  for (int oy = 0; oy < dh; oy++) {
    axis_taps(sh, dh, oy, &t, wbuf);
    for (int x = 0; x < sw; x++) {
      float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
      for (int i = 0; i < t.n; i++) {
        const uint8_t* p = src + ((size_t)(t.left + i) * sw + x) * 3;
        float w = t.w[i];
        a0 += (float)p[0] * w;
        a1 += (float)p[1] * w;
        a2 += (float)p[2] * w;
      }
      float* o = tmp + ((size_t)oy * sw + x) * 3;
      o[0] = a0;
      o[1] = a1;
      o[2] = a2;
    }
  }
  /* pass 2: horizontal, clamp + round-half-away + u8 */
  for (int ox = 0; ox < dw; ox++) {
    axis_taps(sw, dw, ox, &t, wbuf);
    for (int y = 0; y < dh; y++) {
      float a[3] = {0.0f, 0.0f, 0.0f};
      for (int i = 0; i < t.n; i++) {
        const float* p = tmp + ((size_t)y * sw + t.left + i) * 3;
        float w = t.w[i];
        a[0] += p[0] * w;
        a[1] += p[1] * w;
        a[2] += p[2] * w;
      }
      uint8_t* o = dst + ((size_t)y * dw + ox) * 3;
      for (int c = 0; c < 3; c++) {
        float v = a[c];
        v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
        o[c] = (uint8_t)roundf(v);
      }
    }
  }
  free(tmp);
  free(wbuf);
  return UFO_OK;
}

void ufo_normalize_nchw(const uint8_t* rgb, int w, int h, float* out) {
  /* nn.rs:86-88: f32 literals; divide, do not multiply by a reciprocal */
  static const float mean[3] = {0.485f, 0.456f, 0.406f};
  static const float stdv[3] = {0.229f, 0.224f, 0.225f};
  size_t hw = (size_t)w * h;
  for (int c = 0; c < 3; c++)
    for (size_t i = 0; i < hw; i++) out[c * hw + i] = ((float)rgb[3 * i + c] / 255.0f - mean[c]) / stdv[c];
}
