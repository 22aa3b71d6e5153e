/*
 * ultraface_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see ufd_oracle.h) for rows A5/A6:
 *   self.model.run(tvec!(input))                     (infer_server/src/nn.rs:181)
 *
 * The arithmetic lives in tract-onnx 0.19.2 (Cargo.lock:2464-2600) executing the ONNX file
 * version-RFB-{320,640}.onnx (downloaded at run time, nn.rs:21-22); neither is under
 * /root/reference.  This restates the published network the file was exported from
 * (Linzaer/Ultra-Light-Fast-Generic-Face-Detector-1MB, Mb_Tiny_RFB_fd):
 * 52 convolutions with BatchNorm already folded into (w, b), ReLU, the RFB block, four
 * SSD heads, softmax over the 2 classes and the prior-box decode embedded in the graph.
 * Single-threaded like tract's SimplePlan.  Pinned against torch.nn.functional.conv2d
 * (tests/test_oracle_cnn.py); box-level parity with tract itself is unpinned.
 *
 * Accumulation order per output element: acc = bias; then for ci, ky, kx in that order
 * acc = fmaf(w, x, acc).  Out-of-image taps contribute exactly nothing.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ufd_oracle.h"

/* {cin, cout, k, stride, pad, dil, groups, relu} -- SURVEY.md section 8.1 rows 0..51 */
static const ufo_conv_spec kSpecs[UFO_NUM_CONV] = {
    {3, 16, 3, 2, 1, 1, 1, 1},      /* 0  m0.conv_bn */
    {16, 16, 3, 1, 1, 1, 16, 1},    /* 1  m1.dw */
    {16, 32, 1, 1, 0, 1, 1, 1},     /* 2  m1.pw */
    {32, 32, 3, 2, 1, 1, 32, 1},    /* 3  m2.dw */
    {32, 32, 1, 1, 0, 1, 1, 1},     /* 4  m2.pw */
    {32, 32, 3, 1, 1, 1, 32, 1},    /* 5  m3.dw */
    {32, 32, 1, 1, 0, 1, 1, 1},     /* 6  m3.pw */
    {32, 32, 3, 2, 1, 1, 32, 1},    /* 7  m4.dw */
    {32, 64, 1, 1, 0, 1, 1, 1},     /* 8  m4.pw */
    {64, 64, 3, 1, 1, 1, 64, 1},    /* 9  m5.dw */
    {64, 64, 1, 1, 0, 1, 1, 1},     /* 10 m5.pw */
    {64, 64, 3, 1, 1, 1, 64, 1},    /* 11 m6.dw */
    {64, 64, 1, 1, 0, 1, 1, 1},     /* 12 m6.pw */
    {64, 8, 1, 1, 0, 1, 1, 0},      /* 13 rfb.b0.0 */
    {8, 16, 3, 1, 1, 1, 1, 1},      /* 14 rfb.b0.1 */
    {16, 16, 3, 1, 2, 2, 1, 0},     /* 15 rfb.b0.2 */
    {64, 8, 1, 1, 0, 1, 1, 0},      /* 16 rfb.b1.0 */
    {8, 16, 3, 1, 1, 1, 1, 1},      /* 17 rfb.b1.1 */
    {16, 16, 3, 1, 3, 3, 1, 0},     /* 18 rfb.b1.2 */
    {64, 8, 1, 1, 0, 1, 1, 0},      /* 19 rfb.b2.0 */
    {8, 12, 3, 1, 1, 1, 1, 1},      /* 20 rfb.b2.1 */
    {12, 16, 3, 1, 1, 1, 1, 1},     /* 21 rfb.b2.2 */
    {16, 16, 3, 1, 5, 5, 1, 0},     /* 22 rfb.b2.3 */
    {48, 64, 1, 1, 0, 1, 1, 0},     /* 23 rfb.ConvLinear */
    {64, 64, 1, 1, 0, 1, 1, 0},     /* 24 rfb.shortcut (+add, relu) */
    {64, 64, 3, 1, 1, 1, 64, 1},    /* 25 cls0.dw */
    {64, 6, 1, 1, 0, 1, 1, 0},      /* 26 cls0.pw */
    {64, 64, 3, 1, 1, 1, 64, 1},    /* 27 reg0.dw */
    {64, 12, 1, 1, 0, 1, 1, 0},     /* 28 reg0.pw */
    {64, 64, 3, 2, 1, 1, 64, 1},    /* 29 m8.dw */
    {64, 128, 1, 1, 0, 1, 1, 1},    /* 30 m8.pw */
    {128, 128, 3, 1, 1, 1, 128, 1}, /* 31 m9.dw */
    {128, 128, 1, 1, 0, 1, 1, 1},   /* 32 m9.pw */
    {128, 128, 3, 1, 1, 1, 128, 1}, /* 33 m10.dw */
    {128, 128, 1, 1, 0, 1, 1, 1},   /* 34 m10.pw */
    {128, 128, 3, 1, 1, 1, 128, 1}, /* 35 cls1.dw */
    {128, 4, 1, 1, 0, 1, 1, 0},     /* 36 cls1.pw */
    {128, 128, 3, 1, 1, 1, 128, 1}, /* 37 reg1.dw */
    {128, 8, 1, 1, 0, 1, 1, 0},     /* 38 reg1.pw */
    {128, 128, 3, 2, 1, 1, 128, 1}, /* 39 m11.dw */
    {128, 256, 1, 1, 0, 1, 1, 1},   /* 40 m11.pw */
    {256, 256, 3, 1, 1, 1, 256, 1}, /* 41 m12.dw */
    {256, 256, 1, 1, 0, 1, 1, 1},   /* 42 m12.pw */
    {256, 256, 3, 1, 1, 1, 256, 1}, /* 43 cls2.dw */
    {256, 4, 1, 1, 0, 1, 1, 0},     /* 44 cls2.pw */
    {256, 256, 3, 1, 1, 1, 256, 1}, /* 45 reg2.dw */
    {256, 8, 1, 1, 0, 1, 1, 0},     /* 46 reg2.pw */
    {256, 64, 1, 1, 0, 1, 1, 1},    /* 47 extra.0 */
    {64, 64, 3, 2, 1, 1, 64, 1},    /* 48 extra.2.dw */
    {64, 256, 1, 1, 0, 1, 1, 1},    /* 49 extra.2.pw */
    {256, 6, 3, 1, 1, 1, 1, 0},     /* 50 cls3 */
    {256, 12, 3, 1, 1, 1, 1, 0},    /* 51 reg3 */
};

/* input of every conv: -1 = network input, -2 = RFB concat, otherwise a layer index whose
 * (block) output feeds it.  Layer 24's stored output is the whole RFB block output. */
static const int kInputOf[UFO_NUM_CONV] = {-1, 0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 12, 16,
                                           17, 12, 19, 20, 21, -2, 12, 24, 25, 24, 27, 24, 29, 30, 31, 32, 33, 34,
                                           35, 34, 37, 34, 39, 40, 41, 42, 43, 42, 45, 42, 47, 48, 49, 49};

const ufo_conv_spec* ufo_conv_specs(void) { return kSpecs; }

size_t ufo_weight_floats(void) {
  size_t n = 0;
  for (int i = 0; i < UFO_NUM_CONV; i++) {
    const ufo_conv_spec* s = &kSpecs[i];
    n += (size_t)s->cout * (s->cin / s->groups) * s->k * s->k + s->cout;
  }
  return n;
}

static int conv_out(int in, const ufo_conv_spec* s) {
  return (in + 2 * s->pad - s->dil * (s->k - 1) - 1) / s->stride + 1;
}

static void layer_in_hw(int layer, int width, int height, int* ih, int* iw) {
  int src = kInputOf[layer];
  if (src == -1) {
    *ih = height;
    *iw = width;
  } else if (src == -2) {
    ufo_layer_out_hw(15, width, height, ih, iw);
  } else {
    ufo_layer_out_hw(src, width, height, ih, iw);
  }
}

void ufo_layer_out_hw(int layer, int width, int height, int* oh, int* ow) {
  int ih, iw;
  layer_in_hw(layer, width, height, &ih, &iw);
  *oh = conv_out(ih, &kSpecs[layer]);
  *ow = conv_out(iw, &kSpecs[layer]);
}

/* strides 8/16/32/64; feature maps ceil(size/stride); min_boxes per upstream config */
static const int kStrides[4] = {8, 16, 32, 64};
static const int kNumAnchors[4] = {3, 2, 2, 3};
static const double kMinBoxes[4][3] = {{10, 16, 24}, {32, 48, 0}, {64, 96, 0}, {128, 192, 256}};

int ufo_num_priors(int width, int height) {
  int k = 0;
  for (int i = 0; i < 4; i++) {
    int fw = (width + kStrides[i] - 1) / kStrides[i], fh = (height + kStrides[i] - 1) / kStrides[i];
    k += fw * fh * kNumAnchors[i];
  }
  return k;
}

void ufo_gen_priors(int width, int height, float* out) {
  /* upstream vision/utils/box_utils.py generate_priors, evaluated in float64 then cast to f32 */
  for (int idx = 0; idx < 4; idx++) {
    int fw = (width + kStrides[idx] - 1) / kStrides[idx], fh = (height + kStrides[idx] - 1) / kStrides[idx];
    double shrink_w = (double)width / fw, shrink_h = (double)height / fh;
    double scale_w = (double)width / shrink_w, scale_h = (double)height / shrink_h;
    for (int j = 0; j < fh; j++)
      for (int i = 0; i < fw; i++) {
        double xc = (i + 0.5) / scale_w, yc = (j + 0.5) / scale_h;
        for (int m = 0; m < kNumAnchors[idx]; m++) {
          double v[4] = {xc, yc, kMinBoxes[idx][m] / width, kMinBoxes[idx][m] / height};
          for (int c = 0; c < 4; c++) {
            float f = (float)v[c];
            *out++ = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f);
          }
        }
      }
  }
}

/* generic NCHW convolution, groups in {1, cin}; in [cin][ih][iw] -> out [cout][oh][ow] */
static void conv2d(const float* in, int ih, int iw, const ufo_conv_spec* s, const float* w, const float* b, float* out,
                   int oh, int ow, int relu) {
  const int k = s->k, st = s->stride, pad = s->pad, dil = s->dil;
  const int cpg = s->cin / s->groups; /* input channels per group */
  const int opg = s->cout / s->groups;
  for (int co = 0; co < s->cout; co++) {
    float* o = out + (size_t)co * oh * ow;
    const float bias = b[co];
    for (int i = 0; i < oh * ow; i++) o[i] = bias;
    const int g = co / opg;
    for (int cl = 0; cl < cpg; cl++) {
      const float* ip = in + (size_t)(g * cpg + cl) * ih * iw;
      const float* wp = w + ((size_t)co * cpg + cl) * k * k;
      for (int ky = 0; ky < k; ky++)
        for (int kx = 0; kx < k; kx++) {
          const float wv = wp[ky * k + kx];
          const int dy = ky * dil - pad, dx = kx * dil - pad;
          /* valid output range: 0 <= oy*st+dy < ih */
          int oy0 = dy < 0 ? (-dy + st - 1) / st : 0;
          int oy1 = (ih - 1 - dy) >= 0 ? (ih - 1 - dy) / st + 1 : 0;
          if (oy1 > oh) oy1 = oh;
          int ox0 = dx < 0 ? (-dx + st - 1) / st : 0;
          int ox1 = (iw - 1 - dx) >= 0 ? (iw - 1 - dx) / st + 1 : 0;
          if (ox1 > ow) ox1 = ow;
          for (int oy = oy0; oy < oy1; oy++) {
            const float* irow = ip + (size_t)(oy * st + dy) * iw + dx;
            float* orow = o + (size_t)oy * ow;
            if (st == 1) {
              for (int ox = ox0; ox < ox1; ox++) orow[ox] = fmaf(wv, irow[ox], orow[ox]);
            } else {
              for (int ox = ox0; ox < ox1; ox++) orow[ox] = fmaf(wv, irow[ox * st], orow[ox]);
            }
          }
        }
    }
    if (relu)
      for (int i = 0; i < oh * ow; i++) o[i] = o[i] > 0.0f ? o[i] : 0.0f;
  }
}

int ufo_ultraface_forward_layers(const float* input, int width, int height, const float* weights, const float* priors,
                                 float* scores, float* boxes, float* const* layer_out) {
  if (!input || !weights || !priors || !scores || !boxes) return UFO_E_ARG;
  float* act[UFO_NUM_CONV];
  int oh[UFO_NUM_CONV], ow[UFO_NUM_CONV];
  memset(act, 0, sizeof(act));
  const float* wp[UFO_NUM_CONV];
  const float* bp[UFO_NUM_CONV];
  {
    const float* p = weights;
    for (int i = 0; i < UFO_NUM_CONV; i++) {
      const ufo_conv_spec* s = &kSpecs[i];
      wp[i] = p;
      p += (size_t)s->cout * (s->cin / s->groups) * s->k * s->k;
      bp[i] = p;
      p += s->cout;
    }
  }
  float* cat = NULL;
  int rc = UFO_OK;
  for (int i = 0; i < UFO_NUM_CONV; i++) {
    const ufo_conv_spec* s = &kSpecs[i];
    int ih, iw;
    layer_in_hw(i, width, height, &ih, &iw);
    ufo_layer_out_hw(i, width, height, &oh[i], &ow[i]);
    act[i] = (float*)malloc((size_t)s->cout * oh[i] * ow[i] * sizeof(float));
    if (!act[i]) {
      rc = UFO_E_ARG;
      goto done;
    }
    const float* in;
    int src = kInputOf[i];
    if (src == -1) {
      in = input;
    } else if (src == -2) {
      /* torch.cat((x0, x1, x2), 1): layers 15, 18, 22, 16 channels each */
      size_t plane = (size_t)oh[15] * ow[15];
      cat = (float*)malloc(48 * plane * sizeof(float));
      if (!cat) {
        rc = UFO_E_ARG;
        goto done;
      }
      memcpy(cat, act[15], 16 * plane * sizeof(float));
      memcpy(cat + 16 * plane, act[18], 16 * plane * sizeof(float));
      memcpy(cat + 32 * plane, act[22], 16 * plane * sizeof(float));
      in = cat;
    } else {
      in = act[src];
    }
    conv2d(in, ih, iw, s, wp[i], bp[i], act[i], oh[i], ow[i], s->relu);
    if (i == 24) {
      /* out = relu(ConvLinear(cat) * scale(=1.0) + shortcut(x)) */
      size_t n = (size_t)64 * oh[i] * ow[i];
      for (size_t j = 0; j < n; j++) {
        float v = act[23][j] * 1.0f + act[24][j];
        act[24][j] = v > 0.0f ? v : 0.0f;
      }
    }
  }
  /* heads: NCHW [A*c][h][w] -> NHWC -> [h*w*A][c]; concat heads 0..3 */
  {
    static const int cls_layer[4] = {26, 36, 44, 50}, reg_layer[4] = {28, 38, 46, 51};
    int base = 0;
    for (int h = 0; h < 4; h++) {
      int A = kNumAnchors[h];
      int fh = oh[cls_layer[h]], fw = ow[cls_layer[h]];
      size_t plane = (size_t)fh * fw;
      const float* cls = act[cls_layer[h]];
      const float* reg = act[reg_layer[h]];
      for (size_t p = 0; p < plane; p++)
        for (int a = 0; a < A; a++) {
          size_t k = (size_t)base + p * A + a;
          float s0 = cls[(size_t)(a * 2 + 0) * plane + p], s1 = cls[(size_t)(a * 2 + 1) * plane + p];
          float mx = s0 > s1 ? s0 : s1;
          float e0 = expf(s0 - mx), e1 = expf(s1 - mx);
          float sum = e0 + e1;
          scores[2 * k] = e0 / sum;
          scores[2 * k + 1] = e1 / sum;
          float l0 = reg[(size_t)(a * 4 + 0) * plane + p], l1 = reg[(size_t)(a * 4 + 1) * plane + p];
          float l2 = reg[(size_t)(a * 4 + 2) * plane + p], l3 = reg[(size_t)(a * 4 + 3) * plane + p];
          const float* pr = priors + 4 * k;
          float cx = (l0 * 0.1f) * pr[2] + pr[0], cy = (l1 * 0.1f) * pr[3] + pr[1];
          float bw = expf(l2 * 0.2f) * pr[2], bh = expf(l3 * 0.2f) * pr[3];
          boxes[4 * k + 0] = cx - bw / 2.0f;
          boxes[4 * k + 1] = cy - bh / 2.0f;
          boxes[4 * k + 2] = cx + bw / 2.0f;
          boxes[4 * k + 3] = cy + bh / 2.0f;
        }
      base += (int)plane * A;
    }
  }
  if (layer_out)
    for (int i = 0; i < UFO_NUM_CONV; i++)
      if (layer_out[i]) memcpy(layer_out[i], act[i], (size_t)kSpecs[i].cout * oh[i] * ow[i] * sizeof(float));
done:
  for (int i = 0; i < UFO_NUM_CONV; i++) free(act[i]);
  free(cat);
  return rc;
}

int ufo_ultraface_forward(const float* input, int width, int height, const float* weights, const float* priors,
                          float* scores, float* boxes) {
  return ufo_ultraface_forward_layers(input, width, height, weights, priors, scores, boxes, NULL);
}
