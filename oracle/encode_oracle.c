/*
 * encode_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see ufd_oracle.h) for SURVEY 8(f) row N1,
 * the step that follows NMS in Inferer::run:
 *
 *   draw_bboxes_on_image(image, boxes, width, height)            infer_server/src/inferer.rs:38,58-92
 *   turbojpeg::compress_image(&frame, 95, Subsamp::Sub2x2)       infer_server/src/inferer.rs:39
 *   as_jpeg_stream_item(&buf)                                    infer_server/src/lib.rs:48-57
 *
 * Drawing: the first-party arithmetic (inferer.rs:66-73: bbox * (width, height) in f32, `as i32` /
 * `as u32` casts) is restated line by line; draw_hollow_rect is imageproc 0.23 (Cargo.lock, not under
 * /root/reference): four Bresenham segments (left,top)-(right,top), (left,bottom)-(right,bottom),
 * (left,top)-(left,bottom), (right,top)-(right,bottom) with right = left + w - 1, bottom = top + h - 1,
 * every point clipped against the image.  The confidence text (draw_text, inferer.rs:80-88): imageproc 0.23
 * draw_text_mut blends, per glyph pixel, pixel * (1 - v) + colour * v in f32 and casts back with its Clamp (truncation);
 * the coverage v comes from glyph_atlas.inc, a DATA table made by tools/make_glyph_atlas.py, which restates rusttype
 * 0.9.3's layout (caret advance, pixel bounding boxes) and ab_glyph_rasterizer's accumulation for DejaVuSansMono at
 * Scale 16 -- "parity unpinned": none of those crates can be run here (FreeType agrees on the shapes, not bit for bit).
 * The label is format!("{:.2}%", confidence * 100.0): the f32 product, correctly rounded to two decimals.
 *
 * Encoding: turbojpeg 0.5.2 / turbojpeg-sys 0.2.2 (Cargo.lock:2617-2640) = libjpeg-turbo 2.1.x through
 * tjCompress2(flags = 0): jpeg_set_defaults, jpeg_set_quality(q, TRUE), YCbCr 4:2:0, Annex-K Huffman
 * tables, JFIF 1.01 header, no restart markers, and dct_method = JDCT_IFAST below quality 96
 * (JDCT_ISLOW from 96 on).  This file restates the published libjpeg algorithms (jccolor rgb_ycc,
 * jcsample h2v2_downsample, jcprepct / jccoefct edge rules, jfdctfst / jfdctint, jcdctmgr reciprocal
 * quantisation, jchuff, jcmarker).  PINNED byte for byte against libjpeg-turbo itself:
 * tests/golden/encode_q95_420.npz holds streams written by the libjpeg-turbo 2.1.2 of this image
 * (tools/make_encode_golden.py) for both DCT methods, and the ISLOW streams equal PIL's libjpeg-turbo 3.1.x.
 */
#include <stdlib.h>
#include <string.h>

#include "ufd_oracle.h"

/* ------------------------------------------------------------------------------------------------
 * draw_bboxes_on_image, rectangles (inferer.rs:58-92) */

/* Rust `f32 as i32` / `f32 as u32`: truncate toward zero, saturate, NaN -> 0 */
static int32_t f32_as_i32(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return INT32_MAX;
  if (v <= -2147483648.0f) return INT32_MIN;
  return (int32_t)v;
}
static uint32_t f32_as_u32(float v) {
  if (v != v || v <= 0.0f) return 0;
  if (v >= 4294967296.0f) return UINT32_MAX;
  return (uint32_t)v;
}

int ufo_rect_of_det(const ufo_det* d, float width, float height, int64_t* left, int64_t* top, int64_t* right, int64_t* bottom) {
  const float x_tl = d->x_tl * width, y_tl = d->y_tl * height; /* inferer.rs:69 */
  const float x_br = d->x_br * width, y_br = d->y_br * height; /* inferer.rs:70 */
  const float rect_width = x_br - x_tl, rect_height = y_br - y_tl;
  const uint32_t w = f32_as_u32(rect_width), h = f32_as_u32(rect_height);
  if (w == 0 || h == 0) return 0; /* Rect::of_size asserts > 0: the reference task would panic here */
  *left = f32_as_i32(x_tl), *top = f32_as_i32(y_tl);
  *right = *left + (int64_t)w - 1, *bottom = *top + (int64_t)h - 1;
  return 1;
}

void ufo_draw_hollow_rects(uint8_t* rgb, int w, int h, const ufo_det* dets, int n, float label_w, float label_h) {
  for (int i = 0; i < n; i++) { /* in detection order: later rectangles overwrite earlier ones (same colour) */
    int64_t l, t, r, b;
    if (!ufo_rect_of_det(&dets[i], label_w, label_h, &l, &t, &r, &b)) continue;
    const int64_t x0 = l < 0 ? 0 : l, x1 = r > w - 1 ? w - 1 : r;
    const int64_t y0 = t < 0 ? 0 : t, y1 = b > h - 1 ? h - 1 : b;
    for (int64_t x = x0; x <= x1; x++) {
      if (t >= 0 && t < h) rgb[((size_t)t * w + x) * 3] = 0, rgb[((size_t)t * w + x) * 3 + 1] = 255, rgb[((size_t)t * w + x) * 3 + 2] = 0;
      if (b >= 0 && b < h) rgb[((size_t)b * w + x) * 3] = 0, rgb[((size_t)b * w + x) * 3 + 1] = 255, rgb[((size_t)b * w + x) * 3 + 2] = 0;
    }
    for (int64_t y = y0; y <= y1; y++) {
      if (l >= 0 && l < w) rgb[((size_t)y * w + l) * 3] = 0, rgb[((size_t)y * w + l) * 3 + 1] = 255, rgb[((size_t)y * w + l) * 3 + 2] = 0;
      if (r >= 0 && r < w) rgb[((size_t)y * w + r) * 3] = 0, rgb[((size_t)y * w + r) * 3 + 1] = 255, rgb[((size_t)y * w + r) * 3 + 2] = 0;
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * draw_text(&frame, color, x_tl as i32, y_tl as i32, Scale 16, DejaVuSansMono, "{:.2}%") (inferer.rs:80-88) */
#include <math.h>
#include <stdio.h>

#include "glyph_atlas.inc"

/* format!("{:.2}%", confidence * 100.0): characters as indices into "0123456789.%"; returns the length (<= 8) or 0 when
 * the value has no such representation (negative, NaN, >= 1000: never a confidence) */
int ufo_label_chars(float confidence, uint8_t chars[8]) {
  const float v = confidence * 100.0f; /* f32 product, as in the reference */
  if (!(v >= 0.0f) || v >= 1000.0f) return 0;
  /* v * 100 is exact in double (24 + 7 significant bits); nearbyint rounds half to even like Rust's exact formatting */
  const long r = (long)nearbyint((double)v * 100.0);
  char buf[16];
  const int n = snprintf(buf, sizeof(buf), "%ld.%02ld%%", r / 100, r % 100);
  if (n < 1 || n > UFD_GLYPH_POSITIONS) return 0;
  for (int i = 0; i < n; i++) chars[i] = (uint8_t)(buf[i] == '.' ? 10 : (buf[i] == '%' ? 11 : buf[i] - '0'));
  return n;
}

/* imageproc pixelops::weighted_sum + Clamp<f32> for u8 */
static uint8_t blend_channel(uint8_t p, float color, float v) {
  const float lw = 1.0f - v;
  const float t = (float)p * lw + color * v;
  return t < 255.0f ? (t > 0.0f ? (uint8_t)t : 0) : 255;
}

static void draw_label(uint8_t* rgb, int w, int h, int64_t x, int64_t y, const uint8_t* chars, int n) {
  for (int k = 0; k < n; k++) { /* glyph by glyph in layout order: neighbouring boxes may share a pixel column */
    const UfdGlyph* g = &kUfdGlyphs[k][chars[k]];
    const float* cov = kUfdGlyphCoverage + g->off;
    for (int gy = 0; gy < g->h; gy++)
      for (int gx = 0; gx < g->w; gx++) {
        const int64_t ix = gx + g->x + x, iy = gy + g->y + y;
        if (ix < 0 || ix >= w || iy < 0 || iy >= h) continue;
        uint8_t* p = rgb + ((size_t)iy * w + (size_t)ix) * 3;
        const float v = cov[gy * g->w + gx];
        p[0] = blend_channel(p[0], 0.0f, v), p[1] = blend_channel(p[1], 255.0f, v), p[2] = blend_channel(p[2], 0.0f, v);
      }
  }
}

/* draw_bboxes_on_image (inferer.rs:58-92): per detection, in order, the hollow rectangle then its label */
void ufo_draw_labels(uint8_t* rgb, int w, int h, const ufo_det* dets, int n, float label_w, float label_h) {
  for (int i = 0; i < n; i++) {
    int64_t l, t, r, b;
    if (!ufo_rect_of_det(&dets[i], label_w, label_h, &l, &t, &r, &b)) continue; /* (the reference would have panicked) */
    ufo_draw_hollow_rects(rgb, w, h, &dets[i], 1, label_w, label_h);
    uint8_t chars[8];
    const int len = ufo_label_chars(dets[i].conf, chars);
    draw_label(rgb, w, h, l, t, chars, len); /* text origin = (x_tl as i32, y_tl as i32) = the rectangle's corner */
  }
}

/* ------------------------------------------------------------------------------------------------
 * tables */
static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
/* Annex K.1 quantisation tables, natural order (jcparam.c std_luminance_quant_tbl / std_chrominance_quant_tbl) */
static const uint8_t kStdLumQ[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                     14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                     18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                     49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t kStdChrQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                     24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
/* Annex K.3 Huffman tables (jcparam.c std_huff_tables) */
static const uint8_t kDcLumBits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t kDcChrBits[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t kDcVal[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kAcLumBits[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t kAcLumVal[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t kAcChrBits[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t kAcChrVal[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
/* jcdctmgr.c aanscales: scale factors of the AA&N fast DCT, 14 fractional bits, natural order */
static const int16_t kAanScales[64] = {
    16384, 22725, 21407, 19266, 16384, 12873, 8867,  4520,  22725, 31521, 29692, 26722, 22725, 17855, 12299, 6270,
    21407, 29692, 27969, 25172, 21407, 16819, 11585, 5906,  19266, 26722, 25172, 22654, 19266, 15137, 10426, 5315,
    16384, 22725, 21407, 19266, 16384, 12873, 8867,  4520,  12873, 17855, 16819, 15137, 12873, 10114, 6967,  3552,
    8867,  12299, 11585, 10426, 8867,  6967,  4799,  2446,  4520,  6270,  5906,  5315,  4520,  3552,  2446,  1247};

/* jpeg_set_quality(q, force_baseline = TRUE): jpeg_quality_scaling + jpeg_add_quant_table */
void ufo_jpeg_quant_table(int quality, int chroma, uint8_t out[64]) {
  if (quality <= 0) quality = 1;
  if (quality > 100) quality = 100;
  const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
  const uint8_t* base = chroma ? kStdChrQ : kStdLumQ;
  for (int i = 0; i < 64; i++) {
    long t = ((long)base[i] * scale + 50L) / 100L;
    if (t <= 0L) t = 1L;
    if (t > 255L) t = 255L;
    out[i] = (uint8_t)t;
  }
}

/* ------------------------------------------------------------------------------------------------
 * forward DCTs on (sample - 128), 16-bit elements as in the SIMD build of libjpeg-turbo */

/* jfdctfst.c with the arithmetic of its x86-64 SIMD twin (jfdctfst-sse2.asm), which is what a
 * turbojpeg build runs: MULTIPLY(v, c) = pmulhw(v << 2, c << 6) on 16-bit lanes, i.e.
 * ((int16)(v << 2) * (c << 6)) >> 16 -- equal to the C file's (v * c) >> 8 whenever v << 2 fits 16 bits. */
static int16_t ifast_mul(int16_t v, int c) {
  const int16_t pre = (int16_t)((uint16_t)v << 2);
  return (int16_t)(((int32_t)pre * (c << 6)) >> 16);
}
#define W16(x) ((int16_t)(x))
static void fdct_ifast_1d(int16_t* d, int stride) {
  const int16_t tmp0 = W16(d[0] + d[7 * stride]), tmp7 = W16(d[0] - d[7 * stride]);
  const int16_t tmp1 = W16(d[1 * stride] + d[6 * stride]), tmp6 = W16(d[1 * stride] - d[6 * stride]);
  const int16_t tmp2 = W16(d[2 * stride] + d[5 * stride]), tmp5 = W16(d[2 * stride] - d[5 * stride]);
  const int16_t tmp3 = W16(d[3 * stride] + d[4 * stride]), tmp4 = W16(d[3 * stride] - d[4 * stride]);
  /* even part */
  int16_t tmp10 = W16(tmp0 + tmp3), tmp13 = W16(tmp0 - tmp3), tmp11 = W16(tmp1 + tmp2), tmp12 = W16(tmp1 - tmp2);
  d[0] = W16(tmp10 + tmp11);
  d[4 * stride] = W16(tmp10 - tmp11);
  const int16_t z1 = ifast_mul(W16(tmp12 + tmp13), 181); /* FIX_0_707106781 */
  d[2 * stride] = W16(tmp13 + z1);
  d[6 * stride] = W16(tmp13 - z1);
  /* odd part */
  tmp10 = W16(tmp4 + tmp5), tmp11 = W16(tmp5 + tmp6), tmp12 = W16(tmp6 + tmp7);
  const int16_t z5 = ifast_mul(W16(tmp10 - tmp12), 98);          /* FIX_0_382683433 */
  const int16_t z2 = W16(ifast_mul(tmp10, 139) + z5);             /* FIX_0_541196100 */
  const int16_t z4 = W16(ifast_mul(tmp12, 334) + z5);             /* FIX_1_306562965 */
  const int16_t z3 = ifast_mul(tmp11, 181);
  const int16_t z11 = W16(tmp7 + z3), z13 = W16(tmp7 - z3);
  d[5 * stride] = W16(z13 + z2);
  d[3 * stride] = W16(z13 - z2);
  d[1 * stride] = W16(z11 + z4);
  d[7 * stride] = W16(z11 - z4);
}
static void fdct_ifast(int16_t* blk) {
  for (int r = 0; r < 8; r++) fdct_ifast_1d(blk + 8 * r, 1);
  for (int c = 0; c < 8; c++) fdct_ifast_1d(blk + c, 8);
}

/* jfdctint.c (accurate integer DCT): CONST_BITS 13, PASS1_BITS 2, output scaled up by 8 */
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
static void fdct_islow(int16_t* blk) {
  enum { C0298 = 2446, C0390 = 3196, C0541 = 4433, C0765 = 6270, C0899 = 7373, C1175 = 9633, C1501 = 12299,
         C1847 = 15137, C1961 = 16069, C2053 = 16819, C2562 = 20995, C3072 = 25172 };
  for (int pass = 0; pass < 2; pass++) {
    for (int i = 0; i < 8; i++) {
      int16_t* d = pass == 0 ? blk + 8 * i : blk + i;
      const int s = pass == 0 ? 1 : 8;
      int32_t tmp0 = d[0] + d[7 * s], tmp7 = d[0] - d[7 * s], tmp1 = d[s] + d[6 * s], tmp6 = d[s] - d[6 * s];
      int32_t tmp2 = d[2 * s] + d[5 * s], tmp5 = d[2 * s] - d[5 * s], tmp3 = d[3 * s] + d[4 * s], tmp4 = d[3 * s] - d[4 * s];
      const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      const int sh = pass == 0 ? 13 - 2 : 13 + 2;
      if (pass == 0) {
        d[0] = (int16_t)((tmp10 + tmp11) * 4);
        d[4 * s] = (int16_t)((tmp10 - tmp11) * 4);
      } else {
        d[0] = (int16_t)DESCALE(tmp10 + tmp11, 2);
        d[4 * s] = (int16_t)DESCALE(tmp10 - tmp11, 2);
      }
      int32_t z1 = (tmp12 + tmp13) * C0541;
      d[2 * s] = (int16_t)DESCALE(z1 + tmp13 * C0765, sh);
      d[6 * s] = (int16_t)DESCALE(z1 + tmp12 * (-C1847), sh);
      z1 = tmp4 + tmp7;
      int32_t z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
      const int32_t z5 = (z3 + z4) * C1175;
      tmp4 *= C0298, tmp5 *= C2053, tmp6 *= C3072, tmp7 *= C1501;
      z1 *= -C0899, z2 *= -C2562, z3 *= -C1961, z4 *= -C0390;
      z3 += z5, z4 += z5;
      d[7 * s] = (int16_t)DESCALE(tmp4 + z1 + z3, sh);
      d[5 * s] = (int16_t)DESCALE(tmp5 + z2 + z4, sh);
      d[3 * s] = (int16_t)DESCALE(tmp6 + z2 + z3, sh);
      d[1 * s] = (int16_t)DESCALE(tmp7 + z1 + z4, sh);
    }
  }
}

/* jcdctmgr.c: divisor of coefficient i, then compute_reciprocal's (reciprocal, correction, shift) and quantize():
 * q = ((|v| + corr) * recip) >> (16 + shift), sign restored */
typedef struct {
  uint16_t recip[64], corr[64];
  int shift[64]; /* total right shift r (recip has r fractional bits); divisor 1: recip 1, r 0 */
} quant_plan;

static void make_quant_plan(const uint8_t q[64], int ifast, quant_plan* p) {
  for (int i = 0; i < 64; i++) {
    uint32_t divisor = ifast ? (uint32_t)(((int32_t)q[i] * kAanScales[i] + (1 << 10)) >> 11) : (uint32_t)q[i] << 3;
    if (divisor == 1) {
      p->recip[i] = 1, p->corr[i] = 0, p->shift[i] = 0;
      continue;
    }
    int b = 0;
    while ((divisor >> (b + 1)) != 0) b++; /* flss(divisor) - 1 */
    int r = 16 + b;
    uint32_t fq = (uint32_t)((1ull << r) / divisor), fr = (uint32_t)((1ull << r) % divisor);
    uint32_t c = divisor / 2;
    if (fr == 0) {
      fq >>= 1, r--;
    } else if (fr <= divisor / 2) {
      c++;
    } else {
      fq++;
    }
    p->recip[i] = (uint16_t)fq, p->corr[i] = (uint16_t)c, p->shift[i] = r;
  }
}
static int16_t quantize1(int16_t v, const quant_plan* p, int i) {
  const uint32_t a = (uint16_t)(v < 0 ? -v : v);
  const uint32_t prod = (uint32_t)(uint16_t)(a + p->corr[i]) * p->recip[i];
  const int16_t t = (int16_t)(prod >> p->shift[i]);
  return v < 0 ? (int16_t)-t : t;
}

/* ------------------------------------------------------------------------------------------------
 * jchuff.c */
typedef struct {
  uint16_t code[256];
  uint8_t size[256];
} enc_huff;
static void make_enc_huff(enc_huff* h, const uint8_t bits[17], const uint8_t* vals) {
  memset(h, 0, sizeof(*h));
  int code = 0, k = 0;
  for (int l = 1; l <= 16; l++) {
    for (int i = 0; i < bits[l]; i++, k++, code++) h->code[vals[k]] = (uint16_t)code, h->size[vals[k]] = (uint8_t)l;
    code <<= 1;
  }
}
typedef struct {
  uint8_t* out;
  size_t cap, len;
  uint64_t acc; /* bit accumulator, MSB first */
  int nacc;
  int overflow;
} bitw;
static void put_byte(bitw* w, int b) {
  if (w->len < w->cap) w->out[w->len] = (uint8_t)b;
  else w->overflow = 1;
  w->len++;
}
static void put_bits(bitw* w, unsigned code, int size) {
  w->acc = (w->acc << size) | (code & ((1u << size) - 1));
  w->nacc += size;
  while (w->nacc >= 8) {
    const int b = (int)((w->acc >> (w->nacc - 8)) & 0xFF);
    put_byte(w, b);
    if (b == 0xFF) put_byte(w, 0);
    w->nacc -= 8;
  }
}
static int nbits_of(int v) {
  int n = 0;
  if (v < 0) v = -v;
  while (v) n++, v >>= 1;
  return n;
}
static void encode_block(bitw* w, const int16_t* blk /* natural order */, int* last_dc, const enc_huff* dc, const enc_huff* ac) {
  int diff = blk[0] - *last_dc;
  *last_dc = blk[0];
  int nb = nbits_of(diff);
  put_bits(w, dc->code[nb], dc->size[nb]);
  if (nb) put_bits(w, (unsigned)(diff < 0 ? diff - 1 : diff), nb);
  int run = 0;
  for (int k = 1; k < 64; k++) {
    const int v = blk[kZigzag[k]];
    if (v == 0) {
      run++;
      continue;
    }
    while (run > 15) put_bits(w, ac->code[0xF0], ac->size[0xF0]), run -= 16;
    nb = nbits_of(v);
    const int sym = (run << 4) + nb;
    put_bits(w, ac->code[sym], ac->size[sym]);
    put_bits(w, (unsigned)(v < 0 ? v - 1 : v), nb);
    run = 0;
  }
  if (run > 0) put_bits(w, ac->code[0], ac->size[0]);
}

/* ------------------------------------------------------------------------------------------------
 * jcmarker.c: SOI, JFIF APP0, DQT x2, SOF0, DHT x4, SOS */
static void put_marker_bytes(bitw* w, const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; i++) put_byte(w, p[i]);
}
static void write_headers(bitw* w, int width, int height, const uint8_t ql[64], const uint8_t qc[64]) {
  static const uint8_t soi_app0[] = {0xFF, 0xD8, 0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
  put_marker_bytes(w, soi_app0, sizeof(soi_app0));
  for (int t = 0; t < 2; t++) {
    const uint8_t hd[] = {0xFF, 0xDB, 0, 67, (uint8_t)t};
    put_marker_bytes(w, hd, sizeof(hd));
    for (int i = 0; i < 64; i++) put_byte(w, (t ? qc : ql)[kZigzag[i]]);
  }
  const uint8_t sof[] = {0xFF, 0xC0, 0, 17, 8, (uint8_t)(height >> 8), (uint8_t)height, (uint8_t)(width >> 8), (uint8_t)width,
                         3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1};
  put_marker_bytes(w, sof, sizeof(sof));
  const struct {
    int id;
    const uint8_t *bits, *vals;
  } dht[4] = {{0x00, kDcLumBits, kDcVal}, {0x10, kAcLumBits, kAcLumVal}, {0x01, kDcChrBits, kDcVal}, {0x11, kAcChrBits, kAcChrVal}};
  for (int t = 0; t < 4; t++) {
    int n = 0;
    for (int l = 1; l <= 16; l++) n += dht[t].bits[l];
    const uint8_t hd[] = {0xFF, 0xC4, (uint8_t)((n + 19) >> 8), (uint8_t)(n + 19), (uint8_t)dht[t].id};
    put_marker_bytes(w, hd, sizeof(hd));
    put_marker_bytes(w, dht[t].bits + 1, 16);
    put_marker_bytes(w, dht[t].vals, (size_t)n);
  }
  static const uint8_t sos[] = {0xFF, 0xDA, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};
  put_marker_bytes(w, sos, sizeof(sos));
}

/* ------------------------------------------------------------------------------------------------
 * the encoder */
size_t ufo_jpeg_encode_bound(int w, int h) {
  const size_t mcus = (size_t)((w + 15) / 16) * (size_t)((h + 15) / 16);
  return 1024 + mcus * 6 * 64 * 4; /* generous: <= 26 bits + stuffing per coefficient */
}

/* quantised coefficients of the frame in MCU order: [mcu][6 blocks: Y00 Y01 Y10 Y11 Cb Cr][64 natural order] */
static int frame_coefficients(const uint8_t* rgb, int w, int h, const uint8_t ql[64], const uint8_t qc[64], int ifast, int16_t* coef) {
  const int mx = (w + 15) / 16, my = (h + 15) / 16;
  const int yw = mx * 16, yh = my * 16, cw = mx * 8, ch = my * 8;
  const int ybw = (w + 7) / 8, ybh = (h + 7) / 8; /* width_in_blocks / height_in_blocks of Y */
  uint8_t* Y = (uint8_t*)malloc((size_t)yw * yh);
  uint8_t* C[2] = {(uint8_t*)malloc((size_t)cw * ch), (uint8_t*)malloc((size_t)cw * ch)};
  uint8_t* F[2] = {(uint8_t*)malloc((size_t)yw * 2), (uint8_t*)malloc((size_t)yw * 2)}; /* one row group of full-size Cb, Cr */
  if (!Y || !C[0] || !C[1] || !F[0] || !F[1]) {
    free(Y), free(C[0]), free(C[1]), free(F[0]), free(F[1]);
    return UFO_E_ARG;
  }
  /* jccolor.c rgb_ycc_convert: 16-bit fixed point, FIX(x) = (int)(x * 65536 + 0.5) */
  const int32_t half = 1 << 15, off = 128 << 16;
  const int groups = (h + 1) / 2;
  for (int g = 0; g < groups; g++) {
    for (int k = 0; k < 2; k++) {
      const int y = 2 * g + k < h ? 2 * g + k : h - 1; /* jcprepct.c: the conversion buffer's missing row repeats the last one */
      const uint8_t* p = rgb + (size_t)y * w * 3;
      uint8_t* yo = Y + (size_t)(2 * g + k) * yw;
      for (int x = 0; x < yw; x++) {
        const int xs = x < w ? x : w - 1; /* jcsample.c expand_right_edge */
        const int32_t r = p[3 * xs], gg = p[3 * xs + 1], b = p[3 * xs + 2];
        yo[x] = (uint8_t)((19595 * r + 38470 * gg + 7471 * b + half) >> 16);
        F[0][k * yw + x] = (uint8_t)((-11059 * r - 21709 * gg + 32768 * b + off + half - 1) >> 16);
        F[1][k * yw + x] = (uint8_t)((32768 * r - 27439 * gg - 5329 * b + off + half - 1) >> 16);
      }
    }
    for (int c = 0; c < 2; c++) { /* jcsample.c h2v2_downsample: bias 1, 2, 1, 2, ... */
      uint8_t* o = C[c] + (size_t)g * cw;
      for (int x = 0; x < cw; x++)
        o[x] = (uint8_t)((F[c][2 * x] + F[c][2 * x + 1] + F[c][yw + 2 * x] + F[c][yw + 2 * x + 1] + (x & 1 ? 2 : 1)) >> 2);
    }
  }
  /* jcprepct.c: the rest of the last iMCU row repeats the last row of every component */
  for (int y = 2 * groups; y < yh; y++) memcpy(Y + (size_t)y * yw, Y + (size_t)(2 * groups - 1) * yw, (size_t)yw);
  for (int c = 0; c < 2; c++)
    for (int y = groups; y < ch; y++) memcpy(C[c] + (size_t)y * cw, C[c] + (size_t)(groups - 1) * cw, (size_t)cw);

  quant_plan pl, pc;
  make_quant_plan(ql, ifast, &pl);
  make_quant_plan(qc, ifast, &pc);
  for (int my_ = 0; my_ < my; my_++) {
    for (int mx_ = 0; mx_ < mx; mx_++) {
      int16_t* mcu = coef + ((size_t)my_ * mx + mx_) * 6 * 64;
      for (int blk = 0; blk < 6; blk++) {
        int16_t* o = mcu + blk * 64;
        const uint8_t* src;
        int pitch;
        const quant_plan* qp = blk < 4 ? &pl : &pc;
        if (blk < 4) {
          const int bx = 2 * mx_ + (blk & 1), by = 2 * my_ + (blk >> 1);
          /* jccoefct.c compress_data dummy blocks: beyond the component's block grid, AC = 0 and DC = the DC of
           * the block before (right edge: the block to the left; bottom edge: the last block of the row above) */
          if (by >= ybh) {
            memset(o, 0, 128);
            o[0] = mcu[1 * 64];
            continue;
          }
          if (bx >= ybw) {
            memset(o, 0, 128);
            o[0] = o[-64];
            continue;
          }
          src = Y + (size_t)by * 8 * yw + bx * 8, pitch = yw;
        } else {
          src = C[blk - 4] + (size_t)my_ * 8 * cw + mx_ * 8, pitch = cw;
        }
        int16_t ws[64];
        for (int y = 0; y < 8; y++)
          for (int x = 0; x < 8; x++) ws[8 * y + x] = (int16_t)(src[(size_t)y * pitch + x] - 128);
        if (ifast) fdct_ifast(ws);
        else fdct_islow(ws);
        for (int i = 0; i < 64; i++) o[i] = quantize1(ws[i], qp, i);
      }
    }
  }
  free(Y), free(C[0]), free(C[1]), free(F[0]), free(F[1]);
  return UFO_OK;
}

int ufo_jpeg_encode_coefficients(const uint8_t* rgb, int w, int h, int quality, int dct, int16_t* coef) {
  if (!rgb || !coef || w < 1 || h < 1 || w > 65535 || h > 65535) return UFO_E_ARG;
  uint8_t ql[64], qc[64];
  ufo_jpeg_quant_table(quality, 0, ql);
  ufo_jpeg_quant_table(quality, 1, qc);
  return frame_coefficients(rgb, w, h, ql, qc, dct == 1, coef);
}

int ufo_jpeg_encode_rgb(const uint8_t* rgb, int w, int h, int quality, int dct, uint8_t* out, size_t cap, size_t* len) {
  if (!rgb || !out || !len || w < 1 || h < 1 || w > 65535 || h > 65535) return UFO_E_ARG;
  if (dct < 0) dct = quality >= 96 ? 0 : 1; /* turbojpeg.c setCompDefaults: JDCT_ISLOW from quality 96 on, else JDCT_FASTEST */
  uint8_t ql[64], qc[64];
  ufo_jpeg_quant_table(quality, 0, ql);
  ufo_jpeg_quant_table(quality, 1, qc);
  const int mx = (w + 15) / 16, my = (h + 15) / 16;
  int16_t* coef = (int16_t*)malloc((size_t)mx * my * 6 * 64 * sizeof(int16_t));
  if (!coef) return UFO_E_ARG;
  int rc = frame_coefficients(rgb, w, h, ql, qc, dct == 1, coef);
  if (rc) {
    free(coef);
    return rc;
  }
  enc_huff dcl, acl, dcc, acc;
  make_enc_huff(&dcl, kDcLumBits, kDcVal);
  make_enc_huff(&acl, kAcLumBits, kAcLumVal);
  make_enc_huff(&dcc, kDcChrBits, kDcVal);
  make_enc_huff(&acc, kAcChrBits, kAcChrVal);
  bitw bw = {out, cap, 0, 0, 0, 0};
  write_headers(&bw, w, h, ql, qc);
  int last_dc[3] = {0, 0, 0};
  for (size_t m = 0; m < (size_t)mx * my; m++) {
    const int16_t* mcu = coef + m * 6 * 64;
    for (int b = 0; b < 4; b++) encode_block(&bw, mcu + b * 64, &last_dc[0], &dcl, &acl);
    encode_block(&bw, mcu + 4 * 64, &last_dc[1], &dcc, &acc);
    encode_block(&bw, mcu + 5 * 64, &last_dc[2], &dcc, &acc);
  }
  if (bw.nacc) put_bits(&bw, 0x7F, 8 - bw.nacc); /* jchuff.c flush_bits: pad the last byte with ones */
  put_byte(&bw, 0xFF);
  put_byte(&bw, 0xD9);
  free(coef);
  *len = bw.len;
  return bw.overflow ? UFO_E_ARG : UFO_OK;
}

/* as_jpeg_stream_item (lib.rs:48-57) */
size_t ufo_stream_item(const uint8_t* jpeg, size_t len, uint8_t* out, size_t cap) {
  static const char head[] = "--frame\r\nContent-Type: image/jpeg\r\n\r\n";
  static const char tail[] = "\r\n\r\n";
  const size_t hl = sizeof(head) - 1, tl = sizeof(tail) - 1, total = hl + len + tl;
  if (out && cap >= total) {
    memcpy(out, head, hl);
    memcpy(out + hl, jpeg, len);
    memcpy(out + hl + len, tail, tl);
  }
  return total;
}

/* Inferer::run, inferer.rs:35-40: decompress_image -> infer_faces -> draw_bboxes_on_image ->
 * compress_image(95, Sub2x2).  label_w / label_h are the slot's width / height (router.rs:66-67: 1280 x 720). */
int ufo_annotate_encode_jpeg(const uint8_t* jpeg, size_t len, int model_w, int model_h, const float* weights,
                             const float* priors, float min_confidence, float max_iou, float label_w, float label_h, int quality,
                             ufo_det* dets, int cap, uint8_t* out, size_t out_cap, size_t* out_len) {
  ufo_jpeg_info info;
  int rc = ufo_jpeg_probe(jpeg, len, &info);
  if (rc) return rc;
  uint8_t* rgb = (uint8_t*)malloc((size_t)info.width * info.height * 3);
  if (!rgb) return UFO_E_ARG;
  rc = ufo_jpeg_decode_rgb(jpeg, len, rgb, info.width, info.height);
  int n = 0;
  if (rc == UFO_OK) {
    n = ufo_infer_rgb(rgb, info.width, info.height, model_w, model_h, weights, priors, min_confidence, max_iou, dets, cap);
    if (n < 0) rc = n;
  }
  if (rc == UFO_OK) {
    ufo_draw_labels(rgb, info.width, info.height, dets, n < cap ? n : cap, label_w, label_h);
    rc = ufo_jpeg_encode_rgb(rgb, info.width, info.height, quality, -1, out, out_cap, out_len);
  }
  free(rgb);
  return rc ? rc : n;
}
