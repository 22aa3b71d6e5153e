/*
 * jpeg_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see ufd_oracle.h) for row A1:
 *   turbojpeg::decompress_image::<Rgb<u8>>(&[u8])      (infer_server/src/inferer.rs:35)
 *
 * The arithmetic lives in a third-party dependency that is not under /root/reference:
 * turbojpeg 0.5.2 / turbojpeg-sys 0.2.2 (Cargo.lock:2617-2640), i.e. libjpeg-turbo driven
 * through tjDecompress2(flags = 0): accurate integer IDCT (JDCT_ISLOW), fancy (triangle)
 * chroma upsampling, libjpeg fixed-point YCbCr->RGB.  This file restates the published
 * libjpeg algorithms (ITU T.81 entropy decoding, jidctint "islow" IDCT, jdsample fancy
 * h2v1/h2v2 upsampling, jdcolor ycc_rgb tables).  It is pinned bit-exact against the
 * libjpeg-turbo build bundled with PIL (tests/golden/jpeg_*.npz, tests/test_oracle_jpeg.py).
 *
 * Supported: 8-bit baseline/extended-sequential Huffman (SOF0/SOF1) and progressive Huffman
 * (SOF2; the reference's own test pictures, tests/integration_tests.rs:20-31), 1 or 3
 * components, every sampling layout jdsample.c takes (integral expansion factors 1..4: fancy
 * h2v1 / h2v2 / h1v2, plain replication for everything else, e.g. 4:1:1 and 4:1:0; at most 10
 * blocks per interleaved MCU, jdinput.c per_scan_setup), YCbCr / RGB / grey colour spaces by
 * libjpeg's marker rules, restart intervals, Annex-K default tables for DHT-less MJPEG.
 * Pinned on all of these by streams libjpeg-turbo wrote and decoded: tests/golden/jpeg_layouts.npz.
 */
#include <stdlib.h>
#include <string.h>

#include "ufd_oracle.h"

/* ------------------------------------------------------------------------------------------- */
static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* Annex K.3 default Huffman tables (MJPEG streams may omit DHT) */
static const uint8_t kDcLumBits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t kDcLumVal[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kDcChrBits[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t kDcChrVal[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
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

/* ------------------------------------------------------------------------------------------- */
typedef struct {
  int present;
  uint8_t vals[256];
  int maxcode[18]; /* largest code of length l, -1 if none; [17] sentinel */
  int valoff[17];  /* vals index offset for length l */
  /* 9-bit lookahead: (nbits << 8) | symbol, 0 = needs slow path */
  uint16_t look[512];
} huff_t;

static int build_huff(huff_t* h, const uint8_t* bits /*[17], [0] unused*/, const uint8_t* vals, int nvals) {
  int total = 0;
  for (int l = 1; l <= 16; l++) total += bits[l];
  if (total > 256 || total != nvals) return UFO_E_DECODE;
  memcpy(h->vals, vals, (size_t)total);
  int code = 0, k = 0;
  memset(h->look, 0, sizeof(h->look));
  for (int l = 1; l <= 16; l++) {
    h->valoff[l] = k - code;
    if (bits[l]) {
      for (int i = 0; i < bits[l]; i++, k++, code++) {
        if (l <= 9) {
          int lo = code << (9 - l);
          for (int j = 0; j < (1 << (9 - l)); j++) h->look[lo + j] = (uint16_t)((l << 8) | vals[k]);
        }
      }
      h->maxcode[l] = code - 1;
      if (code > (1 << l)) return UFO_E_DECODE;
    } else {
      h->maxcode[l] = -1;
    }
    code <<= 1;
  }
  h->maxcode[17] = 0x7fffffff;
  h->present = 1;
  return UFO_OK;
}

typedef struct {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t acc;
  int nbits;
  int marker;   /* pending marker byte found in the entropy segment (0 = none) */
  int overrun;  /* bits consumed past the end of data / marker */
} bitrd;

static void br_init(bitrd* b, const uint8_t* p, const uint8_t* end) {
  b->p = p;
  b->end = end;
  b->acc = 0;
  b->nbits = 0;
  b->marker = 0;
  b->overrun = 0;
}

static void br_fill(bitrd* b) {
  while (b->nbits <= 56) {
    unsigned c = 0;
    if (b->marker == 0 && b->p < b->end) {
      c = *b->p++;
      if (c == 0xFF) {
        /* skip fill bytes */
        while (b->p < b->end && *b->p == 0xFF) b->p++;
        if (b->p >= b->end) {
          b->marker = 0xD9; /* treat as EOI */
          c = 0;
          b->overrun += 8;
        } else if (*b->p == 0) {
          b->p++; /* stuffed zero */
        } else {
          b->marker = *b->p++;
          c = 0;
          b->overrun += 8;
        }
      }
    } else {
      if (b->marker == 0) b->marker = 0xD9;
      b->overrun += 8;
    }
    b->acc |= (uint64_t)c << (56 - b->nbits);
    b->nbits += 8;
  }
}

static inline int br_peek(bitrd* b, int n) { return (int)(b->acc >> (64 - n)); }
static inline void br_skip(bitrd* b, int n) {
  b->acc <<= n;
  b->nbits -= n;
}
static inline int br_get(bitrd* b, int n) {
  if (n == 0) return 0;
  if (b->nbits < n) br_fill(b);
  int v = br_peek(b, n);
  br_skip(b, n);
  return v;
}
/* bits consumed beyond real data? (overrun counts zero bytes appended; some may still be unread) */
static int br_bad(const bitrd* b) { return b->overrun > 0 && b->nbits < b->overrun; }

static inline int huff_decode(bitrd* b, const huff_t* h) {
  if (b->nbits < 16) br_fill(b);
  int look = h->look[br_peek(b, 9)];
  if (look) {
    br_skip(b, look >> 8);
    return look & 0xFF;
  }
  int code = br_peek(b, 16);
  for (int l = 10; l <= 16; l++) {
    int c = code >> (16 - l);
    if (c <= h->maxcode[l]) {
      br_skip(b, l);
      return h->vals[(h->valoff[l] + c) & 0xFF];
    }
  }
  return -1;
}

static inline int extend(int r, int s) { return r < (1 << (s - 1)) ? r + (int)(((unsigned)-1) << s) + 1 : r; }

/* ------------------------------------------------------------------------------------------- */
typedef struct {
  int id, h, v, tq;
  int wblk, hblk;   /* allocated blocks (padded to MCU multiples) */
  int dw, dh;       /* downsampled_width / downsampled_height (libjpeg naming) */
  int16_t* coef;    /* [hblk][wblk][64], natural order, un-dequantized */
  uint8_t* plane;   /* [hblk*8][wblk*8] samples after IDCT */
} comp_t;

typedef struct {
  int width, height, ncomp, progressive;
  int hmax, vmax, mcux, mcuy;
  comp_t comp[4];
  uint16_t qt[4][64]; /* natural order */
  int qt_present[4];
  huff_t dc[4], ac[4];
  int restart_interval;
  int saw_jfif, saw_adobe, adobe_transform;
  int saw_sof;
} dec_t;

static void dec_free(dec_t* d) {
  for (int i = 0; i < 4; i++) {
    free(d->comp[i].coef);
    free(d->comp[i].plane);
    d->comp[i].coef = NULL;
    d->comp[i].plane = NULL;
  }
}

static int parse_dqt(dec_t* d, const uint8_t* p, int len) {
  while (len > 0) {
    int pq = p[0] >> 4, tq = p[0] & 15;
    if (tq > 3 || pq > 1) return UFO_E_DECODE;
    int need = 1 + 64 * (pq ? 2 : 1);
    if (len < need) return UFO_E_DECODE;
    for (int i = 0; i < 64; i++) {
      int v = pq ? ((p[1 + 2 * i] << 8) | p[2 + 2 * i]) : p[1 + i];
      d->qt[tq][kZigzag[i]] = (uint16_t)v;
    }
    d->qt_present[tq] = 1;
    p += need;
    len -= need;
  }
  return UFO_OK;
}

static int parse_dht(dec_t* d, const uint8_t* p, int len) {
  while (len > 0) {
    if (len < 17) return UFO_E_DECODE;
    int tc = p[0] >> 4, th = p[0] & 15;
    if (tc > 1 || th > 3) return UFO_E_DECODE;
    uint8_t bits[17];
    bits[0] = 0;
    int total = 0;
    for (int i = 1; i <= 16; i++) {
      bits[i] = p[i];
      total += p[i];
    }
    if (total > 256 || len < 17 + total) return UFO_E_DECODE;
    int rc = build_huff(tc ? &d->ac[th] : &d->dc[th], bits, p + 17, total);
    if (rc) return rc;
    p += 17 + total;
    len -= 17 + total;
  }
  return UFO_OK;
}

static int parse_sof(dec_t* d, const uint8_t* p, int len, int progressive) {
  if (len < 6) return UFO_E_DECODE;
  if (p[0] != 8) return UFO_E_UNSUPPORTED;
  d->height = (p[1] << 8) | p[2];
  d->width = (p[3] << 8) | p[4];
  d->ncomp = p[5];
  d->progressive = progressive;
  if (d->width <= 0 || d->height <= 0) return UFO_E_DECODE;
  if (d->ncomp != 1 && d->ncomp != 3) return UFO_E_UNSUPPORTED;
  if (len < 6 + 3 * d->ncomp) return UFO_E_DECODE;
  d->hmax = d->vmax = 1;
  for (int i = 0; i < d->ncomp; i++) {
    comp_t* c = &d->comp[i];
    c->id = p[6 + 3 * i];
    c->h = p[7 + 3 * i] >> 4;
    c->v = p[7 + 3 * i] & 15;
    c->tq = p[8 + 3 * i];
    if (c->h < 1 || c->h > 4 || c->v < 1 || c->v > 4 || c->tq > 3) return UFO_E_DECODE;
    if (c->h > d->hmax) d->hmax = c->h;
    if (c->v > d->vmax) d->vmax = c->v;
  }
  d->mcux = (d->width + 8 * d->hmax - 1) / (8 * d->hmax);
  d->mcuy = (d->height + 8 * d->vmax - 1) / (8 * d->vmax);
  for (int i = 0; i < d->ncomp; i++) {
    comp_t* c = &d->comp[i];
    c->wblk = d->mcux * c->h;
    c->hblk = d->mcuy * c->v;
    c->dw = (d->width * c->h + d->hmax - 1) / d->hmax;
    c->dh = (d->height * c->v + d->vmax - 1) / d->vmax;
    size_t n = (size_t)c->wblk * c->hblk * 64;
    c->coef = (int16_t*)calloc(n, sizeof(int16_t));
    c->plane = (uint8_t*)malloc(n);
    if (!c->coef || !c->plane) return UFO_E_ARG;
  }
  d->saw_sof = 1;
  return UFO_OK;
}

/* ---- entropy-coded scan ---- */
typedef struct {
  int ns;
  int ci[4];
  int td[4], ta[4];
  int ss, se, ah, al;
} scan_t;

static int decode_block_baseline(bitrd* b, const huff_t* dc, const huff_t* ac, int* pred, int16_t* blk) {
  int s = huff_decode(b, dc);
  if (s < 0 || s > 15) return UFO_E_DECODE;
  int diff = 0;
  if (s) diff = extend(br_get(b, s), s);
  *pred += diff;
  blk[0] = (int16_t)*pred;
  for (int k = 1; k < 64;) {
    int rs = huff_decode(b, ac);
    if (rs < 0) return UFO_E_DECODE;
    int r = rs >> 4;
    s = rs & 15;
    if (s) {
      k += r;
      if (k > 63) return UFO_E_DECODE;
      blk[kZigzag[k]] = (int16_t)extend(br_get(b, s), s);
      k++;
    } else {
      if (r != 15) break;
      k += 16;
    }
  }
  return UFO_OK;
}

static int decode_scan(dec_t* d, const scan_t* sc, const uint8_t* p, const uint8_t* end, const uint8_t** next) {
  bitrd b;
  br_init(&b, p, end);
  int pred[4] = {0, 0, 0, 0};
  int eobrun = 0;
  int rst_left = d->restart_interval;
  int next_rst = 0;
  int interleaved = sc->ns > 1;
  int nx, ny;
  if (interleaved) {
    nx = d->mcux;
    ny = d->mcuy;
  } else {
    comp_t* c = &d->comp[sc->ci[0]];
    nx = (c->dw + 7) / 8;
    ny = (c->dh + 7) / 8;
  }
  if (interleaved) { /* jdinput.c per_scan_setup: JERR_BAD_MCU_SIZE above D_MAX_BLOCKS_IN_MCU = 10 */
    int blocks = 0;
    for (int i = 0; i < sc->ns; i++) blocks += d->comp[sc->ci[i]].h * d->comp[sc->ci[i]].v;
    if (blocks > 10) return UFO_E_DECODE;
  }
  for (int i = 0; i < sc->ns; i++) {
    if (!d->progressive || sc->ss == 0) {
      if (sc->ah == 0 && !d->dc[sc->td[i]].present) return UFO_E_DECODE;
    }
    if (!d->progressive || sc->ss > 0) {
      if (!d->ac[sc->ta[i]].present) return UFO_E_DECODE;
    }
  }
  for (int my = 0; my < ny; my++) {
    for (int mx = 0; mx < nx; mx++) {
      if (d->restart_interval && rst_left == 0) {
        /* expect RSTn: discard remaining bits, find marker */
        if (b.marker == 0) {
          /* scan forward for the marker */
          const uint8_t* q = b.p;
          while (q + 1 < b.end && !(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF)) q++;
          if (q + 1 >= b.end) return UFO_E_DECODE;
          b.marker = q[1];
          b.p = q + 2;
        }
        if (b.marker != 0xD0 + next_rst) return UFO_E_DECODE;
        next_rst = (next_rst + 1) & 7;
        const uint8_t* q = b.p;
        br_init(&b, q, end);
        pred[0] = pred[1] = pred[2] = pred[3] = 0;
        eobrun = 0;
        rst_left = d->restart_interval;
      }
      for (int i = 0; i < sc->ns; i++) {
        comp_t* c = &d->comp[sc->ci[i]];
        int bw = interleaved ? c->h : 1, bh = interleaved ? c->v : 1;
        for (int by = 0; by < bh; by++) {
          for (int bx = 0; bx < bw; bx++) {
            int row = interleaved ? my * c->v + by : my;
            int col = interleaved ? mx * c->h + bx : mx;
            int16_t* blk = c->coef + ((size_t)row * c->wblk + col) * 64;
            if (!d->progressive) {
              int rc = decode_block_baseline(&b, &d->dc[sc->td[i]], &d->ac[sc->ta[i]], &pred[i], blk);
              if (rc) return rc;
            } else if (sc->ss == 0) {
              if (sc->ah == 0) { /* DC first */
                int s = huff_decode(&b, &d->dc[sc->td[i]]);
                if (s < 0 || s > 15) return UFO_E_DECODE;
                int diff = s ? extend(br_get(&b, s), s) : 0;
                pred[i] += diff;
                blk[0] = (int16_t)(pred[i] * (1 << sc->al));
              } else { /* DC refine */
                if (br_get(&b, 1)) blk[0] |= (int16_t)(1 << sc->al);
              }
            } else {
              const huff_t* ac = &d->ac[sc->ta[i]];
              if (sc->ah == 0) { /* AC first */
                if (eobrun > 0) {
                  eobrun--;
                } else {
                  for (int k = sc->ss; k <= sc->se; k++) {
                    int rs = huff_decode(&b, ac);
                    if (rs < 0) return UFO_E_DECODE;
                    int r = rs >> 4, s = rs & 15;
                    if (s) {
                      k += r;
                      if (k > 63) return UFO_E_DECODE;
                      blk[kZigzag[k]] = (int16_t)(extend(br_get(&b, s), s) * (1 << sc->al));
                    } else {
                      if (r == 15) {
                        k += 15;
                      } else {
                        eobrun = 1 << r;
                        if (r) eobrun += br_get(&b, r);
                        eobrun--;
                        break;
                      }
                    }
                  }
                }
              } else { /* AC refine */
                int p1 = 1 << sc->al, m1 = -(1 << sc->al);
                int k = sc->ss;
                if (eobrun == 0) {
                  for (; k <= sc->se; k++) {
                    int rs = huff_decode(&b, ac);
                    if (rs < 0) return UFO_E_DECODE;
                    int r = rs >> 4, s = rs & 15;
                    if (s) {
                      s = br_get(&b, 1) ? p1 : m1;
                    } else if (r != 15) {
                      eobrun = 1 << r;
                      if (r) eobrun += br_get(&b, r);
                      break;
                    }
                    do {
                      int16_t* co = blk + kZigzag[k];
                      if (*co != 0) {
                        if (br_get(&b, 1)) {
                          if ((*co & p1) == 0) *co = (int16_t)(*co >= 0 ? *co + p1 : *co + m1);
                        }
                      } else {
                        if (--r < 0) break;
                      }
                      k++;
                    } while (k <= sc->se);
                    if (s) {
                      if (k > 63) return UFO_E_DECODE;
                      blk[kZigzag[k]] = (int16_t)s;
                    }
                  }
                }
                if (eobrun > 0) {
                  for (; k <= sc->se; k++) {
                    int16_t* co = blk + kZigzag[k];
                    if (*co != 0) {
                      if (br_get(&b, 1)) {
                        if ((*co & p1) == 0) *co = (int16_t)(*co >= 0 ? *co + p1 : *co + m1);
                      }
                    }
                  }
                  eobrun--;
                }
              }
            }
          }
        }
      }
      if (br_bad(&b)) return UFO_E_DECODE;
      if (d->restart_interval) rst_left--;
    }
  }
  /* next marker segment: first 0xFF followed by a byte that is not stuffing, fill or RSTn */
  {
    const uint8_t* q = p;
    while (q + 1 < end && !(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
    *next = (q + 1 < end) ? q : end;
  }
  return UFO_OK;
}

/* ---- jidctint.c "islow" 8x8 IDCT with dequantisation ---- */
#define CONST_BITS 13
#define PASS1_BITS 2
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

static inline uint8_t idct_range_limit(int x) {
  /* libjpeg post-IDCT table: index (x & 1023) into {128..255, 255 x384, 0 x384, 0..127} */
  int i = x & 1023;
  if (i < 128) return (uint8_t)(i + 128);
  if (i < 512) return 255;
  if (i < 896) return 0;
  return (uint8_t)(i - 896);
}

static void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int pitch) {
  int ws[64];
  for (int c = 0; c < 8; c++) {
    const int16_t* ip = in + c;
    const uint16_t* qp = q + c;
    int* wp = ws + c;
    if (ip[8] == 0 && ip[16] == 0 && ip[24] == 0 && ip[32] == 0 && ip[40] == 0 && ip[48] == 0 && ip[56] == 0) {
      int dc = (ip[0] * qp[0]) * (1 << PASS1_BITS);
      for (int r = 0; r < 8; r++) wp[8 * r] = dc;
      continue;
    }
    int z2 = ip[16] * qp[16], z3 = ip[48] * qp[48];
    int z1 = (z2 + z3) * FIX_0_541196100;
    int tmp2 = z1 + z3 * (-FIX_1_847759065);
    int tmp3 = z1 + z2 * FIX_0_765366865;
    z2 = ip[0] * qp[0];
    z3 = ip[32] * qp[32];
    int tmp0 = (z2 + z3) * (1 << CONST_BITS);
    int tmp1 = (z2 - z3) * (1 << CONST_BITS);
    int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = ip[56] * qp[56];
    tmp1 = ip[40] * qp[40];
    tmp2 = ip[24] * qp[24];
    tmp3 = ip[8] * qp[8];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    int z5 = (z3 + z4) * FIX_1_175875602;
    tmp0 *= FIX_0_298631336;
    tmp1 *= FIX_2_053119869;
    tmp2 *= FIX_3_072711026;
    tmp3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223;
    z2 *= -FIX_2_562915447;
    z3 *= -FIX_1_961570560;
    z4 *= -FIX_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    wp[0] = DESCALE(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
    wp[56] = DESCALE(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
    wp[8] = DESCALE(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
    wp[48] = DESCALE(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
    wp[16] = DESCALE(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
    wp[40] = DESCALE(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
    wp[24] = DESCALE(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
    wp[32] = DESCALE(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
  }
  for (int r = 0; r < 8; r++) {
    const int* wp = ws + 8 * r;
    uint8_t* op = out + (size_t)r * pitch;
    int z2 = wp[2], z3 = wp[6];
    int z1 = (z2 + z3) * FIX_0_541196100;
    int tmp2 = z1 + z3 * (-FIX_1_847759065);
    int tmp3 = z1 + z2 * FIX_0_765366865;
    int tmp0 = (wp[0] + wp[4]) * (1 << CONST_BITS);
    int tmp1 = (wp[0] - wp[4]) * (1 << CONST_BITS);
    int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = wp[7];
    tmp1 = wp[5];
    tmp2 = wp[3];
    tmp3 = wp[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    int z5 = (z3 + z4) * FIX_1_175875602;
    tmp0 *= FIX_0_298631336;
    tmp1 *= FIX_2_053119869;
    tmp2 *= FIX_3_072711026;
    tmp3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223;
    z2 *= -FIX_2_562915447;
    z3 *= -FIX_1_961570560;
    z4 *= -FIX_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    const int sh = CONST_BITS + PASS1_BITS + 3;
    op[0] = idct_range_limit(DESCALE(tmp10 + tmp3, sh));
    op[7] = idct_range_limit(DESCALE(tmp10 - tmp3, sh));
    op[1] = idct_range_limit(DESCALE(tmp11 + tmp2, sh));
    op[6] = idct_range_limit(DESCALE(tmp11 - tmp2, sh));
    op[2] = idct_range_limit(DESCALE(tmp12 + tmp1, sh));
    op[5] = idct_range_limit(DESCALE(tmp12 - tmp1, sh));
    op[3] = idct_range_limit(DESCALE(tmp13 + tmp0, sh));
    op[4] = idct_range_limit(DESCALE(tmp13 - tmp0, sh));
  }
}

/* ---- upsampling (jdsample.c): sample of component c at full-res pixel (x,y) ---- */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Returns the upsampled row `y` (full resolution) of component c into out[0..width). */
static void upsample_row(const dec_t* d, const comp_t* c, int y, uint8_t* out) {
  int hx = d->hmax / c->h, vx = d->vmax / c->v; /* expansion factors */
  int pitch = c->wblk * 8;
  int W = d->width;
  if (hx == 1 && vx == 1) {
    memcpy(out, c->plane + (size_t)y * pitch, (size_t)W);
    return;
  }
  int fancy_ok = c->dw > 2;
  if (hx == 2 && vx == 1 && d->hmax % c->h == 0) {
    const uint8_t* in = c->plane + (size_t)y * pitch;
    if (!fancy_ok) {
      for (int x = 0; x < W; x++) out[x] = in[x >> 1];
      return;
    }
    for (int x = 0; x < W; x++) {
      int i = x >> 1;
      int v;
      if (x & 1) {
        v = (i == c->dw - 1) ? in[i] : (in[i] * 3 + in[i + 1] + 2) >> 2;
      } else {
        v = (i == 0) ? in[i] : (in[i] * 3 + in[i - 1] + 1) >> 2;
      }
      out[x] = (uint8_t)v;
    }
    return;
  }
  if (hx == 2 && vx == 2) {
    int iy = y >> 1;
    if (!fancy_ok) {
      const uint8_t* in = c->plane + (size_t)iy * pitch;
      for (int x = 0; x < W; x++) out[x] = in[x >> 1];
      return;
    }
    int ny = (y & 1) ? iy + 1 : iy - 1; /* next-nearest row: below for odd output rows, above for even */
    ny = clampi(ny, 0, c->dh - 1);
    const uint8_t* in0 = c->plane + (size_t)iy * pitch;
    const uint8_t* in1 = c->plane + (size_t)ny * pitch;
    for (int x = 0; x < W; x++) {
      int i = x >> 1;
      int cur = in0[i] * 3 + in1[i];
      int v;
      if (x & 1) {
        if (i == c->dw - 1)
          v = (cur * 4 + 7) >> 4;
        else
          v = (cur * 3 + (in0[i + 1] * 3 + in1[i + 1]) + 7) >> 4;
      } else {
        if (i == 0)
          v = (cur * 4 + 8) >> 4;
        else
          v = (cur * 3 + (in0[i - 1] * 3 + in1[i - 1]) + 8) >> 4;
      }
      out[x] = (uint8_t)v;
    }
    return;
  }
  if (hx == 1 && vx == 2) {
    /* h1v2 fancy (libjpeg-turbo jdsample.c h1v2_fancy_upsample): 3/4 nearer + 1/4 further row */
    int iy = y >> 1;
    int ny = clampi((y & 1) ? iy + 1 : iy - 1, 0, c->dh - 1);
    const uint8_t* in0 = c->plane + (size_t)iy * pitch;
    const uint8_t* in1 = c->plane + (size_t)ny * pitch;
    int bias = (y & 1) ? 2 : 1;
    for (int x = 0; x < W; x++) out[x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
    return;
  }
  /* generic integral box replication (int_upsample) */
  {
    const uint8_t* in = c->plane + (size_t)(y / vx) * pitch;
    for (int x = 0; x < W; x++) out[x] = in[x / hx];
  }
}

static inline uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

static int finish_image(dec_t* d, uint8_t* rgb) {
  for (int i = 0; i < d->ncomp; i++) {
    comp_t* c = &d->comp[i];
    if (!d->qt_present[c->tq]) return UFO_E_DECODE;
    if ((d->hmax % c->h) || (d->vmax % c->v)) return UFO_E_UNSUPPORTED;
    int pitch = c->wblk * 8;
    for (int by = 0; by < c->hblk; by++)
      for (int bx = 0; bx < c->wblk; bx++)
        idct_islow(c->coef + ((size_t)by * c->wblk + bx) * 64, d->qt[c->tq], c->plane + (size_t)by * 8 * pitch + bx * 8,
                   pitch);
  }
  int W = d->width, H = d->height;
  if (d->ncomp == 1) {
    for (int y = 0; y < H; y++) {
      const uint8_t* in = d->comp[0].plane + (size_t)y * d->comp[0].wblk * 8;
      uint8_t* o = rgb + (size_t)y * W * 3;
      for (int x = 0; x < W; x++) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = in[x];
    }
    return UFO_OK;
  }
  /* colour space decision as libjpeg's default_decompress_parms for 3 components */
  int is_rgb = 0;
  if (d->saw_jfif) {
    is_rgb = 0;
  } else if (d->saw_adobe) {
    is_rgb = (d->adobe_transform == 0);
  } else {
    is_rgb = (d->comp[0].id == 'R' && d->comp[1].id == 'G' && d->comp[2].id == 'B');
  }
  uint8_t* r0 = (uint8_t*)malloc((size_t)W * 3);
  if (!r0) return UFO_E_ARG;
  uint8_t *r1 = r0 + W, *r2 = r1 + W;
  for (int y = 0; y < H; y++) {
    upsample_row(d, &d->comp[0], y, r0);
    upsample_row(d, &d->comp[1], y, r1);
    upsample_row(d, &d->comp[2], y, r2);
    uint8_t* o = rgb + (size_t)y * W * 3;
    if (is_rgb) {
      for (int x = 0; x < W; x++) {
        o[3 * x] = r0[x];
        o[3 * x + 1] = r1[x];
        o[3 * x + 2] = r2[x];
      }
    } else {
      for (int x = 0; x < W; x++) {
        int yy = r0[x], cb = r1[x] - 128, cr = r2[x] - 128;
        /* jdcolor.c build_ycc_rgb_table: SCALEBITS 16, ONE_HALF 32768; >> is arithmetic */
        int cr_r = (91881 * cr + 32768) >> 16;
        int cb_b = (116130 * cb + 32768) >> 16;
        int g = ((-22554) * cb + 32768 + (-46802) * cr) >> 16;
        o[3 * x] = clamp255(yy + cr_r);
        o[3 * x + 1] = clamp255(yy + g);
        o[3 * x + 2] = clamp255(yy + cb_b);
      }
    }
  }
  free(r0);
  return UFO_OK;
}

static int install_default_tables(dec_t* d) {
  int rc = 0;
  if (!d->dc[0].present) rc |= build_huff(&d->dc[0], kDcLumBits, kDcLumVal, 12);
  if (!d->dc[1].present) rc |= build_huff(&d->dc[1], kDcChrBits, kDcChrVal, 12);
  if (!d->ac[0].present) rc |= build_huff(&d->ac[0], kAcLumBits, kAcLumVal, 162);
  if (!d->ac[1].present) rc |= build_huff(&d->ac[1], kAcChrBits, kAcChrVal, 162);
  return rc;
}

/* Walk markers. If rgb == NULL only the header is parsed (probe). */
static int decode_stream_ex(const uint8_t* data, size_t len, dec_t* d, uint8_t* rgb, int want_w, int want_h, int coef_only);
static int decode_stream(const uint8_t* data, size_t len, dec_t* d, uint8_t* rgb, int want_w, int want_h) {
  return decode_stream_ex(data, len, d, rgb, want_w, want_h, 0);
}
static int decode_stream_ex(const uint8_t* data, size_t len, dec_t* d, uint8_t* rgb, int want_w, int want_h, int coef_only) {
  memset(d, 0, sizeof(*d));
  if (len < 4 || data[0] != 0xFF || data[1] != 0xD8) return UFO_E_DECODE;
  const uint8_t* p = data + 2;
  const uint8_t* end = data + len;
  int scans = 0;
  for (;;) {
    /* find next marker */
    while (p < end && *p != 0xFF) p++;
    while (p < end && *p == 0xFF) p++;
    /* missing EOI: libjpeg inserts a fake EOI and warns; tjDecompress2 turns any warning into
     * a failure return, which the reference `expect`s on (inferer.rs:35-36) */
    if (p >= end) return UFO_E_DECODE;
    int m = *p++;
    if (m == 0xD9) break;                       /* EOI */
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue; /* TEM / stray RST */
    if (p + 2 > end) return UFO_E_DECODE;
    int seglen = (p[0] << 8) | p[1];
    if (seglen < 2 || p + seglen > end) return UFO_E_DECODE;
    const uint8_t* s = p + 2;
    int sl = seglen - 2;
    int rc = UFO_OK;
    switch (m) {
      case 0xDB: rc = parse_dqt(d, s, sl); break;
      case 0xC4: rc = parse_dht(d, s, sl); break;
      case 0xC0:
      case 0xC1:
        if (d->saw_sof) return UFO_E_DECODE;
        rc = parse_sof(d, s, sl, 0);
        break;
      case 0xC2:
        if (d->saw_sof) return UFO_E_DECODE;
        rc = parse_sof(d, s, sl, 1);
        break;
      case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
        return UFO_E_UNSUPPORTED;
      case 0xDD:
        if (sl < 2) return UFO_E_DECODE;
        d->restart_interval = (s[0] << 8) | s[1];
        break;
      case 0xE0:
        if (sl >= 5 && s[0] == 'J' && s[1] == 'F' && s[2] == 'I' && s[3] == 'F' && s[4] == 0) d->saw_jfif = 1;
        break;
      case 0xEE:
        if (sl >= 12 && s[0] == 'A' && s[1] == 'd' && s[2] == 'o' && s[3] == 'b' && s[4] == 'e') {
          d->saw_adobe = 1;
          d->adobe_transform = s[11];
        }
        break;
      case 0xDA: {
        if (!d->saw_sof) return UFO_E_DECODE;
        if (!rgb) return UFO_OK; /* probe: header complete */
        if (!coef_only && (d->width != want_w || d->height != want_h)) return UFO_E_ARG;
        if (sl < 1) return UFO_E_DECODE;
        scan_t sc;
        sc.ns = s[0];
        if (sc.ns < 1 || sc.ns > d->ncomp || sl < 1 + 2 * sc.ns + 3) return UFO_E_DECODE;
        for (int i = 0; i < sc.ns; i++) {
          int cid = s[1 + 2 * i], ci = -1;
          for (int j = 0; j < d->ncomp; j++)
            if (d->comp[j].id == cid) ci = j;
          if (ci < 0) return UFO_E_DECODE;
          sc.ci[i] = ci;
          sc.td[i] = s[2 + 2 * i] >> 4;
          sc.ta[i] = s[2 + 2 * i] & 15;
          if (sc.td[i] > 3 || sc.ta[i] > 3) return UFO_E_DECODE;
        }
        sc.ss = s[1 + 2 * sc.ns];
        sc.se = s[2 + 2 * sc.ns];
        sc.ah = s[3 + 2 * sc.ns] >> 4;
        sc.al = s[3 + 2 * sc.ns] & 15;
        if (!d->progressive) {
          sc.ss = 0;
          sc.se = 63;
          sc.ah = sc.al = 0;
        } else {
          if (sc.ss > sc.se || sc.se > 63 || sc.al > 13) return UFO_E_DECODE;
          if (sc.ss == 0 && sc.se != 0) return UFO_E_DECODE;
          if (sc.ss > 0 && sc.ns != 1) return UFO_E_DECODE;
        }
        rc = install_default_tables(d);
        if (rc) return UFO_E_DECODE;
        const uint8_t* next = NULL;
        rc = decode_scan(d, &sc, p + seglen, end, &next);
        if (rc) return rc;
        scans++;
        p = next;
        continue;
      }
      default: break; /* APPn, COM, ... skipped */
    }
    if (rc) return rc;
    p += seglen;
  }
  if (!rgb) return d->saw_sof ? UFO_OK : UFO_E_DECODE;
  if (!scans) return UFO_E_DECODE;
  if (coef_only) return UFO_OK;
  return finish_image(d, rgb);
}

int ufo_jpeg_probe(const uint8_t* data, size_t len, ufo_jpeg_info* info) {
  dec_t d;
  int rc = decode_stream(data, len, &d, NULL, 0, 0);
  if (rc == UFO_OK && info) {
    memset(info, 0, sizeof(*info));
    info->width = d.width;
    info->height = d.height;
    info->ncomp = d.ncomp;
    info->progressive = d.progressive;
    info->restart_interval = d.restart_interval;
    for (int i = 0; i < d.ncomp; i++) {
      info->hsamp[i] = d.comp[i].h;
      info->vsamp[i] = d.comp[i].v;
    }
  }
  dec_free(&d);
  return rc;
}

int ufo_jpeg_decode_rgb(const uint8_t* data, size_t len, uint8_t* rgb, int width, int height) {
  if (!data || !rgb) return UFO_E_ARG;
  dec_t d;
  int rc = decode_stream(data, len, &d, rgb, width, height);
  dec_free(&d);
  return rc;
}

int ufo_jpeg_coefficients(const uint8_t* data, size_t len, int16_t* coef, size_t cap_i16, size_t* n_i16) {
  dec_t d;
  uint8_t dummy;
  int rc = decode_stream_ex(data, len, &d, coef ? &dummy : NULL, 0, 0, 1);
  if (rc == UFO_OK) {
    size_t total = 0;
    for (int i = 0; i < d.ncomp; i++) total += (size_t)d.comp[i].wblk * d.comp[i].hblk * 64;
    if (n_i16) *n_i16 = total;
    if (coef) {
      if (cap_i16 < total) {
        rc = UFO_E_ARG;
      } else {
        size_t o = 0;
        for (int i = 0; i < d.ncomp; i++) {
          size_t n = (size_t)d.comp[i].wblk * d.comp[i].hblk * 64;
          memcpy(coef + o, d.comp[i].coef, n * sizeof(int16_t));
          o += n;
        }
      }
    }
  }
  dec_free(&d);
  return rc;
}
