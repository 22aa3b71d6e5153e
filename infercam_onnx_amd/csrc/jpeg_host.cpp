// jpeg_host.cpp -- marker parser + Huffman entropy decoder (ITU T.81), host side of row A1
// (turbojpeg::decompress_image, infer_server/src/inferer.rs:35).  Produces quantised DCT
// coefficients only; reconstruction happens on the GPU.  Baseline / extended-sequential and
// progressive Huffman, 8-bit, 1 or 3 components, restart intervals, Annex-K default tables
// for DHT-less MJPEG camera streams.
#include "jpeg_host.hpp"

#include <algorithm>
#include <cstring>

namespace ufd {
namespace {

const uint8_t kZigzag[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33,
                                  40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
                                  29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                                  47, 55, 62, 63,
                                  // guard entries: a corrupt run may index past 63 before the bounds check
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

// Annex K.3 tables: {counts[16]}, {symbols}
const uint8_t kStdDcLumCnt[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kStdDcChrCnt[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kStdDcSym[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kStdAcLumCnt[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kStdAcLumSym[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kStdAcChrCnt[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kStdAcChrSym[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

constexpr int kFast = 10;  // lookup width

struct HuffTable {
  bool present = false;
  // fast[peek(kFast)] = (len << 8) | symbol, 0 -> code longer than kFast bits
  uint16_t fast[1 << kFast];
  // AC shortcut (sequential scans): for prefixes where code + magnitude bits fit in kFast bits:
  // (value << 8) | (run << 4) | total_bits; 0 -> not available
  int32_t fast_ac[1 << kFast];
  int32_t maxcode[18];  // left-justified-to-16 upper bound per length
  int32_t delta[17];
  uint8_t sym[256];

  // The checks of build() without building: a table that passes is `present` but must not be used for decoding.
  bool check(const uint8_t* counts /*[16]*/, int nsym) {
    int total = 0, code = 0;
    for (int i = 0; i < 16; i++) total += counts[i];
    if (total > 256 || total != nsym) return false;
    for (int l = 1; l <= 16; l++) {
      const int cnt = counts[l - 1];
      if (code + cnt > (1 << l)) return false;
      code = (code + cnt) << 1;
    }
    present = true;
    return true;
  }
  bool build(const uint8_t* counts /*[16]*/, const uint8_t* symbols, int nsym) {
    int total = 0;
    for (int i = 0; i < 16; i++) total += counts[i];
    if (total > 256 || total != nsym) return false;
    std::memset(sym, 0, sizeof(sym));  // unused entries are compared too (device table-set cache)
    std::memset(maxcode, 0, sizeof(maxcode));
    std::memset(delta, 0, sizeof(delta));
    std::memcpy(sym, symbols, total);
    std::memset(fast, 0, sizeof(fast));
    std::memset(fast_ac, 0, sizeof(fast_ac));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
      delta[l] = k - code;
      int cnt = counts[l - 1];
      if (code + cnt > (1 << l)) return false;
      if (l <= kFast) {
        for (int i = 0; i < cnt; i++) {
          int c = (code + i) << (kFast - l);
          uint16_t e = static_cast<uint16_t>((l << 8) | symbols[k + i]);
          for (int j = 0; j < (1 << (kFast - l)); j++) fast[c + j] = e;
        }
      }
      code += cnt;
      k += cnt;
      maxcode[l] = cnt ? (code << (16 - l)) : -1;  // exclusive upper bound, left-justified
      if (!cnt) maxcode[l] = code << (16 - l);
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    // AC fast path
    for (int i = 0; i < (1 << kFast); i++) {
      int e = fast[i];
      if (!e) continue;
      int len = e >> 8, rs = e & 0xFF, run = rs >> 4, mag = rs & 15;
      if (mag && len + mag <= kFast) {
        int bits = (i << len) & ((1 << kFast) - 1);
        int v = bits >> (kFast - mag);
        if (v < (1 << (mag - 1))) v += static_cast<int>(0xFFFFFFFFu << mag) + 1;
        if (v >= -128 && v <= 127) fast_ac[i] = (v * 256) + (run * 16) + (len + mag);
      }
    }
    present = true;
    return true;
  }
};

inline uint64_t load_be64(const uint8_t* p) {
  uint64_t v;
  std::memcpy(&v, p, 8);
  return __builtin_bswap64(v);
}

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t acc = 0;
  int n = 0;        // valid bits at the top of acc
  int marker = 0;   // marker byte that terminated the segment
  int pad_bits = 0; // zero bits appended after the end of data

  BitReader(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}

  inline void fill() {
    // fast path: 8 bytes with no 0xFF
    if (!marker && end - p >= 8) {
      uint64_t v = load_be64(p);
      uint64_t nv = ~v;
      if (!((nv - 0x0101010101010101ULL) & ~nv & 0x8080808080808080ULL)) {
        int bytes = (63 - n) >> 3;  // 7 or fewer
        if (bytes > 0) {
          acc |= (v >> (64 - 8 * bytes)) << (64 - n - 8 * bytes);
          n += 8 * bytes;
          p += bytes;
        }
        return;
      }
    }
    while (n <= 56) {
      unsigned c = 0;
      if (!marker && p < end) {
        c = *p++;
        if (c == 0xFF) {
          while (p < end && *p == 0xFF) p++;
          if (p >= end) {
            marker = 0xD9;
            c = 0;
            pad_bits += 8;
          } else if (*p == 0) {
            p++;
          } else {
            marker = *p++;
            c = 0;
            pad_bits += 8;
          }
        }
      } else {
        if (!marker) marker = 0xD9;
        pad_bits += 8;
      }
      acc |= static_cast<uint64_t>(c) << (56 - n);
      n += 8;
    }
  }
  inline uint32_t peek(int k) const { return static_cast<uint32_t>(acc >> (64 - k)); }
  inline void skip(int k) {
    acc <<= k;
    n -= k;
  }
  inline int get(int k) {  // k in 1..16
    if (n < k) fill();
    int v = static_cast<int>(peek(k));
    skip(k);
    return v;
  }
  inline bool overrun() const { return pad_bits > 0 && n < pad_bits; }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v + static_cast<int>(0xFFFFFFFFu << s) + 1 : v; }

inline int decode_symbol(BitReader& br, const HuffTable& h) {
  if (br.n < 16) br.fill();
  int e = h.fast[br.peek(kFast)];
  if (e) {
    br.skip(e >> 8);
    return e & 0xFF;
  }
  int code = static_cast<int>(br.peek(16));
  int l = kFast + 1;
  while (code >= h.maxcode[l]) l++;
  if (l > 16) return -1;
  br.skip(l);
  return h.sym[((code >> (16 - l)) + h.delta[l]) & 0xFF];
}

struct Scan {
  int ns;
  int comp[kMaxComps], td[kMaxComps], ta[kMaxComps];
  int ss, se, ah, al;
};

struct Decoder {
  JpegFrameDesc* d;
  int16_t* coef;
  GpuScanPlan* plan = nullptr;  // non-null: plan a device decode instead of decoding
  bool build_luts = true;       // plan mode: false = check + hash the DHT segments, build no table
  uint32_t key_len = 0;         // plan mode: DHT payload bytes collected into plan->key_bytes so far
  bool key_overflow = false;
  HuffTable dc[4], ac[4];
  uint16_t qtab[4][64];
  bool qt_present[4] = {false, false, false, false};
  int comp_id[kMaxComps], comp_tq[kMaxComps];
  bool saw_sof = false, saw_jfif = false, saw_adobe = false;
  int adobe_transform = 0;

  int parse_dqt(const uint8_t* s, int len) {
    while (len > 0) {
      int pq = s[0] >> 4, tq = s[0] & 15;
      if (pq > 1 || tq > 3) return kJpegCorrupt;
      int need = 1 + (pq ? 128 : 64);
      if (len < need) return kJpegCorrupt;
      for (int i = 0; i < 64; i++) qtab[tq][kZigzag[i]] = pq ? static_cast<uint16_t>((s[1 + 2 * i] << 8) | s[2 + 2 * i]) : s[1 + i];
      qt_present[tq] = true;
      s += need;
      len -= need;
    }
    return kJpegOk;
  }
  int parse_dht(const uint8_t* s, int len) {
    while (len > 0) {
      if (len < 17) return kJpegCorrupt;
      int tc = s[0] >> 4, th = s[0] & 15;
      if (tc > 1 || th > 3) return kJpegCorrupt;
      int total = 0;
      for (int i = 0; i < 16; i++) total += s[1 + i];
      if (total > 256 || len < 17 + total) return kJpegCorrupt;
      HuffTable& t = tc ? ac[th] : dc[th];
      if (!(plan && !build_luts ? t.check(s + 1, total) : t.build(s + 1, s + 17, total))) return kJpegCorrupt;
      if (plan) {
        const uint32_t nb = 17 + (uint32_t)total;
        if (key_len + nb + 8 > GpuScanPlan::kMaxKeyBytes) key_overflow = true;
        else std::memcpy(plan->key_bytes + key_len, s, nb), key_len += nb;
      }
      s += 17 + total;
      len -= 17 + total;
    }
    return kJpegOk;
  }
  int parse_sof(const uint8_t* s, int len, bool progressive) {
    if (saw_sof || len < 6) return kJpegCorrupt;
    if (s[0] != 8) return kJpegUnsupported;
    d->height = (s[1] << 8) | s[2];
    d->width = (s[3] << 8) | s[4];
    d->ncomp = s[5];
    d->progressive = progressive;
    if (d->width <= 0 || d->height <= 0) return kJpegCorrupt;
    if (d->ncomp != 1 && d->ncomp != 3) return kJpegUnsupported;
    if (len < 6 + 3 * d->ncomp) return kJpegCorrupt;
    d->hmax = d->vmax = 1;
    for (int i = 0; i < d->ncomp; i++) {
      comp_id[i] = s[6 + 3 * i];
      d->h[i] = s[7 + 3 * i] >> 4;
      d->v[i] = s[7 + 3 * i] & 15;
      comp_tq[i] = s[8 + 3 * i];
      if (d->h[i] < 1 || d->h[i] > 4 || d->v[i] < 1 || d->v[i] > 4 || comp_tq[i] > 3) return kJpegCorrupt;
      if (d->h[i] > d->hmax) d->hmax = d->h[i];
      if (d->v[i] > d->vmax) d->vmax = d->v[i];
    }
    if (d->ncomp == 1) d->h[0] = d->v[0] = d->hmax = d->vmax = 1;  // single component: always 1x1 MCU
    d->mcux = (d->width + 8 * d->hmax - 1) / (8 * d->hmax);
    d->mcuy = (d->height + 8 * d->vmax - 1) / (8 * d->vmax);
    uint32_t coff = 0, poff = 0, blocks = 0;
    for (int i = 0; i < d->ncomp; i++) {
      // integral expansion factors only, as jdsample.c (JERR_FRACT_SAMPLE_NOTIMPL otherwise): 1..4 each way
      if (d->hmax % d->h[i] || d->vmax % d->v[i]) return kJpegUnsupported;
      d->wblk[i] = d->mcux * d->h[i];
      d->hblk[i] = d->mcuy * d->v[i];
      d->dw[i] = (d->width * d->h[i] + d->hmax - 1) / d->hmax;
      d->dh[i] = (d->height * d->v[i] + d->vmax - 1) / d->vmax;
      d->coef_off[i] = coff;
      d->plane_off[i] = poff;
      uint32_t nb = static_cast<uint32_t>(d->wblk[i]) * d->hblk[i];
      coff += nb * 64;
      poff += nb * 64;
      blocks += nb;
    }
    d->coef_total = coff;
    d->plane_total = poff;
    d->total_blocks = blocks;
    saw_sof = true;
    return kJpegOk;
  }
  void install_defaults() {
    if (plan && !build_luts) {  // (which defaults get installed follows from the DHT bytes already in the key)
      dc[0].present = dc[1].present = ac[0].present = ac[1].present = true;
      return;
    }
    if (!dc[0].present) dc[0].build(kStdDcLumCnt, kStdDcSym, 12);
    if (!dc[1].present) dc[1].build(kStdDcChrCnt, kStdDcSym, 12);
    if (!ac[0].present) ac[0].build(kStdAcLumCnt, kStdAcLumSym, 162);
    if (!ac[1].present) ac[1].build(kStdAcChrCnt, kStdAcChrSym, 162);
  }
  void finish_desc() {
    if (d->ncomp == 1) {
      d->color = kColorGray;
    } else if (saw_jfif) {
      d->color = kColorYCbCr;
    } else if (saw_adobe) {
      d->color = adobe_transform == 0 ? kColorRGB : kColorYCbCr;
    } else {
      d->color = (comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B') ? kColorRGB : kColorYCbCr;
    }
  }

  // ---- sequential (baseline) block: DC diff + run/size AC ----
  inline int block_sequential(BitReader& br, const HuffTable& hd, const HuffTable& ha, int& pred, int16_t* blk) {
    std::memset(blk, 0, 128);
    int s = decode_symbol(br, hd);
    if (s < 0 || s > 15) return kJpegCorrupt;
    if (s) pred += extend(br.get(s), s);
    blk[0] = static_cast<int16_t>(pred);
    int k = 1;
    do {
      if (br.n < 16) br.fill();
      int f = ha.fast_ac[br.peek(kFast)];
      if (f) {
        k += (f >> 4) & 15;
        br.skip(f & 15);
        blk[kZigzag[k++]] = static_cast<int16_t>(f >> 8);  // k <= 63+15 covered by guard entries
        if (k > 64) return kJpegCorrupt;
        continue;
      }
      int rs = decode_symbol(br, ha);
      if (rs < 0) return kJpegCorrupt;
      int r = rs >> 4;
      s = rs & 15;
      if (s == 0) {
        if (r != 15) break;
        k += 16;
      } else {
        k += r;
        if (k > 63) return kJpegCorrupt;
        blk[kZigzag[k++]] = static_cast<int16_t>(extend(br.get(s), s));
      }
    } while (k < 64);
    return kJpegOk;
  }

  static inline void refine_nonzero(BitReader& br, int16_t* co, int p1, int m1) {
    if (br.get(1)) {
      if ((*co & p1) == 0) *co = static_cast<int16_t>(*co >= 0 ? *co + p1 : *co + m1);
    }
  }

  int block_progressive(BitReader& br, const Scan& sc, int i, int& pred, int& eobrun, int16_t* blk) {
    if (sc.ss == 0) {
      if (sc.ah == 0) {
        int s = decode_symbol(br, dc[sc.td[i]]);
        if (s < 0 || s > 15) return kJpegCorrupt;
        if (s) pred += extend(br.get(s), s);
        blk[0] = static_cast<int16_t>(pred * (1 << sc.al));
      } else if (br.get(1)) {
        blk[0] |= static_cast<int16_t>(1 << sc.al);
      }
      return kJpegOk;
    }
    const HuffTable& ha = ac[sc.ta[i]];
    if (sc.ah == 0) {
      if (eobrun > 0) {
        eobrun--;
        return kJpegOk;
      }
      for (int k = sc.ss; k <= sc.se; k++) {
        int rs = decode_symbol(br, ha);
        if (rs < 0) return kJpegCorrupt;
        int r = rs >> 4, s = rs & 15;
        if (s) {
          k += r;
          if (k > 63) return kJpegCorrupt;
          blk[kZigzag[k]] = static_cast<int16_t>(extend(br.get(s), s) * (1 << sc.al));
        } else if (r == 15) {
          k += 15;
        } else {
          eobrun = 1 << r;
          if (r) eobrun += br.get(r);
          eobrun--;
          break;
        }
      }
      return kJpegOk;
    }
    const int p1 = 1 << sc.al, m1 = -(1 << sc.al);
    int k = sc.ss;
    if (eobrun == 0) {
      for (; k <= sc.se; k++) {
        int rs = decode_symbol(br, ha);
        if (rs < 0) return kJpegCorrupt;
        int r = rs >> 4, s = rs & 15;
        if (s) {
          s = br.get(1) ? p1 : m1;
        } else if (r != 15) {
          eobrun = 1 << r;
          if (r) eobrun += br.get(r);
          break;
        }
        do {
          int16_t* co = blk + kZigzag[k];
          if (*co != 0) {
            refine_nonzero(br, co, p1, m1);
          } else if (--r < 0) {
            break;
          }
          k++;
        } while (k <= sc.se);
        if (s) {
          if (k > 63) return kJpegCorrupt;
          blk[kZigzag[k]] = static_cast<int16_t>(s);
        }
      }
    }
    if (eobrun > 0) {
      for (; k <= sc.se; k++) {
        int16_t* co = blk + kZigzag[k];
        if (*co != 0) refine_nonzero(br, co, p1, m1);
      }
      eobrun--;
    }
    return kJpegOk;
  }

  int decode_scan(const Scan& sc, const uint8_t* p, const uint8_t* end) {
    BitReader br(p, end);
    int pred[kMaxComps] = {0, 0, 0};
    int eobrun = 0;
    const int ri = d->restart_interval;
    int rst_left = ri, next_rst = 0;
    const bool inter = sc.ns > 1;
    int nx = d->mcux, ny = d->mcuy;
    if (!inter) {
      nx = (d->dw[sc.comp[0]] + 7) / 8;
      ny = (d->dh[sc.comp[0]] + 7) / 8;
    }
    if (inter) {  // jdinput.c per_scan_setup: more than D_MAX_BLOCKS_IN_MCU = 10 blocks is JERR_BAD_MCU_SIZE
      int blocks = 0;
      for (int i = 0; i < sc.ns; i++) blocks += d->h[sc.comp[i]] * d->v[sc.comp[i]];
      if (blocks > 10) return kJpegCorrupt;
    }
    for (int i = 0; i < sc.ns; i++) {
      bool need_dc = !d->progressive || (sc.ss == 0 && sc.ah == 0);
      bool need_ac = !d->progressive || sc.ss > 0;
      if (need_dc && !dc[sc.td[i]].present) return kJpegCorrupt;
      if (need_ac && !ac[sc.ta[i]].present) return kJpegCorrupt;
    }
    for (int my = 0; my < ny; my++) {
      for (int mx = 0; mx < nx; mx++) {
        if (ri && rst_left == 0) {
          if (!br.marker) {  // padding bits / fill bytes before the marker
            const uint8_t* q = br.p;
            while (q + 1 < end && !(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF)) q++;
            if (q + 1 >= end) return kJpegCorrupt;
            br.marker = q[1];
            br.p = q + 2;
          }
          if (br.marker != 0xD0 + next_rst) return kJpegCorrupt;
          next_rst = (next_rst + 1) & 7;
          br = BitReader(br.p, end);
          pred[0] = pred[1] = pred[2] = 0;
          eobrun = 0;
          rst_left = ri;
        }
        for (int i = 0; i < sc.ns; i++) {
          const int c = sc.comp[i];
          const int bw = inter ? d->h[c] : 1, bh = inter ? d->v[c] : 1;
          for (int by = 0; by < bh; by++) {
            for (int bx = 0; bx < bw; bx++) {
              const int row = inter ? my * d->v[c] + by : my, col = inter ? mx * d->h[c] + bx : mx;
              int16_t* blk = coef + d->coef_off[c] + (static_cast<size_t>(row) * d->wblk[c] + col) * 64;
              int rc = d->progressive ? block_progressive(br, sc, i, pred[i], eobrun, blk)
                                      : block_sequential(br, dc[sc.td[i]], ac[sc.ta[i]], pred[i], blk);
              if (rc) return rc;
            }
          }
        }
        if (br.overrun()) return kJpegCorrupt;
        if (ri) rst_left--;
      }
    }
    return kJpegOk;
  }

  static void export_lut(const HuffTable& t, HuffLut* o) {
    static_assert(sizeof(o->fast) == sizeof(t.fast), "lookup width mismatch");
    std::memcpy(o->fast, t.fast, sizeof(o->fast));
    std::memcpy(o->maxcode, t.maxcode, sizeof(o->maxcode));
    std::memcpy(o->delta, t.delta, sizeof(o->delta));
    std::memcpy(o->sym, t.sym, sizeof(o->sym));
  }

  // Marker scan of one interleaved baseline scan with restart intervals: no bit is decoded.
  int plan_scan(const Scan& sc, const uint8_t* base, const uint8_t* p, const uint8_t* end) {
    // without restart markers the whole scan is one interval (self-synchronising device decoder)
    const long total_mcus = (long)d->mcux * d->mcuy;
    const long ri = d->restart_interval > 0 ? d->restart_interval : total_mcus;
    if (d->progressive || sc.ns != d->ncomp || total_mcus <= 0) return kJpegNotEligible;
    if (d->ncomp == 1 && (d->h[0] != 1 || d->v[0] != 1)) return kJpegNotEligible;  // non-interleaved MCU = one block
    std::memset(&plan->scan, 0, sizeof(plan->scan));
    // table slots: at most 2 distinct DC and 2 distinct AC tables
    int dc_ids[2] = {-1, -1}, ac_ids[2] = {-1, -1};
    auto slot = [](int (&ids)[2], int id) {
      for (int i = 0; i < 2; i++) {
        if (ids[i] == id) return i;
        if (ids[i] < 0) {
          ids[i] = id;
          return i;
        }
      }
      return -1;
    };
    int nb = 0;
    for (int i = 0; i < sc.ns; i++) {
      const int c = sc.comp[i];
      if (!dc[sc.td[i]].present || !ac[sc.ta[i]].present) return kJpegCorrupt;
      const int ds = slot(dc_ids, sc.td[i]), as = slot(ac_ids, sc.ta[i]);
      if (ds < 0 || as < 0) return kJpegNotEligible;
      for (int by = 0; by < d->v[c]; by++)
        for (int bx = 0; bx < d->h[c]; bx++) {
          if (nb >= 10) return kJpegCorrupt;  // (as decode_scan: libjpeg refuses the scan)
          plan->scan.blk_comp[nb] = (uint8_t)c, plan->scan.blk_bx[nb] = (uint8_t)bx, plan->scan.blk_by[nb] = (uint8_t)by;
          plan->scan.blk_dc[nb] = (uint8_t)ds, plan->scan.blk_ac[nb] = (uint8_t)as;
          nb++;
        }
    }
    plan->scan.blocks_per_mcu = nb;
    if (build_luts) {
      std::memset(plan->luts, 0, sizeof(plan->luts));
      for (int i = 0; i < 2; i++) {
        if (dc_ids[i] >= 0) export_lut(dc[dc_ids[i]], &plan->luts[i]);
        if (ac_ids[i] >= 0) export_lut(ac[ac_ids[i]], &plan->luts[2 + i]);
      }
    }
    // key of the table set: DHT payloads so far + the scan's selectors
    plan->key_hash = 0, plan->key_len = 0;
    if (!key_overflow) {
      uint8_t* kb = plan->key_bytes;
      kb[key_len++] = 0xFF;
      kb[key_len++] = (uint8_t)sc.ns;
      for (int i = 0; i < sc.ns; i++) kb[key_len++] = (uint8_t)((sc.td[i] << 4) | sc.ta[i]);
      uint64_t h = 1469598103934665603ull;
      for (uint32_t i = 0; i < key_len; i++) h = (h ^ kb[i]) * 1099511628211ull;
      plan->key_hash = h ? h : 1;
      plan->key_len = key_len;
    }
    const long n_iv = (total_mcus + ri - 1) / ri;
    if (n_iv > GpuScanPlan::kMaxIntervals) return kJpegNotEligible;
    // walk the entropy-coded segment: 0xFF00 = stuffing, 0xFFFF.. = fill, 0xFFD0-D7 = RSTn
    const uint8_t* q = p;
    uint32_t begin = (uint32_t)(p - base);
    long k = 0;
    int next_rst = 0;
    for (;;) {
      const uint8_t* f = static_cast<const uint8_t*>(std::memchr(q, 0xFF, end - q));
      if (!f || f + 1 >= end) return kJpegCorrupt;  // ran off the data: no EOI
      const int m = f[1];
      if (m == 0x00) {
        q = f + 2;
        continue;
      }
      if (m == 0xFF) {
        q = f + 1;
        continue;
      }
      if (k >= n_iv) return kJpegCorrupt;
      HuffInterval& iv = plan->iv[k];
      iv.frame = 0;
      iv.begin = begin;
      iv.end = (uint32_t)(f - base);
      // (optional 0xFF fill bytes in front of the marker stay inside the range: the device bit
      // reader stops at the first 0xFF that is not followed by 0x00)
      iv.mcu0 = (uint32_t)(k * ri);
      iv.nmcu = (uint32_t)std::min<long>(ri, total_mcus - k * ri);
      k++;
      if (m >= 0xD0 && m <= 0xD7) {
        if (m != 0xD0 + next_rst || d->restart_interval <= 0) return kJpegCorrupt;
        next_rst = (next_rst + 1) & 7;
        begin = (uint32_t)(f + 2 - base);
        q = f + 2;
        continue;
      }
      // any other marker ends the scan
      if (k != n_iv) return kJpegCorrupt;
      plan->n_intervals = (uint32_t)n_iv;
      scan_end = f;
      return kJpegOk;
    }
  }
  const uint8_t* scan_end = nullptr;

  // header_only: stop at the first SOS.
  int run(const uint8_t* data, size_t len, size_t coef_cap, bool header_only) {
    std::memset(d, 0, sizeof(*d));
    if (len < 4 || data[0] != 0xFF || data[1] != 0xD8) return kJpegCorrupt;
    const uint8_t* p = data + 2;
    const uint8_t* end = data + len;
    int scans = 0;
    bool zeroed = false;
    for (;;) {
      while (p < end && *p != 0xFF) p++;
      while (p < end && *p == 0xFF) p++;
      if (p >= end) return kJpegCorrupt;  // no EOI: libjpeg warns, turbojpeg reports failure
      int m = *p++;
      if (m == 0xD9) break;
      if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
      if (end - p < 2) return kJpegCorrupt;
      int seglen = (p[0] << 8) | p[1];
      if (seglen < 2 || end - p < seglen) return kJpegCorrupt;
      const uint8_t* s = p + 2;
      int sl = seglen - 2, rc = kJpegOk;
      switch (m) {
        case 0xDB: rc = parse_dqt(s, sl); break;
        case 0xC4: rc = parse_dht(s, sl); break;
        case 0xC0: case 0xC1: rc = parse_sof(s, sl, false); break;
        case 0xC2: rc = parse_sof(s, sl, true); break;
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
          return kJpegUnsupported;
        case 0xDD:
          if (sl < 2) return kJpegCorrupt;
          d->restart_interval = (s[0] << 8) | s[1];
          break;
        case 0xE0:
          if (sl >= 5 && !std::memcmp(s, "JFIF\0", 5)) saw_jfif = true;
          break;
        case 0xEE:
          if (sl >= 12 && !std::memcmp(s, "Adobe", 5)) {
            saw_adobe = true;
            adobe_transform = s[11];
          }
          break;
        case 0xDA: {
          if (!saw_sof) return kJpegCorrupt;
          if (header_only) {
            finish_desc();
            return kJpegOk;
          }
          if (!plan && d->coef_total > coef_cap) return kJpegUnsupported;
          Scan sc;
          if (sl < 1) return kJpegCorrupt;
          sc.ns = s[0];
          if (sc.ns < 1 || sc.ns > d->ncomp || sl < 4 + 2 * sc.ns) return kJpegCorrupt;
          for (int i = 0; i < sc.ns; i++) {
            int ci = -1;
            for (int j = 0; j < d->ncomp; j++)
              if (comp_id[j] == s[1 + 2 * i]) ci = j;
            if (ci < 0) return kJpegCorrupt;
            sc.comp[i] = ci;
            sc.td[i] = s[2 + 2 * i] >> 4;
            sc.ta[i] = s[2 + 2 * i] & 15;
            if (sc.td[i] > 3 || sc.ta[i] > 3) return kJpegCorrupt;
          }
          sc.ss = s[1 + 2 * sc.ns];
          sc.se = s[2 + 2 * sc.ns];
          sc.ah = s[3 + 2 * sc.ns] >> 4;
          sc.al = s[3 + 2 * sc.ns] & 15;
          if (!d->progressive) {
            sc.ss = 0, sc.se = 63, sc.ah = sc.al = 0;
          } else {
            if (sc.ss > sc.se || sc.se > 63 || sc.al > 13) return kJpegCorrupt;
            if ((sc.ss == 0 && sc.se != 0) || (sc.ss > 0 && sc.ns != 1)) return kJpegCorrupt;
          }
          install_defaults();
          if (plan) {
            if (scans) return kJpegNotEligible;
            rc = plan_scan(sc, data, p + seglen, end);
            if (rc) return rc;
            scans++;
            p = scan_end;
            continue;
          }
          // a sequential interleaved scan over all components writes every block in full;
          // anything else (progressive, per-component scans) accumulates into a zeroed slab
          const bool writes_all = !d->progressive && sc.ns == d->ncomp;
          if (!writes_all && !zeroed) {
            std::memset(coef, 0, static_cast<size_t>(d->coef_total) * sizeof(int16_t));
            zeroed = true;
          }
          rc = decode_scan(sc, p + seglen, end);
          if (rc) return rc;
          scans++;
          // next marker segment: first 0xFF followed by neither stuffing, fill nor RSTn
          const uint8_t* q = p + seglen;
          while (q + 1 < end && !(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
          p = (q + 1 < end) ? q : end;
          continue;
        }
        default: break;
      }
      if (rc) return rc;
      p += seglen;
    }
    if (!saw_sof || (!header_only && !scans)) return kJpegCorrupt;
    finish_desc();
    for (int i = 0; i < d->ncomp; i++) {
      if (!qt_present[comp_tq[i]]) return kJpegCorrupt;
      std::memcpy(d->qt[i], qtab[comp_tq[i]], sizeof(d->qt[i]));
    }
    return kJpegOk;
  }
};

}  // namespace

int jpeg_parse_header(const uint8_t* data, size_t len, JpegFrameDesc* d) {
  Decoder dec;
  dec.d = d;
  dec.coef = nullptr;
  return dec.run(data, len, 0, true);
}

int jpeg_plan_gpu_scan(const uint8_t* data, size_t len, JpegFrameDesc* d, GpuScanPlan* plan, bool build_luts) {
  Decoder dec;
  dec.d = d;
  dec.coef = nullptr;
  dec.plan = plan;
  dec.build_luts = build_luts;
  plan->n_intervals = 0;
  plan->key_hash = 0, plan->key_len = 0;
  return dec.run(data, len, 0, false);
}

int jpeg_decode_coefficients(const uint8_t* data, size_t len, JpegFrameDesc* d, int16_t* coef, size_t coef_cap) {
  Decoder dec;
  dec.d = d;
  dec.coef = coef;
  return dec.run(data, len, coef_cap, false);
}

}  // namespace ufd
