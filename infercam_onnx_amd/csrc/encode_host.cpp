// encode_host.cpp -- host half of row N1's JPEG encoder (turbojpeg::compress_image(&frame, 95, Subsamp::Sub2x2),
// infer_server/src/inferer.rs:39): everything that depends only on the quality setting -- the scaled Annex-K
// quantisation tables (jcparam.c jpeg_set_quality / jpeg_add_quant_table with force_baseline), their reciprocal form
// (jcdctmgr.c compute_reciprocal, divisors scaled by the AA&N factors for the fast DCT), the Annex-K Huffman code
// tables (jchuff.c jpeg_make_c_derived_tbl) and the marker segments in the order jcmarker.c writes them
// (SOI, JFIF APP0, DQT 0, DQT 1, SOF0, DHT DC0 AC0 DC1 AC1, SOS).  The per-pixel work is in encode_kernels.hip.
#include <cstring>

#include "kernels.hpp"

namespace ufd {
namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const uint8_t kLumaBase[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40, 57,
                               69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55, 64,
                               81, 104, 113, 92, 49, 64, 78, 87,  103, 121, 120, 101, 72, 92, 95,  98,  112, 100, 103, 99};
const uint8_t kChromaBase[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                 99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const int16_t kAanScales[64] = {16384, 22725, 21407, 19266, 16384, 12873, 8867,  4520,  22725, 31521, 29692, 26722, 22725,
                                17855, 12299, 6270,  21407, 29692, 27969, 25172, 21407, 16819, 11585, 5906,  19266, 26722,
                                25172, 22654, 19266, 15137, 10426, 5315,  16384, 22725, 21407, 19266, 16384, 12873, 8867,
                                4520,  12873, 17855, 16819, 15137, 12873, 10114, 6967,  3552,  8867,  12299, 11585, 10426,
                                8867,  6967,  4799,  2446,  4520,  6270,  5906,  5315,  4520,  3552,  2446,  1247};

const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct HuffSpec {
  uint8_t id;  // Tc << 4 | Th
  const uint8_t* bits;
  const uint8_t* vals;
  int nvals;
};
const HuffSpec kHuff[4] = {{0x00, kDcLumaBits, kDcVals, 12},
                           {0x10, kAcLumaBits, kAcLumaVals, 162},
                           {0x01, kDcChromaBits, kDcVals, 12},
                           {0x11, kAcChromaBits, kAcChromaVals, 162}};

void derive_codes(const HuffSpec& h, uint32_t* out /* indexed by symbol */, int nout) {
  for (int i = 0; i < nout; i++) out[i] = 0;
  uint32_t code = 0;
  int k = 0;
  for (int len = 1; len <= 16; len++) {
    for (int i = 0; i < h.bits[len - 1]; i++, k++, code++) out[h.vals[k]] = ((uint32_t)len << 16) | code;
    code <<= 1;
  }
}

}  // namespace

void enc_quant_tables(int quality, uint8_t luma[64], uint8_t chroma[64]) {
  if (quality <= 0) quality = 1;
  if (quality > 100) quality = 100;
  const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;  // jpeg_quality_scaling
  for (int t = 0; t < 2; t++)
    for (int i = 0; i < 64; i++) {
      long v = ((long)(t ? kChromaBase : kLumaBase)[i] * scale + 50L) / 100L;
      v = v <= 0 ? 1 : (v > 255 ? 255 : v);  // force_baseline
      (t ? chroma : luma)[i] = (uint8_t)v;
    }
}

void enc_make_quant(const uint8_t luma[64], const uint8_t chroma[64], bool ifast, EncQuant* out) {
  for (int t = 0; t < 2; t++)
    for (int i = 0; i < 64; i++) {
      const uint32_t q = (t ? chroma : luma)[i];
      // jcdctmgr.c start_pass_fdctmgr: the accurate DCT leaves a factor 8, the fast one the AA&N scale factors as well
      const uint32_t divisor = ifast ? (uint32_t)(((int32_t)q * kAanScales[i] + (1 << 10)) >> 11) : q << 3;
      if (divisor == 1) {
        out->recip[t][i] = 1, out->corr[t][i] = 0, out->shift[t][i] = 0;
        continue;
      }
      int b = 0;
      while ((divisor >> (b + 1)) != 0) b++;
      int r = 16 + b;
      uint32_t fq = (uint32_t)((1ull << r) / divisor);
      const uint32_t fr = (uint32_t)((1ull << r) % divisor);
      uint32_t c = divisor / 2;
      if (fr == 0) fq >>= 1, r--;           // power of two
      else if (fr <= divisor / 2) c++;      // fractional part < 0.5
      else fq++;                            // fractional part > 0.5
      out->recip[t][i] = (uint16_t)fq, out->corr[t][i] = (uint16_t)c, out->shift[t][i] = (uint8_t)r;
    }
}

void enc_make_code_tables(uint32_t out[2 * 272]) {
  derive_codes(kHuff[0], out, 16);
  derive_codes(kHuff[1], out + 16, 256);
  derive_codes(kHuff[2], out + 272, 16);
  derive_codes(kHuff[3], out + 272 + 16, 256);
}

size_t enc_make_header(const uint8_t luma[64], const uint8_t chroma[64], bool multipart, uint8_t* out, uint32_t* pre_len,
                       uint32_t* hdr_len, uint32_t* dim_off, uint32_t* post_len) {
  static const char kPre[] = "--frame\r\nContent-Type: image/jpeg\r\n\r\n";  // as_jpeg_stream_item, lib.rs:48-57
  static const char kPost[] = "\r\n\r\n";
  size_t n = 0;
  auto put = [&](const void* p, size_t k) {
    std::memcpy(out + n, p, k);
    n += k;
  };
  *pre_len = multipart ? (uint32_t)(sizeof(kPre) - 1) : 0;
  if (multipart) put(kPre, sizeof(kPre) - 1);
  const size_t h0 = n;
  static const uint8_t soi_app0[] = {0xFF, 0xD8, 0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
  put(soi_app0, sizeof(soi_app0));
  for (int t = 0; t < 2; t++) {
    const uint8_t hd[] = {0xFF, 0xDB, 0, 67, (uint8_t)t};
    put(hd, sizeof(hd));
    for (int i = 0; i < 64; i++) out[n++] = (t ? chroma : luma)[kZigzag[i]];
  }
  const uint8_t sof[] = {0xFF, 0xC0, 0, 17, 8, 0, 0, 0, 0, 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1};  // dimensions per frame
  *dim_off = (uint32_t)(n - h0 + 5);
  put(sof, sizeof(sof));
  for (const HuffSpec& h : kHuff) {
    const uint8_t hd[] = {0xFF, 0xC4, (uint8_t)((h.nvals + 19) >> 8), (uint8_t)(h.nvals + 19), h.id};
    put(hd, sizeof(hd));
    put(h.bits, 16);
    put(h.vals, (size_t)h.nvals);
  }
  static const uint8_t sos[] = {0xFF, 0xDA, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};
  put(sos, sizeof(sos));
  *hdr_len = (uint32_t)(n - h0);
  *post_len = multipart ? (uint32_t)(sizeof(kPost) - 1) : 0;
  if (multipart) put(kPost, sizeof(kPost) - 1);
  return n;
}

}  // namespace ufd
