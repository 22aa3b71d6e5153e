// encode_kernels.hip -- SURVEY 8(f) row N1: what Inferer::run does with the detections of a frame
// (infer_server/src/inferer.rs:38-46), on the frame that is already decoded in HBM:
//   draw_bboxes_on_image  (inferer.rs:58-92: hollow rectangles at bbox * (width, height) cast to i32 / u32, colour (0,255,0);
//                          the confidence text is not drawn -- DESIGN.md section 7)
//   turbojpeg::compress_image(&frame, 95, Subsamp::Sub2x2)  (inferer.rs:39: libjpeg-turbo through tjCompress2, flags 0 --
//                          quality tables of jpeg_set_quality, YCbCr 4:2:0 with h2v2 box downsampling, the FAST integer DCT
//                          below quality 96 (the accurate one from 96 on), reciprocal quantisation, Annex-K Huffman tables,
//                          JFIF header, no restart markers)
//   as_jpeg_stream_item   (lib.rs:48-57: multipart framing, optional)
// Byte work, HBM/latency bound, no matrix cores.  Stages (all per batch, one launch each):
//   k_draw_rects   one workgroup per detection: the four clipped edges into the RGB frame
//   k_enc_ycc      one thread per 8 x 2 pixels: RGB -> Y / box-filtered Cb, Cr planes, MCU padded (edge samples
//                  replicated as jcsample / jcprepct do)
//   k_enc_fdct     one thread per 8x8 block of the planes: forward DCT -> quantised coefficients in zigzag order,
//                  [mcu][Y00 Y01 Y10 Y11 Cb Cr][64]; dummy blocks beyond a component's block grid as jccoefct.c makes
//                  them (DC of the block before, AC 0)
//   k_enc_bits     one thread per block: length in bits of its Huffman code
//   k_enc_scan     one workgroup per frame: exclusive scan -> bit offset of every block; zeroes the bit buffer
//   k_enc_write    one thread per block: code words at its bit offset (whole words stored, shared edge words OR-ed)
//   k_enc_ffcount  0xFF bytes per 4 KiB of the bit stream; k_enc_layout: stuffed length and offset of every frame in
//                  the packed output; k_enc_stuff: header + byte-stuffed entropy data + EOI (+ multipart framing)
// Bit-exact with the CPU oracle (oracle/encode_oracle.c), which is pinned to libjpeg-turbo's own streams.
#include "kernels.hpp"

namespace ufd {
namespace {

// Rust `f32 as i32` / `f32 as u32` (inferer.rs:73-76): truncate toward zero, saturate, NaN -> 0
__device__ __forceinline__ long long f32_as_i32(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return 2147483647LL;
  if (v <= -2147483648.0f) return -2147483648LL;
  return (long long)(int)v;
}
__device__ __forceinline__ unsigned long long f32_as_u32(float v) {
  if (v != v || v <= 0.0f) return 0;
  if (v >= 4294967296.0f) return 4294967295ULL;
  return (unsigned long long)(unsigned)v;
}

__global__ __launch_bounds__(256) void k_draw_rects(const JpegFrameDesc* __restrict__ descs, const Det* __restrict__ dets,
                                                    uint32_t det_stride, const uint32_t* __restrict__ ndet,
                                                    uint8_t* __restrict__ rgb, size_t rgb_stride, float label_w, float label_h) {
  const int frame = blockIdx.y;
  const int w = descs[frame].width, h = descs[frame].height;
  if (w <= 0 || h <= 0) return;  // failed / skipped frame (zero descriptor)
  const uint32_t n = min(ndet[frame], det_stride);
  uint8_t* img = rgb + (size_t)frame * rgb_stride;
  auto put = [&](long long x, long long y) {
    uint8_t* p = img + ((size_t)y * w + (size_t)x) * 3;
    p[0] = 0, p[1] = 255, p[2] = 0;
  };
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const Det d = dets[(size_t)frame * det_stride + i];
    // inferer.rs:69-76 in f32, then Rect::at(x as i32, y as i32).of_size(w as u32, h as u32)
    const float x_tl = __fmul_rn(d.x_tl, label_w), y_tl = __fmul_rn(d.y_tl, label_h);
    const float x_br = __fmul_rn(d.x_br, label_w), y_br = __fmul_rn(d.y_br, label_h);
    const unsigned long long rw = f32_as_u32(__fsub_rn(x_br, x_tl)), rh = f32_as_u32(__fsub_rn(y_br, y_tl));
    if (rw == 0 || rh == 0) continue;  // Rect::of_size asserts: the reference task would panic; no rectangle here
    const long long l = f32_as_i32(x_tl), t = f32_as_i32(y_tl);
    const long long r = l + (long long)rw - 1, b = t + (long long)rh - 1;
    const long long x0 = max(l, 0LL), x1 = min(r, (long long)w - 1), y0 = max(t, 0LL), y1 = min(b, (long long)h - 1);
    const bool top_in = t >= 0 && t < h, bot_in = b >= 0 && b < h, left_in = l >= 0 && l < w, right_in = r >= 0 && r < w;
    for (long long x = x0 + threadIdx.x; x <= x1; x += 256) {
      if (top_in) put(x, t);
      if (bot_in) put(x, b);
    }
    for (long long y = y0 + threadIdx.x; y <= y1; y += 256) {
      if (left_in) put(l, y);
      if (right_in) put(r, y);
    }
  }
}

// ---- forward DCTs on 16-bit elements (the arithmetic of libjpeg-turbo's SIMD build) ----
__device__ __forceinline__ int w16(int x) { return (int)(short)x; }
// jfdctfst-sse2.asm: MULTIPLY(v, c) = pmulhw(v << 2, c << 6)
__device__ __forceinline__ int ifast_mul(int v, int c) { return w16((w16(v << 2) * (c << 6)) >> 16); }

template <int S>
__device__ __forceinline__ void fdct_ifast_1d(int* d) {
  const int tmp0 = w16(d[0] + d[7 * S]), tmp7 = w16(d[0] - d[7 * S]), tmp1 = w16(d[S] + d[6 * S]), tmp6 = w16(d[S] - d[6 * S]);
  const int tmp2 = w16(d[2 * S] + d[5 * S]), tmp5 = w16(d[2 * S] - d[5 * S]), tmp3 = w16(d[3 * S] + d[4 * S]),
            tmp4 = w16(d[3 * S] - d[4 * S]);
  int tmp10 = w16(tmp0 + tmp3), tmp13 = w16(tmp0 - tmp3), tmp11 = w16(tmp1 + tmp2), tmp12 = w16(tmp1 - tmp2);
  d[0] = w16(tmp10 + tmp11);
  d[4 * S] = w16(tmp10 - tmp11);
  const int z1 = ifast_mul(w16(tmp12 + tmp13), 181);
  d[2 * S] = w16(tmp13 + z1);
  d[6 * S] = w16(tmp13 - z1);
  tmp10 = w16(tmp4 + tmp5), tmp11 = w16(tmp5 + tmp6), tmp12 = w16(tmp6 + tmp7);
  const int z5 = ifast_mul(w16(tmp10 - tmp12), 98);
  const int z2 = w16(ifast_mul(tmp10, 139) + z5);
  const int z4 = w16(ifast_mul(tmp12, 334) + z5);
  const int z3 = ifast_mul(tmp11, 181);
  const int z11 = w16(tmp7 + z3), z13 = w16(tmp7 - z3);
  d[5 * S] = w16(z13 + z2);
  d[3 * S] = w16(z13 - z2);
  d[S] = w16(z11 + z4);
  d[7 * S] = w16(z11 - z4);
}

__device__ __forceinline__ int descale_r(int x, int n) { return (x + (1 << (n - 1))) >> n; }
template <int S, int PASS>
__device__ __forceinline__ void fdct_islow_1d(int* d) {  // jfdctint.c, CONST_BITS 13, PASS1_BITS 2
  int tmp0 = d[0] + d[7 * S], tmp7 = d[0] - d[7 * S], tmp1 = d[S] + d[6 * S], tmp6 = d[S] - d[6 * S];
  int tmp2 = d[2 * S] + d[5 * S], tmp5 = d[2 * S] - d[5 * S], tmp3 = d[3 * S] + d[4 * S], tmp4 = d[3 * S] - d[4 * S];
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  constexpr int sh = PASS == 0 ? 13 - 2 : 13 + 2;
  if (PASS == 0) {
    d[0] = w16((tmp10 + tmp11) * 4);
    d[4 * S] = w16((tmp10 - tmp11) * 4);
  } else {
    d[0] = w16(descale_r(tmp10 + tmp11, 2));
    d[4 * S] = w16(descale_r(tmp10 - tmp11, 2));
  }
  int z1 = (tmp12 + tmp13) * 4433;
  d[2 * S] = w16(descale_r(z1 + tmp13 * 6270, sh));
  d[6 * S] = w16(descale_r(z1 + tmp12 * (-15137), sh));
  z1 = tmp4 + tmp7;
  int z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
  const int z5 = (z3 + z4) * 9633;
  tmp4 *= 2446, tmp5 *= 16819, tmp6 *= 25172, tmp7 *= 12299;
  z1 *= -7373, z2 *= -20995, z3 *= -16069, z4 *= -3196;
  z3 += z5, z4 += z5;
  d[7 * S] = w16(descale_r(tmp4 + z1 + z3, sh));
  d[5 * S] = w16(descale_r(tmp5 + z2 + z4, sh));
  d[3 * S] = w16(descale_r(tmp6 + z2 + z3, sh));
  d[S] = w16(descale_r(tmp7 + z1 + z4, sh));
}

// jccolor.c rgb_ycc_convert (16-bit fixed point)
__device__ __forceinline__ int ycc_y(int r, int g, int b) { return (19595 * r + 38470 * g + 7471 * b + 32768) >> 16; }
__device__ __forceinline__ int ycc_cb(int r, int g, int b) {
  return (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16;
}
__device__ __forceinline__ int ycc_cr(int r, int g, int b) {
  return (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16;
}

// jcdctmgr.c quantize(): ((|v| + corr) * recip) >> shift, sign restored
__device__ __forceinline__ int quant1(int v, const EncQuant& q, int t, int i) {
  const unsigned a = (unsigned)(v < 0 ? -v : v) & 0xFFFFu;
  const unsigned prod = ((a + q.corr[t][i]) & 0xFFFFu) * (unsigned)q.recip[t][i];
  const int r = (int)(short)(prod >> q.shift[t][i]);
  return v < 0 ? -r : r;
}

// RGB -> the encoder's sample planes: Y at full size and box-filtered Cb, Cr (h2v2), all padded to whole MCUs the way
// libjpeg pads them -- columns past the image repeat the last pixel (jcsample.c expand_right_edge), rows past it the
// last row of each component (jcprepct.c: the last chroma row that exists is made of rows 2g, min(2g + 1, h - 1)).
// One thread per 8 x 2 pixels: 2 x 24 bytes in, 2 x 8 luma + 2 x 4 chroma bytes out.
// Frame layout: Y [16 mcuy][16 mcux] | Cb [8 mcuy][8 mcux] | Cr.
__global__ __launch_bounds__(256) void k_enc_ycc(const JpegFrameDesc* __restrict__ descs, const uint8_t* __restrict__ rgb,
                                                 size_t rgb_stride, uint8_t* __restrict__ planes, size_t plane_stride) {
  const int frame = blockIdx.y;
  const int w = descs[frame].width, h = descs[frame].height;
  if (w <= 0 || h <= 0) return;
  const int mcux = (w + 15) >> 4, mcuy = (h + 15) >> 4;
  const int yw = 16 * mcux, cw = 8 * mcux, crows = 8 * mcuy;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 2 * mcux * crows) return;
  const int cy = t / (2 * mcux), x0 = (t - cy * 2 * mcux) * 8;
  const int groups = (h + 1) >> 1;
  const int cyc = min(cy, groups - 1);
  const int r0 = 2 * cyc, r1 = min(2 * cyc + 1, h - 1);
  const uint8_t* img = rgb + (size_t)frame * rgb_stride;
  const uint8_t* p0 = img + (size_t)r0 * w * 3;
  const uint8_t* p1 = img + (size_t)r1 * w * 3;
  int px[2][24];
  if (x0 + 8 <= w && ((reinterpret_cast<uintptr_t>(p0 + 3 * x0) | reinterpret_cast<uintptr_t>(p1 + 3 * x0)) & 3) == 0) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const uint32_t* q = reinterpret_cast<const uint32_t*>((r ? p1 : p0) + 3 * x0);
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const uint32_t v = q[i];
        px[r][4 * i] = v & 255, px[r][4 * i + 1] = (v >> 8) & 255, px[r][4 * i + 2] = (v >> 16) & 255, px[r][4 * i + 3] = v >> 24;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int x = min(x0 + j, w - 1);
#pragma unroll
      for (int c = 0; c < 3; c++) px[0][3 * j + c] = p0[3 * x + c], px[1][3 * j + c] = p1[3 * x + c];
    }
  }
  uint8_t* fp = planes + (size_t)frame * plane_stride;
  uint32_t yv[2][2] = {{0, 0}, {0, 0}};
  int cb[2][8], cr[2][8];
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int R = px[r][3 * j], G = px[r][3 * j + 1], B = px[r][3 * j + 2];
      yv[r][j >> 2] |= (uint32_t)ycc_y(R, G, B) << (8 * (j & 3));
      cb[r][j] = ycc_cb(R, G, B), cr[r][j] = ycc_cr(R, G, B);
    }
  // luma rows 2cy and 2cy + 1: below the image both repeat row h - 1 (= r1 there)
  const bool below = cy >= groups;
  uint8_t* yo = fp + (size_t)(2 * cy) * yw + x0;
  *reinterpret_cast<uint2*>(yo) = below ? make_uint2(yv[1][0], yv[1][1]) : make_uint2(yv[0][0], yv[0][1]);
  *reinterpret_cast<uint2*>(yo + yw) = make_uint2(yv[1][0], yv[1][1]);
  uint32_t cbo = 0, cro = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {  // h2v2_downsample: bias 1, 2, 1, 2, ... by output column (x0 / 2 is a multiple of 4)
    const int bias = (k & 1) ? 2 : 1;
    cbo |= (uint32_t)((cb[0][2 * k] + cb[0][2 * k + 1] + cb[1][2 * k] + cb[1][2 * k + 1] + bias) >> 2) << (8 * k);
    cro |= (uint32_t)((cr[0][2 * k] + cr[0][2 * k + 1] + cr[1][2 * k] + cr[1][2 * k + 1] + bias) >> 2) << (8 * k);
  }
  const size_t ysz = (size_t)yw * 16 * mcuy, csz = (size_t)cw * crows;
  *reinterpret_cast<uint32_t*>(fp + ysz + (size_t)cy * cw + (x0 >> 1)) = cbo;
  *reinterpret_cast<uint32_t*>(fp + ysz + csz + (size_t)cy * cw + (x0 >> 1)) = cro;
}

// One thread per 8x8 block of the planes: forward DCT + quantisation -> [mcu][Y00 Y01 Y10 Y11 Cb Cr][64] in zigzag order.
template <bool IFAST>
__global__ __launch_bounds__(256) void k_enc_fdct(const JpegFrameDesc* __restrict__ descs, const uint8_t* __restrict__ planes,
                                                  size_t plane_stride, EncQuant q, int16_t* __restrict__ coef, size_t coef_stride) {
  const int frame = blockIdx.y;
  const int w = descs[frame].width, h = descs[frame].height;
  if (w <= 0 || h <= 0) return;
  const int mcux = (w + 15) >> 4, mcuy = (h + 15) >> 4, nmcu = mcux * mcuy;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 6 * nmcu) return;
  const uint8_t* fp = planes + (size_t)frame * plane_stride;
  const int yw = 16 * mcux, cw = 8 * mcux;
  int comp, bx, by, mcu, blk;
  if (t < 4 * nmcu) {
    comp = 0, by = t / (2 * mcux), bx = t - by * 2 * mcux;
    mcu = (by >> 1) * mcux + (bx >> 1), blk = (by & 1) * 2 + (bx & 1);
  } else {
    const int u = t - 4 * nmcu;
    comp = 1 + u / nmcu;
    mcu = u - (comp - 1) * nmcu;
    by = mcu / mcux, bx = mcu - by * mcux, blk = 3 + comp;
  }
  int16_t* out = coef + (size_t)frame * coef_stride + ((size_t)mcu * 6 + blk) * 64;
  const int tq = comp ? 1 : 0;
  bool dummy = false;
  const uint8_t* src;
  int pitch;
  if (comp == 0) {
    // jccoefct.c: blocks beyond the component's own block grid (odd block counts) are dummies -- AC 0, DC = the DC of
    // the block before: the block to the left, or for a whole dummy row block 1 of the MCU (itself a dummy of block 0
    // when that column is one)
    const int ybw = (w + 7) >> 3, ybh = (h + 7) >> 3;
    int sbx = bx, sby = by;
    if (by >= ybh) dummy = true, sby = by - 1, sbx = (bx | 1) < ybw ? (bx | 1) : (bx & ~1);
    else if (bx >= ybw) dummy = true, sbx = bx - 1;
    src = fp + (size_t)sby * 8 * yw + sbx * 8, pitch = yw;
  } else {
    const size_t ysz = (size_t)yw * 16 * mcuy, csz = (size_t)cw * 8 * mcuy;
    src = fp + ysz + (comp == 2 ? csz : 0) + (size_t)by * 8 * cw + bx * 8, pitch = cw;
  }
  int ws[64];
#pragma unroll
  for (int yy = 0; yy < 8; yy++) {
    const uint2 v = *reinterpret_cast<const uint2*>(src + (size_t)yy * pitch);
#pragma unroll
    for (int xx = 0; xx < 8; xx++) ws[yy * 8 + xx] = (int)(((xx < 4 ? v.x : v.y) >> (8 * (xx & 3))) & 255) - 128;
  }
  uint4* o4 = reinterpret_cast<uint4*>(out);
  if (dummy) {
    // DC of either DCT = the sum of the 64 centred samples (16-bit wrap for the fast one, exact for the accurate one)
    int s = 0;
#pragma unroll
    for (int i = 0; i < 64; i++) s += ws[i];
    const int dc = quant1(IFAST ? w16(s) : s, q, tq, 0) & 0xFFFF;
    o4[0] = make_uint4((unsigned)dc, 0, 0, 0);
#pragma unroll
    for (int i = 1; i < 8; i++) o4[i] = make_uint4(0, 0, 0, 0);
    return;
  }
#pragma unroll
  for (int r = 0; r < 8; r++) {
    if (IFAST) fdct_ifast_1d<1>(ws + 8 * r);
    else fdct_islow_1d<1, 0>(ws + 8 * r);
  }
#pragma unroll
  for (int c = 0; c < 8; c++) {
    if (IFAST) fdct_ifast_1d<8>(ws + c);
    else fdct_islow_1d<8, 1>(ws + c);
  }
  // zigzag order, 16-byte stores
  constexpr int zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                          41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                          30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
#pragma unroll
  for (int g = 0; g < 8; g++) {
    unsigned v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int a = quant1(ws[zz[8 * g + 2 * j]], q, tq, zz[8 * g + 2 * j]) & 0xFFFF;
      const int b = quant1(ws[zz[8 * g + 2 * j + 1]], q, tq, zz[8 * g + 2 * j + 1]) & 0xFFFF;
      v[j] = (unsigned)a | ((unsigned)b << 16);
    }
    o4[g] = make_uint4(v[0], v[1], v[2], v[3]);
  }
}

// ---- Huffman coding (jchuff.c encode_one_block) ----
// tables: [2][272] words, (size << 16) | code: 16 DC entries then 256 AC entries of the luma / chroma set
constexpr int kEncTabWords = 272;

__device__ __forceinline__ int prev_block_of(int i) {  // block that holds the previous DC of block i's component, -1: none
  const int mcu = i / 6, blk = i - mcu * 6;
  if (blk >= 1 && blk <= 3) return i - 1;
  return mcu == 0 ? -1 : (blk == 0 ? i - 3 : i - 6);  // Y00 follows Y11 of the MCU before; Cb / Cr follow their own
}

template <class Emit>
__device__ __forceinline__ void encode_block(const uint4* __restrict__ blk4, int prev_dc, const uint32_t* __restrict__ tab,
                                             Emit&& emit) {
  const uint32_t* dc_tab = tab;
  const uint32_t* ac_tab = tab + 16;
  int run = 0;
  auto coefficient = [&](int v, bool is_dc) {
    if (is_dc) {
      const int diff = v - prev_dc;
      const int nb = diff ? 32 - __clz(diff < 0 ? -diff : diff) : 0;
      const uint32_t e = dc_tab[nb];
      const uint32_t bits = (uint32_t)(diff < 0 ? diff - 1 : diff) & ((1u << nb) - 1u);
      emit(((e & 0xFFFFu) << nb) | bits, (int)(e >> 16) + nb);
      return;
    }
    if (v == 0) {
      run++;
      return;
    }
    while (run > 15) {
      const uint32_t z = ac_tab[0xF0];
      emit(z & 0xFFFFu, (int)(z >> 16));
      run -= 16;
    }
    const int nb = 32 - __clz(v < 0 ? -v : v);
    const uint32_t e = ac_tab[(run << 4) + nb];
    const uint32_t bits = (uint32_t)(v < 0 ? v - 1 : v) & ((1u << nb) - 1u);
    emit(((e & 0xFFFFu) << nb) | bits, (int)(e >> 16) + nb);
    run = 0;
  };
#pragma unroll
  for (int g = 0; g < 8; g++) {
    const uint4 q = blk4[g];
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      coefficient((int)(short)(w[j] & 0xFFFFu), g == 0 && j == 0);
      coefficient((int)(short)(w[j] >> 16), false);
    }
  }
  if (run > 0) {
    const uint32_t e = ac_tab[0];
    emit(e & 0xFFFFu, (int)(e >> 16));
  }
}

__device__ __forceinline__ void load_tables(uint32_t* s_tab, const uint32_t* __restrict__ tabs) {
  for (int i = threadIdx.x; i < 2 * kEncTabWords; i += blockDim.x) s_tab[i] = tabs[i];
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_enc_bits(const JpegFrameDesc* __restrict__ descs, const int16_t* __restrict__ coef,
                                                  size_t coef_stride, const uint32_t* __restrict__ tabs,
                                                  uint32_t* __restrict__ bits, size_t blk_stride) {
  __shared__ uint32_t s_tab[2 * kEncTabWords];
  load_tables(s_tab, tabs);
  const int frame = blockIdx.y;
  const int w = descs[frame].width, h = descs[frame].height;
  if (w <= 0 || h <= 0) return;
  const int nblk = 6 * ((w + 15) >> 4) * ((h + 15) >> 4);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nblk) return;
  const int16_t* fc = coef + (size_t)frame * coef_stride;
  const int p = prev_block_of(i);
  const int prev_dc = p < 0 ? 0 : fc[(size_t)p * 64];
  int total = 0;
  encode_block(reinterpret_cast<const uint4*>(fc + (size_t)i * 64), prev_dc, s_tab + ((i % 6) >= 4 ? kEncTabWords : 0),
               [&](uint32_t, int n) { total += n; });
  bits[(size_t)frame * blk_stride + i] = (uint32_t)total;
}

// one workgroup per frame: bits[] -> exclusive prefix (in place), total_bits[frame]; zeroes the frame's bit buffer
__global__ __launch_bounds__(1024) void k_enc_scan(const JpegFrameDesc* __restrict__ descs, uint32_t* __restrict__ bits,
                                                   size_t blk_stride, uint32_t* __restrict__ total_bits,
                                                   uint32_t* __restrict__ words, size_t word_stride) {
  __shared__ uint32_t s_sum[1024];
  const int frame = blockIdx.x;
  const int w = descs[frame].width, h = descs[frame].height;
  if (w <= 0 || h <= 0) {
    if (threadIdx.x == 0) total_bits[frame] = 0;
    return;
  }
  const int nblk = 6 * ((w + 15) >> 4) * ((h + 15) >> 4);
  uint32_t* b = bits + (size_t)frame * blk_stride;
  const int per = (nblk + 1023) >> 10, i0 = threadIdx.x * per, i1 = min(i0 + per, nblk);
  uint32_t s = 0;
  for (int i = i0; i < i1; i++) s += b[i];
  s_sum[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan (Hillis-Steele)
    const uint32_t v = threadIdx.x >= (unsigned)d ? s_sum[threadIdx.x - d] : 0u;
    __syncthreads();
    s_sum[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t off = s_sum[threadIdx.x] - s;
  for (int i = i0; i < i1; i++) {
    const uint32_t n = b[i];
    b[i] = off;
    off += n;
  }
  const uint32_t total = s_sum[1023];
  if (threadIdx.x == 0) total_bits[frame] = total;
  uint32_t* wd = words + (size_t)frame * word_stride;
  const uint32_t nwords = (total >> 5) + 2;  // (the stuffing pass reads whole 16-byte groups: the launcher pads the stride)
  for (uint32_t i = threadIdx.x; i < nwords; i += 1024) wd[i] = 0u;
}

// The bit stream of a frame is kept as 32-bit words, most significant bit first.
__global__ __launch_bounds__(256) void k_enc_write(const JpegFrameDesc* __restrict__ descs, const int16_t* __restrict__ coef,
                                                   size_t coef_stride, const uint32_t* __restrict__ tabs,
                                                   const uint32_t* __restrict__ offs, size_t blk_stride,
                                                   uint32_t* __restrict__ words, size_t word_stride) {
  __shared__ uint32_t s_tab[2 * kEncTabWords];
  load_tables(s_tab, tabs);
  const int frame = blockIdx.y;
  const int w = descs[frame].width, h = descs[frame].height;
  if (w <= 0 || h <= 0) return;
  const int nblk = 6 * ((w + 15) >> 4) * ((h + 15) >> 4);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nblk) return;
  const int16_t* fc = coef + (size_t)frame * coef_stride;
  const int p = prev_block_of(i);
  const int prev_dc = p < 0 ? 0 : fc[(size_t)p * 64];
  const uint32_t start = offs[(size_t)frame * blk_stride + i];
  uint32_t* wd = words + (size_t)frame * word_stride;
  uint32_t widx = start >> 5;
  const uint32_t first = widx;
  unsigned long long acc = 0;  // pending bits, left-aligned at bit 63 - (start & 31)
  int fill = (int)(start & 31);  // bits of the current word already used (by the blocks before, for the first word)
  encode_block(reinterpret_cast<const uint4*>(fc + (size_t)i * 64), prev_dc, s_tab + ((i % 6) >= 4 ? kEncTabWords : 0),
               [&](uint32_t code, int n) {  // n <= 27
                 acc |= (unsigned long long)code << (64 - fill - n);
                 fill += n;
                 if (fill >= 32) {
                   const uint32_t out = (uint32_t)(acc >> 32);
                   if (widx == first) atomicOr(&wd[widx], out);  // shared with the blocks before
                   else wd[widx] = out;                          // only this block writes this word
                   widx++;
                   acc <<= 32;
                   fill -= 32;
                 }
               });
  if (fill > 0) atomicOr(&wd[widx], (uint32_t)(acc >> 32));  // shared with the blocks after
}

__device__ __forceinline__ uint32_t stream_bytes(uint32_t total_bits) { return (total_bits + 7) >> 3; }
// byte j (0..3) of stream word w; the last byte of the stream is padded with ones (jchuff.c flush_bits)
__device__ __forceinline__ uint32_t word_with_pad(uint32_t w, uint32_t widx, uint32_t total_bits) {
  if ((total_bits & 7) && widx == ((total_bits - 1) >> 5)) {
    const uint32_t used = total_bits & 31;  // != 0 here
    const uint32_t padbits = 8 - (total_bits & 7);
    w |= ((1u << padbits) - 1u) << (32 - used - padbits);
  }
  return w;
}
__device__ __forceinline__ int count_ff(uint32_t w, int nbytes /* leading bytes of the word that belong to the stream */) {
  int c = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) c += (j < nbytes && ((w >> (24 - 8 * j)) & 0xFF) == 0xFF) ? 1 : 0;
  return c;
}

constexpr int kEncChunk = 4096;  // bytes of bit stream per workgroup of the stuffing passes (16 per thread)

__global__ __launch_bounds__(256) void k_enc_ffcount(const uint32_t* __restrict__ total_bits, const uint32_t* __restrict__ words,
                                                     size_t word_stride, uint32_t* __restrict__ chunk_ff, size_t chunk_stride) {
  __shared__ int s_c[256];
  const int frame = blockIdx.y;
  const uint32_t tb = total_bits[frame], nbytes = stream_bytes(tb);
  const uint32_t base = blockIdx.x * kEncChunk;
  if (base >= nbytes) return;
  const uint32_t* wd = words + (size_t)frame * word_stride;
  const uint32_t b0 = base + threadIdx.x * 16;
  int c = 0;
  if (b0 < nbytes) {
    const uint4 q = *reinterpret_cast<const uint4*>(wd + (b0 >> 2));
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int nb = (int)min(4u, nbytes > b0 + 4 * j ? nbytes - (b0 + 4 * j) : 0u);
      c += count_ff(word_with_pad(w[j], (b0 >> 2) + j, tb), nb);
    }
  }
  s_c[threadIdx.x] = c;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) s_c[threadIdx.x] += s_c[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) chunk_ff[(size_t)frame * chunk_stride + blockIdx.x] = (uint32_t)s_c[0];
}

// one workgroup: length of every frame's finished stream and its offset in the packed output
__global__ __launch_bounds__(256) void k_enc_layout(const uint32_t* __restrict__ total_bits, const uint32_t* __restrict__ chunk_ff,
                                                    size_t chunk_stride, uint32_t count, uint32_t fixed_bytes /* framing + header + EOI */,
                                                    uint32_t* __restrict__ out_len, uint32_t* __restrict__ out_off,
                                                    uint32_t* __restrict__ out_total) {
  __shared__ uint32_t s_len[UFD_ENC_MAX_FRAMES];
  for (uint32_t f = threadIdx.x; f < count; f += 256) {
    const uint32_t nbytes = stream_bytes(total_bits[f]);
    uint32_t nff = 0;
    const uint32_t nch = (nbytes + kEncChunk - 1) / kEncChunk;
    for (uint32_t c = 0; c < nch; c++) nff += chunk_ff[(size_t)f * chunk_stride + c];
    s_len[f] = nbytes ? fixed_bytes + nbytes + nff : 0u;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t off = 0;
    for (uint32_t f = 0; f < count; f++) {
      out_len[f] = s_len[f];
      out_off[f] = off;
      off += (s_len[f] + 15u) & ~15u;
    }
    *out_total = off;
  }
}

__global__ __launch_bounds__(256) void k_enc_stuff(const JpegFrameDesc* __restrict__ descs, const uint32_t* __restrict__ total_bits,
                                                   const uint32_t* __restrict__ words, size_t word_stride,
                                                   const uint32_t* __restrict__ chunk_ff, size_t chunk_stride,
                                                   const uint8_t* __restrict__ header, uint32_t pre_len, uint32_t hdr_len,
                                                   uint32_t dim_off, uint32_t post_len, const uint32_t* __restrict__ out_off,
                                                   uint8_t* __restrict__ out) {
  __shared__ uint32_t s_c[256];
  __shared__ uint32_t s_before;
  const int frame = blockIdx.y;
  const uint32_t tb = total_bits[frame], nbytes = stream_bytes(tb);
  const uint32_t base = blockIdx.x * kEncChunk;
  if (base >= nbytes) return;
  uint8_t* o = out + out_off[frame];
  if (blockIdx.x == 0) {  // framing prefix + header with this frame's SOF0 dimensions
    const int w = descs[frame].width, h = descs[frame].height;
    for (uint32_t i = threadIdx.x; i < pre_len + hdr_len; i += 256) {
      uint8_t v = header[i];
      const uint32_t k = i - pre_len;  // (wraps for the prefix: never equal to dim_off + ...)
      if (k == dim_off) v = (uint8_t)(h >> 8);
      if (k == dim_off + 1) v = (uint8_t)h;
      if (k == dim_off + 2) v = (uint8_t)(w >> 8);
      if (k == dim_off + 3) v = (uint8_t)w;
      o[i] = v;
    }
  }
  // 0xFF bytes in the chunks before this one
  uint32_t part = 0;
  for (uint32_t c = threadIdx.x; c < blockIdx.x; c += 256) part += chunk_ff[(size_t)frame * chunk_stride + c];
  s_c[threadIdx.x] = part;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) s_c[threadIdx.x] += s_c[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) s_before = s_c[0];
  __syncthreads();
  const uint32_t before = s_before;
  __syncthreads();
  // this thread's 16 bytes
  const uint32_t* wd = words + (size_t)frame * word_stride;
  const uint32_t b0 = base + threadIdx.x * 16;
  uint32_t w[4] = {0, 0, 0, 0};
  int nb[4] = {0, 0, 0, 0};
  uint32_t c = 0;
  if (b0 < nbytes) {
    const uint4 q = *reinterpret_cast<const uint4*>(wd + (b0 >> 2));
    const uint32_t ww[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      nb[j] = (int)min(4u, nbytes > b0 + 4 * j ? nbytes - (b0 + 4 * j) : 0u);
      w[j] = word_with_pad(ww[j], (b0 >> 2) + j, tb);
      c += (uint32_t)count_ff(w[j], nb[j]);
    }
  }
  s_c[threadIdx.x] = c;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const uint32_t v = threadIdx.x >= (unsigned)d ? s_c[threadIdx.x - d] : 0u;
    __syncthreads();
    s_c[threadIdx.x] += v;
    __syncthreads();
  }
  if (b0 >= nbytes) return;
  uint8_t* p = o + pre_len + hdr_len + b0 + before + (s_c[threadIdx.x] - c);
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < nb[j]) {
        const uint8_t v = (uint8_t)(w[j] >> (24 - 8 * k));
        *p++ = v;
        if (v == 0xFF) *p++ = 0;
      }
  if (b0 + 16 >= nbytes) {  // the thread that wrote the last byte: EOI and the framing suffix
    *p++ = 0xFF, *p++ = 0xD9;
    for (uint32_t i = 0; i < post_len; i++) *p++ = header[pre_len + hdr_len + i];
  }
}

}  // namespace

void launch_draw_rects(const JpegFrameDesc* d_descs, const Det* d_dets, uint32_t det_stride, const uint32_t* d_ndet, uint8_t* d_rgb,
                       size_t rgb_stride, float label_w, float label_h, uint32_t count, hipStream_t s) {
  if (!count) return;
  hipLaunchKernelGGL(k_draw_rects, dim3(64, count), dim3(256), 0, s, d_descs, d_dets, det_stride, d_ndet, d_rgb, rgb_stride, label_w,
                     label_h);
}

void launch_jpeg_encode(const JpegFrameDesc* d_descs, const uint8_t* d_rgb, size_t rgb_stride, uint32_t max_w, uint32_t max_h,
                        uint32_t count, const EncQuant& q, bool ifast, const EncBuffers& e, hipStream_t s, const EncStageHook* hook) {
  if (!count) return;
  const uint32_t max_blocks = 6 * ((max_w + 15) / 16) * ((max_h + 15) / 16);
  const dim3 gblk((max_blocks + 255) / 256, count);
  const uint32_t max_chunks = (uint32_t)e.chunk_stride;
  auto stage = [&](const char* name, bool begin) {
    if (hook) (*hook)(name, begin);
  };
  stage("enc_fdct", true);
  const uint32_t ycc_threads = 2 * ((max_w + 15) / 16) * 8 * ((max_h + 15) / 16);
  hipLaunchKernelGGL(k_enc_ycc, dim3((ycc_threads + 255) / 256, count), dim3(256), 0, s, d_descs, d_rgb, rgb_stride, e.planes,
                     e.plane_stride);
  if (ifast) hipLaunchKernelGGL(k_enc_fdct<true>, gblk, dim3(256), 0, s, d_descs, e.planes, e.plane_stride, q, e.coef, e.coef_stride);
  else hipLaunchKernelGGL(k_enc_fdct<false>, gblk, dim3(256), 0, s, d_descs, e.planes, e.plane_stride, q, e.coef, e.coef_stride);
  stage("enc_fdct", false);
  stage("enc_huffman", true);
  hipLaunchKernelGGL(k_enc_bits, gblk, dim3(256), 0, s, d_descs, e.coef, e.coef_stride, e.tables, e.bits, e.blk_stride);
  hipLaunchKernelGGL(k_enc_scan, dim3(count), dim3(1024), 0, s, d_descs, e.bits, e.blk_stride, e.total_bits, e.words, e.word_stride);
  hipLaunchKernelGGL(k_enc_write, gblk, dim3(256), 0, s, d_descs, e.coef, e.coef_stride, e.tables, e.bits, e.blk_stride, e.words,
                     e.word_stride);
  stage("enc_huffman", false);
  stage("enc_stuff", true);
  const dim3 gch(max_chunks, count);
  hipLaunchKernelGGL(k_enc_ffcount, gch, dim3(256), 0, s, e.total_bits, e.words, e.word_stride, e.chunk_ff, e.chunk_stride);
  hipLaunchKernelGGL(k_enc_layout, dim3(1), dim3(256), 0, s, e.total_bits, e.chunk_ff, e.chunk_stride, count,
                     e.pre_len + e.hdr_len + 2 + e.post_len, e.out_len, e.out_off, e.out_total);
  hipLaunchKernelGGL(k_enc_stuff, gch, dim3(256), 0, s, d_descs, e.total_bits, e.words, e.word_stride, e.chunk_ff, e.chunk_stride,
                     e.header, e.pre_len, e.hdr_len, e.dim_off, e.post_len, e.out_off, e.out);
  stage("enc_stuff", false);
}

}  // namespace ufd
