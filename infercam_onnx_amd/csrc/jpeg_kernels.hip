// jpeg_kernels.hip -- device half of row A1 (turbojpeg::decompress_image, inferer.rs:35):
// dequantisation + accurate integer IDCT (libjpeg "islow"), fancy chroma upsampling and the
// libjpeg fixed-point YCbCr->RGB conversion, bit-exact with libjpeg-turbo's defaults
// (tjDecompress2 flags = 0).  Integer/byte work, HBM-bound: 16-byte coalesced coefficient
// loads staged through LDS, 8-byte row stores, no matrix cores.
#include "kernels.hpp"

namespace ufd {
namespace {

constexpr int kBlocksPerWG = 32;  // 32 DCT blocks (4 KiB of coefficients) per 256-thread workgroup
constexpr int kWsStride = 65;     // padded LDS row so that pass-2 lanes (one block each) hit distinct banks
// ... and the coefficient blocks in LDS 33 dwords apart, one lane per block in pass 1 too (round 5).  With blocks 64 int16 = 32
// dwords apart and a wave covering 8 blocks x 8 columns, the eight blocks' copies of one coefficient sat in the same bank
// (8-way conflict on each of the 8 reads of a thread) and the column results of blocks lb, columns c with equal lb + c in the
// same bank of the padded workspace: SQ_LDS_BANK_CONFLICT was 276 % of the kernel's LDS-active cycles.
constexpr int kInStride = 66;     // int16 per block in LDS

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

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// libjpeg post-IDCT range-limit table: index (x & 1023) into {128..255, 255 x384, 0 x384, 0..127}
__device__ __forceinline__ unsigned range_limit(int x) {
  int i = x & 1023;
  return i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896));
}

// one 1-D pass of the islow IDCT on 8 inputs (already dequantised / from the workspace)
__device__ __forceinline__ void idct_1d(const int (&x)[8], int (&o)[8], int shift) {
  int z2 = x[2], z3 = x[6];
  int z1 = (z2 + z3) * FIX_0_541196100;
  int tmp2 = z1 + z3 * (-FIX_1_847759065);
  int tmp3 = z1 + z2 * FIX_0_765366865;
  int tmp0 = (x[0] + x[4]) * 8192;
  int tmp1 = (x[0] - x[4]) * 8192;
  int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = x[7];
  tmp1 = x[5];
  tmp2 = x[3];
  tmp3 = x[1];
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
  o[0] = descale(tmp10 + tmp3, shift);
  o[7] = descale(tmp10 - tmp3, shift);
  o[1] = descale(tmp11 + tmp2, shift);
  o[6] = descale(tmp11 - tmp2, shift);
  o[2] = descale(tmp12 + tmp1, shift);
  o[5] = descale(tmp12 - tmp1, shift);
  o[3] = descale(tmp13 + tmp0, shift);
  o[4] = descale(tmp13 - tmp0, shift);
}

// natural (row-major) coefficient index -> position in zigzag order
__constant__ uint8_t c_nat_to_zig[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
                                         41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
                                         46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

// ZZ: the blocks hold their coefficients in zigzag order (device entropy decoder without restart
// markers) instead of natural order.
template <bool ZZ>
__global__ __launch_bounds__(256) void k_idct(const JpegFrameDesc* __restrict__ descs, const int16_t* __restrict__ coef,
                                              size_t coef_stride, uint8_t* __restrict__ planes, size_t plane_stride,
                                              const int16_t* __restrict__ dc, size_t dc_stride) {
  __shared__ __attribute__((aligned(16))) int16_t s_in[kBlocksPerWG * kInStride];
  __shared__ int s_ws[kBlocksPerWG * kWsStride];
  const int frame = blockIdx.y;
  const JpegFrameDesc& d = descs[frame];
  const uint32_t g0 = blockIdx.x * kBlocksPerWG;
  const uint32_t total = d.total_blocks;
  if (g0 >= total) return;  // whole workgroup exits together
  const int nblk = min((uint32_t)kBlocksPerWG, total - g0);
  const int tid = threadIdx.x;
  const int16_t* src = coef + (size_t)frame * coef_stride + (size_t)g0 * 64;
  if (tid * 8 < nblk * 64) {
    uint4 v = *reinterpret_cast<const uint4*>(src + tid * 8);
    // the device entropy decoder keeps the DC terms in a compact side array (the thread that loads a block's head patches it in)
    if (dc && (tid & 7) == 0) v.x = (v.x & 0xFFFF0000u) | (uint32_t)(uint16_t)dc[(size_t)frame * dc_stride + g0 + (tid >> 3)];
    uint32_t* dst = reinterpret_cast<uint32_t*>(&s_in[(tid >> 3) * kInStride + (tid & 7) * 8]);  // (4-byte aligned: the stride is odd in dwords)
    dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
  }
  __syncthreads();
  // pass 1: columns.  thread -> (column, local block): a wave covers 2 columns x 32 neighbouring blocks, like pass 2
  {
    const int lb = tid & 31, c = tid >> 5;
    if (lb < nblk) {
      const uint32_t g = g0 + lb;
      const int comp = (d.ncomp > 1 && g * 64 >= d.coef_off[1]) ? ((g * 64 >= d.coef_off[2]) ? 2 : 1) : 0;
      const uint16_t* q = d.qt[comp];
      int x[8], o[8];
#pragma unroll
      for (int r = 0; r < 8; r++) x[r] = (int)s_in[lb * kInStride + (ZZ ? (int)c_nat_to_zig[r * 8 + c] : r * 8 + c)] * (int)q[r * 8 + c];
      idct_1d(x, o, 11);  // CONST_BITS - PASS1_BITS
#pragma unroll
      for (int r = 0; r < 8; r++) s_ws[lb * kWsStride + r * 8 + c] = o[r];
    }
  }
  __syncthreads();
  // pass 2: rows.  thread -> (row, local block): a wave covers 2 rows x 32 neighbouring blocks
  {
    const int lb = tid & 31, r = tid >> 5;
    if (lb < nblk) {
      const uint32_t g = g0 + lb;
      const int comp = (d.ncomp > 1 && g * 64 >= d.coef_off[1]) ? ((g * 64 >= d.coef_off[2]) ? 2 : 1) : 0;
      const uint32_t local = g - d.coef_off[comp] / 64;
      const uint32_t wb = d.wblk[comp];
      const uint32_t by = local / wb, bx = local - by * wb;
      int x[8], o[8];
#pragma unroll
      for (int j = 0; j < 8; j++) x[j] = s_ws[lb * kWsStride + r * 8 + j];
      idct_1d(x, o, 18);  // CONST_BITS + PASS1_BITS + 3
      uint2 v;
      v.x = range_limit(o[0]) | (range_limit(o[1]) << 8) | (range_limit(o[2]) << 16) | (range_limit(o[3]) << 24);
      v.y = range_limit(o[4]) | (range_limit(o[5]) << 8) | (range_limit(o[6]) << 16) | (range_limit(o[7]) << 24);
      uint8_t* dst = planes + (size_t)frame * plane_stride + d.plane_off[comp] + (size_t)(by * 8 + r) * (wb * 8) + bx * 8;
      *reinterpret_cast<uint2*>(dst) = v;
    }
  }
}

// ---- jdsample.c: value of component `c` at full-resolution pixels x0..x0+3 of row y ----
__device__ __forceinline__ void upsample4(const JpegFrameDesc& d, const uint8_t* __restrict__ fplanes, int c, int x0,
                                          int y, int (&out)[4]) {
  const uint8_t* pl = fplanes + d.plane_off[c];
  const int pitch = d.wblk[c] * 8;
  const int hx = d.hmax / d.h[c], vx = d.vmax / d.v[c];
  const int dw = d.dw[c], dh = d.dh[c];
  if (hx == 1 && vx == 1) {
    const uint8_t* row = pl + (size_t)y * pitch;
#pragma unroll
    for (int j = 0; j < 4; j++) out[j] = row[min(x0 + j, pitch - 1)];
    return;
  }
  const bool fancy = dw > 2;  // jdsample.c: fancy upsampling only when downsampled_width > 2
  if (hx == 2 && vx == 1) {
    const uint8_t* row = pl + (size_t)y * pitch;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int x = x0 + j, i = min(x >> 1, dw - 1);
      const int cur = row[i];
      int v;
      if (!fancy) {
        v = cur;
      } else if (x & 1) {
        v = (i == dw - 1) ? cur : (cur * 3 + row[i + 1] + 2) >> 2;
      } else {
        v = (i == 0) ? cur : (cur * 3 + row[i - 1] + 1) >> 2;
      }
      out[j] = v;
    }
    return;
  }
  if (hx == 2 && vx == 2) {
    const int iy = y >> 1;
    const int ny = max(0, min(dh - 1, (y & 1) ? iy + 1 : iy - 1));
    const uint8_t* r0 = pl + (size_t)iy * pitch;
    const uint8_t* r1 = pl + (size_t)ny * pitch;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int x = x0 + j, i = min(x >> 1, dw - 1);
      int v;
      if (!fancy) {
        v = r0[i];
      } else {
        const int cur = r0[i] * 3 + r1[i];
        if (x & 1) {
          v = (i == dw - 1) ? (cur * 4 + 7) >> 4 : (cur * 3 + (r0[i + 1] * 3 + r1[i + 1]) + 7) >> 4;
        } else {
          v = (i == 0) ? (cur * 4 + 8) >> 4 : (cur * 3 + (r0[i - 1] * 3 + r1[i - 1]) + 8) >> 4;
        }
      }
      out[j] = v;
    }
    return;
  }
  if (hx == 1 && vx == 2) {  // 4:4:0: h1v2 fancy (no width condition in jdsample.c)
    const int iy = y >> 1;
    const int ny = max(0, min(dh - 1, (y & 1) ? iy + 1 : iy - 1));
    const uint8_t* r0 = pl + (size_t)iy * pitch;
    const uint8_t* r1 = pl + (size_t)ny * pitch;
    const int bias = (y & 1) ? 2 : 1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = min(x0 + j, pitch - 1);
      out[j] = (r0[i] * 3 + r1[i] + bias) >> 2;
    }
    return;
  }
  {  // every other integral expansion (4:1:1, 4:1:0, 4:4:1, 3x, ...): jdsample.c int_upsample, plain replication
    const uint8_t* row = pl + (size_t)(y / vx) * pitch;
#pragma unroll
    for (int j = 0; j < 4; j++) out[j] = row[min((x0 + j) / hx, pitch - 1)];
  }
}

__device__ __forceinline__ int clamp255(int v) { return min(255, max(0, v)); }

// -> r,g,b for 4 pixels (jdcolor.c ycc_rgb_convert fixed point, SCALEBITS 16)
__device__ __forceinline__ void pixels4(const JpegFrameDesc& d, const uint8_t* __restrict__ fplanes, int x0, int y,
                                        int (&r)[4], int (&g)[4], int (&b)[4]) {
  int c0[4];
  upsample4(d, fplanes, 0, x0, y, c0);
  if (d.color == kColorGray) {
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = g[j] = b[j] = c0[j];
    return;
  }
  int c1[4], c2[4];
  upsample4(d, fplanes, 1, x0, y, c1);
  upsample4(d, fplanes, 2, x0, y, c2);
  if (d.color == kColorRGB) {
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = c0[j], g[j] = c1[j], b[j] = c2[j];
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int yy = c0[j], cb = c1[j] - 128, cr = c2[j] - 128;
    r[j] = clamp255(yy + ((91881 * cr + 32768) >> 16));
    g[j] = clamp255(yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
    b[j] = clamp255(yy + ((116130 * cb + 32768) >> 16));
  }
}

__global__ __launch_bounds__(256) void k_upsample_rgb(const JpegFrameDesc* __restrict__ descs,
                                                      const uint8_t* __restrict__ planes, size_t plane_stride,
                                                      uint8_t* __restrict__ rgb, size_t rgb_stride) {
  const int frame = blockIdx.z;
  const JpegFrameDesc& d = descs[frame];
  const int y = blockIdx.y;
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (y >= d.height || x0 >= d.width) return;
  int r[4], g[4], b[4];
  pixels4(d, planes + (size_t)frame * plane_stride, x0, y, r, g, b);
  uint8_t* o = rgb + (size_t)frame * rgb_stride + ((size_t)y * d.width + x0) * 3;
  if (x0 + 4 <= d.width && ((reinterpret_cast<uintptr_t>(o) & 3) == 0)) {
    uint32_t w0 = r[0] | (g[0] << 8) | (b[0] << 16) | (r[1] << 24);
    uint32_t w1 = g[1] | (b[1] << 8) | (r[2] << 16) | (g[2] << 24);
    uint32_t w2 = b[2] | (r[3] << 8) | (g[3] << 16) | (b[3] << 24);
    uint32_t* o32 = reinterpret_cast<uint32_t*>(o);
    o32[0] = w0;
    o32[1] = w1;
    o32[2] = w2;
  } else {
#pragma unroll  // (constant indices: a run-time j sends r / g / b to scratch memory)
    for (int j = 0; j < 4; j++) {
      if (x0 + j >= d.width) break;
      o[3 * j] = (uint8_t)r[j];
      o[3 * j + 1] = (uint8_t)g[j];
      o[3 * j + 2] = (uint8_t)b[j];
    }
  }
}

__global__ __launch_bounds__(256) void k_upsample_norm(const JpegFrameDesc* __restrict__ descs,
                                                       const uint8_t* __restrict__ planes, size_t plane_stride,
                                                       const float* __restrict__ lut, float* __restrict__ out, int W,
                                                       int H) {
  const int frame = blockIdx.z;
  const JpegFrameDesc& d = descs[frame];
  if (d.width != W || d.height != H) return;  // failed / skipped frame: its descriptor is all zero
  const int y = blockIdx.y;
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (y >= H || x0 >= W) return;
  int r[4], g[4], b[4];
  pixels4(d, planes + (size_t)frame * plane_stride, x0, y, r, g, b);
  const size_t hw = (size_t)W * H;
  float* o = out + (size_t)frame * 3 * hw + (size_t)y * W + x0;
  if (x0 + 4 <= W && (W & 3) == 0) {
    *reinterpret_cast<float4*>(o) = make_float4(lut[r[0]], lut[r[1]], lut[r[2]], lut[r[3]]);
    *reinterpret_cast<float4*>(o + hw) = make_float4(lut[256 + g[0]], lut[256 + g[1]], lut[256 + g[2]], lut[256 + g[3]]);
    *reinterpret_cast<float4*>(o + 2 * hw) =
        make_float4(lut[512 + b[0]], lut[512 + b[1]], lut[512 + b[2]], lut[512 + b[3]]);
  } else {
#pragma unroll  // (constant indices, as above)
    for (int j = 0; j < 4; j++) {
      if (x0 + j >= W) break;
      o[j] = lut[r[j]];
      o[hw + j] = lut[256 + g[j]];
      o[2 * hw + j] = lut[512 + b[j]];
    }
  }
}

// 4:2:0 YCbCr fast path of the same stage (the camera-stream case): 8 output pixels per thread,
// 8-byte luma load, 4-byte chroma loads, column sums shared between the 8 pixels; the edge
// formulas of jdsample.c h2v2_fancy_upsample fall out of clamping the neighbour column
// ((3s + s + 8) >> 4 == (4s + 8) >> 4).  Bit-identical to k_upsample_norm.
__global__ __launch_bounds__(256) void k_upsample_norm_420(const JpegFrameDesc* __restrict__ descs,
                                                           const uint8_t* __restrict__ planes, size_t plane_stride,
                                                           const float* __restrict__ lut, float* __restrict__ out, int W,
                                                           int H) {
  __shared__ float s_lut[768];
  for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = lut[i];
  __syncthreads();
  const int frame = blockIdx.y;
  const JpegFrameDesc& d = descs[frame];
  if (d.width != W || d.height != H) return;  // failed / skipped frame
  const int gw = W >> 3;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= gw * H) return;
  const int y = t / gw, x0 = (t - y * gw) * 8;
  const uint8_t* fp = planes + (size_t)frame * plane_stride;
  const int ypitch = d.wblk[0] * 8, cpitch = d.wblk[1] * 8;
  const int dw = d.dw[1], dh = d.dh[1];
  const uint2 yy = *reinterpret_cast<const uint2*>(fp + d.plane_off[0] + (size_t)y * ypitch + x0);
  const int iy = y >> 1, ny = max(0, min(dh - 1, (y & 1) ? iy + 1 : iy - 1));
  const int c0 = x0 >> 1, cl = max(c0 - 1, 0), cr = min(c0 + 4, dw - 1);
  int s[2][6];  // column sums 3*near + far for chroma columns c0-1 .. c0+4 (clamped)
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const uint8_t* p0 = fp + d.plane_off[1 + c] + (size_t)iy * cpitch;
    const uint8_t* p1 = fp + d.plane_off[1 + c] + (size_t)ny * cpitch;
    const uint32_t a = *reinterpret_cast<const uint32_t*>(p0 + c0), b = *reinterpret_cast<const uint32_t*>(p1 + c0);
    s[c][0] = 3 * p0[cl] + p1[cl];
    s[c][5] = 3 * p0[cr] + p1[cr];
#pragma unroll
    for (int i = 0; i < 4; i++) s[c][1 + i] = 3 * (int)((a >> (8 * i)) & 255) + (int)((b >> (8 * i)) & 255);
  }
  float r[8], g[8], bl[8];
  const int bias = 0;
  (void)bias;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int i = 1 + (j >> 1);  // chroma column of this pixel inside s[]
    int cbv, crv;
    if (j & 1) {
      cbv = (s[0][i] * 3 + s[0][i + 1] + 7) >> 4;
      crv = (s[1][i] * 3 + s[1][i + 1] + 7) >> 4;
    } else {
      cbv = (s[0][i] * 3 + s[0][i - 1] + 8) >> 4;
      crv = (s[1][i] * 3 + s[1][i - 1] + 8) >> 4;
    }
    const int yv = (int)(((j < 4 ? yy.x : yy.y) >> (8 * (j & 3))) & 255);
    const int cb = cbv - 128, crr = crv - 128;
    const int rr = clamp255(yv + ((91881 * crr + 32768) >> 16));
    const int gg = clamp255(yv + ((-22554 * cb + 32768 - 46802 * crr) >> 16));
    const int bb = clamp255(yv + ((116130 * cb + 32768) >> 16));
    r[j] = s_lut[rr], g[j] = s_lut[256 + gg], bl[j] = s_lut[512 + bb];
  }
  const size_t hw = (size_t)W * H;
  float* o = out + (size_t)frame * 3 * hw + (size_t)y * W + x0;
  *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
  *reinterpret_cast<float4*>(o + 4) = make_float4(r[4], r[5], r[6], r[7]);
  *reinterpret_cast<float4*>(o + hw) = make_float4(g[0], g[1], g[2], g[3]);
  *reinterpret_cast<float4*>(o + hw + 4) = make_float4(g[4], g[5], g[6], g[7]);
  *reinterpret_cast<float4*>(o + 2 * hw) = make_float4(bl[0], bl[1], bl[2], bl[3]);
  *reinterpret_cast<float4*>(o + 2 * hw + 4) = make_float4(bl[4], bl[5], bl[6], bl[7]);
}

// The same 4:2:0 fast path with interleaved RGB8 output (frames of any size whose width is a multiple of 8): what the
// resize stage reads for frames that are not at the model size, and what N1 draws on and re-encodes.
// Bit-identical to k_upsample_rgb.
__global__ __launch_bounds__(256) void k_upsample_rgb_420(const JpegFrameDesc* __restrict__ descs,
                                                          const uint8_t* __restrict__ planes, size_t plane_stride,
                                                          uint8_t* __restrict__ rgb, size_t rgb_stride) {
  const int frame = blockIdx.y;
  const JpegFrameDesc& d = descs[frame];
  const int W = d.width, H = d.height;
  if (W <= 0) return;  // failed / skipped frame
  const int gw = W >> 3;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= gw * H) return;
  const int y = t / gw, x0 = (t - y * gw) * 8;
  const uint8_t* fp = planes + (size_t)frame * plane_stride;
  const int ypitch = d.wblk[0] * 8, cpitch = d.wblk[1] * 8;
  const int dw = d.dw[1], dh = d.dh[1];
  const uint2 yy = *reinterpret_cast<const uint2*>(fp + d.plane_off[0] + (size_t)y * ypitch + x0);
  const int iy = y >> 1, ny = max(0, min(dh - 1, (y & 1) ? iy + 1 : iy - 1));
  const int c0 = x0 >> 1, cl = max(c0 - 1, 0), cr = min(c0 + 4, dw - 1);
  int s[2][6];  // column sums 3*near + far for chroma columns c0-1 .. c0+4 (clamped)
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const uint8_t* p0 = fp + d.plane_off[1 + c] + (size_t)iy * cpitch;
    const uint8_t* p1 = fp + d.plane_off[1 + c] + (size_t)ny * cpitch;
    const uint32_t a = *reinterpret_cast<const uint32_t*>(p0 + c0), b = *reinterpret_cast<const uint32_t*>(p1 + c0);
    s[c][0] = 3 * p0[cl] + p1[cl];
    s[c][5] = 3 * p0[cr] + p1[cr];
#pragma unroll
    for (int i = 0; i < 4; i++) s[c][1 + i] = 3 * (int)((a >> (8 * i)) & 255) + (int)((b >> (8 * i)) & 255);
  }
  uint32_t px[24];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int i = 1 + (j >> 1);  // chroma column of this pixel inside s[]
    int cbv, crv;
    if (j & 1) {
      cbv = (s[0][i] * 3 + s[0][i + 1] + 7) >> 4;
      crv = (s[1][i] * 3 + s[1][i + 1] + 7) >> 4;
    } else {
      cbv = (s[0][i] * 3 + s[0][i - 1] + 8) >> 4;
      crv = (s[1][i] * 3 + s[1][i - 1] + 8) >> 4;
    }
    const int yv = (int)(((j < 4 ? yy.x : yy.y) >> (8 * (j & 3))) & 255);
    const int cb = cbv - 128, crr = crv - 128;
    px[3 * j] = (uint32_t)clamp255(yv + ((91881 * crr + 32768) >> 16));
    px[3 * j + 1] = (uint32_t)clamp255(yv + ((-22554 * cb + 32768 - 46802 * crr) >> 16));
    px[3 * j + 2] = (uint32_t)clamp255(yv + ((116130 * cb + 32768) >> 16));
  }
  // 24 bytes at a 4-byte aligned address (W % 8 == 0, rgb_stride % 4 == 0: launcher-checked)
  uint32_t* o = reinterpret_cast<uint32_t*>(rgb + (size_t)frame * rgb_stride + ((size_t)y * W + x0) * 3);
#pragma unroll
  for (int q = 0; q < 6; q++) o[q] = px[4 * q] | (px[4 * q + 1] << 8) | (px[4 * q + 2] << 16) | (px[4 * q + 3] << 24);
}

}  // namespace

void launch_upsample_rgb_420(const JpegFrameDesc* d_descs, const uint8_t* d_planes, size_t plane_stride, uint8_t* d_rgb,
                             size_t rgb_stride, uint32_t max_w, uint32_t max_h, uint32_t count, hipStream_t s) {
  if (!count) return;
  dim3 grid(((max_w / 8) * max_h + 255) / 256, count);
  ufd_launch(k_upsample_rgb_420, grid, dim3(256), 0, s, d_descs, d_planes, plane_stride, d_rgb, rgb_stride);
}

void launch_idct(const JpegFrameDesc* d_descs, const int16_t* d_coef, size_t coef_stride, uint8_t* d_planes,
                 size_t plane_stride, uint32_t max_blocks, uint32_t count, bool zigzag, hipStream_t s, const int16_t* d_dc,
                 size_t dc_stride) {
  if (!count || !max_blocks) return;
  dim3 grid((max_blocks + kBlocksPerWG - 1) / kBlocksPerWG, count);
  if (zigzag)
    ufd_launch(k_idct<true>, grid, dim3(256), 0, s, d_descs, d_coef, coef_stride, d_planes, plane_stride, d_dc, dc_stride);
  else
    ufd_launch(k_idct<false>, grid, dim3(256), 0, s, d_descs, d_coef, coef_stride, d_planes, plane_stride, d_dc, dc_stride);
}

void launch_upsample_rgb(const JpegFrameDesc* d_descs, const uint8_t* d_planes, size_t plane_stride, uint8_t* d_rgb,
                         size_t rgb_stride, uint32_t max_w, uint32_t max_h, uint32_t count, hipStream_t s) {
  if (!count) return;
  dim3 grid((max_w + 1023) / 1024, max_h, count);
  ufd_launch(k_upsample_rgb, grid, dim3(256), 0, s, d_descs, d_planes, plane_stride, d_rgb, rgb_stride);
}

void launch_upsample_norm_420(const JpegFrameDesc* d_descs, const uint8_t* d_planes, size_t plane_stride,
                              const float* d_norm_lut, float* d_out, uint32_t W, uint32_t H, uint32_t count,
                              hipStream_t s) {
  if (!count) return;
  dim3 grid(((W / 8) * H + 255) / 256, count);
  ufd_launch(k_upsample_norm_420, grid, dim3(256), 0, s, d_descs, d_planes, plane_stride, d_norm_lut, d_out,
                     (int)W, (int)H);
}

void launch_upsample_norm(const JpegFrameDesc* d_descs, const uint8_t* d_planes, size_t plane_stride,
                          const float* d_norm_lut, float* d_out, uint32_t W, uint32_t H, uint32_t count,
                          hipStream_t s) {
  if (!count) return;
  dim3 grid((W + 1023) / 1024, H, count);
  ufd_launch(k_upsample_norm, grid, dim3(256), 0, s, d_descs, d_planes, plane_stride, d_norm_lut, d_out,
                     (int)W, (int)H);
}

}  // namespace ufd
