// preproc_kernels.hip -- rows A2-A4: UltrafaceModel::preproc (infer_server/src/nn.rs:70-94):
//   image::imageops::resize(input, W, H, FilterType::Triangle)   -- image 0.24.5 sample.rs
//   (v as f32 / 255.0 - mean[c]) / std[c]  -> NCHW f32
// fused into one pass.  The two-pass structure of the crate (vertical pass into an unrounded
// f32 image, then horizontal pass, clamp, round-half-away, u8) is kept per output pixel so the
// u8 value is bit-identical: f32 multiply and add are never contracted (__fmul_rn/__fadd_rn),
// window weights come from the host (IEEE, same formulas), and the normalisation is a 3x256
// table computed on the host with IEEE division.  HBM-bound byte work:
//   k_resize_norm_tiled  a workgroup owns 256 output pixels of one output row: the source rows of its vertical window
//                        are staged in LDS with coalesced 4-byte loads, the vertical pass runs ONCE per source column
//                        into an f32 row in LDS (what the crate's intermediate image holds), the horizontal pass reads
//                        that row; writes are coalesced f32 rows.  Same operation order per value as the crate.
//   k_resize_norm        the per-output-pixel form (every output pixel gathers its window byte by byte and repeats
//                        the vertical pass of the columns it shares with its neighbours): windows too large for LDS.
#include "kernels.hpp"

#include <algorithm>
#include <cmath>

namespace ufd {
namespace {

__global__ __launch_bounds__(256) void k_resize_norm(const uint8_t* __restrict__ src, int sw, int sh, int pitch,
                                                     size_t src_stride, ResizeTaps vt, ResizeTaps ht,
                                                     const float* __restrict__ lut, float* __restrict__ out, int dw,
                                                     int dh) {
  const int frame = blockIdx.z;
  const int oy = blockIdx.y;
  const int ox = blockIdx.x * 256 + threadIdx.x;
  if (ox >= dw) return;
  const uint8_t* img = src + (size_t)frame * src_stride;
  const int vl = vt.left[oy], vn = vt.cnt[oy];
  const float* vw = vt.w + (size_t)oy * vt.stride;
  const int hl = ht.left[ox], hn = ht.cnt[ox];
  const float* hw = ht.w + (size_t)ox * ht.stride;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
  for (int j = 0; j < hn; j++) {
    const int sx = hl + j;
    // vertical pass value of source column sx (sample.rs vertical_sample): t += px * w, in row order
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
    for (int i = 0; i < vn; i++) {
      const uint8_t* p = img + (size_t)(vl + i) * pitch + (size_t)sx * 3;
      const float w = vw[i];
      t0 = __fadd_rn(t0, __fmul_rn((float)p[0], w));
      t1 = __fadd_rn(t1, __fmul_rn((float)p[1], w));
      t2 = __fadd_rn(t2, __fmul_rn((float)p[2], w));
    }
    const float w = hw[j];
    a0 = __fadd_rn(a0, __fmul_rn(t0, w));
    a1 = __fadd_rn(a1, __fmul_rn(t1, w));
    a2 = __fadd_rn(a2, __fmul_rn(t2, w));
  }
  // clamp(t, 0, 255) then f32::round (half away from zero), cast to u8
  const int v0 = (int)roundf(fminf(fmaxf(a0, 0.0f), 255.0f));
  const int v1 = (int)roundf(fminf(fmaxf(a1, 0.0f), 255.0f));
  const int v2 = (int)roundf(fminf(fmaxf(a2, 0.0f), 255.0f));
  const size_t hwp = (size_t)dw * dh;
  float* o = out + (size_t)frame * 3 * hwp + (size_t)oy * dw + ox;
  o[0] = lut[v0];
  o[hwp] = lut[256 + v1];
  o[2 * hwp] = lut[512 + v2];
}

constexpr int kTileMaxRows = 16;   // source rows of a vertical window the tiled kernel takes
constexpr int kTileMaxCols = 2048; // source columns under one tile

// TW = output pixels per workgroup = threads.  Exact f32 arithmetic on bytes is instruction work, not bandwidth: ~270
// instructions per output pixel (byte reads from LDS, conversions, separate multiplies and adds: nothing may be fused),
// 1.1 TB/s of algorithmic traffic at 1280x720 -> 640x480; the per-pixel form above took 100 us for the same batch, this 87.
template <int TW>
__global__ __launch_bounds__(TW) void k_resize_norm_tiled(const uint8_t* __restrict__ src, int sw, int sh, int pitch,
                                                           size_t src_stride, ResizeTaps vt, ResizeTaps ht,
                                                           const float* __restrict__ lut, float* __restrict__ out, int dw,
                                                           int dh, int row_words) {
  extern __shared__ uint32_t s_raw[];  // [vn][row_words] source bytes as loaded (aligned words) | [ncols * 3] f32 vertical sums
  __shared__ float s_vw[kTileMaxRows];
  __shared__ int s_mis[kTileMaxRows];
  const int frame = blockIdx.z, oy = blockIdx.y, tid = threadIdx.x;
  const int ox0 = blockIdx.x * TW, ox1 = min(ox0 + TW, dw) - 1;
  const uint8_t* img = src + (size_t)frame * src_stride;
  const int vl = vt.left[oy], vn = vt.cnt[oy];
  const int c_lo = ht.left[ox0], c_hi = ht.left[ox1] + ht.cnt[ox1];  // source columns [c_lo, c_hi): windows move right with ox
  const int nb = (c_hi - c_lo) * 3;                                   // bytes per source row under the tile
  if (tid < vn) s_vw[tid] = vt.w[(size_t)oy * vt.stride + tid];
  float* s_t = reinterpret_cast<float*>(s_raw + (size_t)vn * row_words);
  // ---- stage: per source row the aligned words that cover its bytes (the row's misalignment is kept in LDS)
  for (int i = 0; i < vn; i++) {
    const uint8_t* rp = img + (size_t)(vl + i) * pitch + (size_t)c_lo * 3;
    const int mis = (int)(reinterpret_cast<uintptr_t>(rp) & 3);
    if (tid == 0) s_mis[i] = mis;
    const uint32_t* wp = reinterpret_cast<const uint32_t*>(rp - mis);
    const int nw = (mis + nb + 3) >> 2;
    for (int w = tid; w < nw; w += TW) s_raw[(size_t)i * row_words + w] = wp[w];
  }
  __syncthreads();
  // ---- vertical pass, once per (source column, channel): t += px * w in row order (sample.rs vertical_sample)
  const uint8_t* s_bytes = reinterpret_cast<const uint8_t*>(s_raw);
  for (int idx = tid; idx < nb; idx += TW) {
    float t = 0.0f;
    for (int i = 0; i < vn; i++) t = __fadd_rn(t, __fmul_rn((float)s_bytes[(size_t)i * row_words * 4 + s_mis[i] + idx], s_vw[i]));
    s_t[idx] = t;
  }
  __syncthreads();
  // ---- horizontal pass per output pixel (horizontal_sample), clamp, round half away, normalise
  const int ox = ox0 + tid;
  if (ox >= dw) return;
  const int hl = ht.left[ox], hn = ht.cnt[ox];
  const float* hw = ht.w + (size_t)ox * ht.stride;
  const float* tp = s_t + (hl - c_lo) * 3;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
  for (int j = 0; j < hn; j++) {
    const float w = hw[j];
    a0 = __fadd_rn(a0, __fmul_rn(tp[3 * j], w));
    a1 = __fadd_rn(a1, __fmul_rn(tp[3 * j + 1], w));
    a2 = __fadd_rn(a2, __fmul_rn(tp[3 * j + 2], w));
  }
  const int v0 = (int)roundf(fminf(fmaxf(a0, 0.0f), 255.0f));
  const int v1 = (int)roundf(fminf(fmaxf(a1, 0.0f), 255.0f));
  const int v2 = (int)roundf(fminf(fmaxf(a2, 0.0f), 255.0f));
  const size_t hwp = (size_t)dw * dh;
  float* o = out + (size_t)frame * 3 * hwp + (size_t)oy * dw + ox;
  o[0] = lut[v0];
  o[hwp] = lut[256 + v1];
  o[2 * hwp] = lut[512 + v2];
}

__global__ __launch_bounds__(256) void k_norm_only(const uint8_t* __restrict__ src, int w, int h, int pitch,
                                                   size_t src_stride, const float* __restrict__ lut,
                                                   float* __restrict__ out) {
  const int frame = blockIdx.z;
  const int y = blockIdx.y;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= w) return;
  const uint8_t* p = src + (size_t)frame * src_stride + (size_t)y * pitch + (size_t)x * 3;
  const size_t hwp = (size_t)w * h;
  float* o = out + (size_t)frame * 3 * hwp + (size_t)y * w + x;
  o[0] = lut[p[0]];
  o[hwp] = lut[256 + p[1]];
  o[2 * hwp] = lut[512 + p[2]];
}

}  // namespace

void launch_resize_norm(const uint8_t* d_src, uint32_t sw, uint32_t sh, uint32_t pitch, size_t src_stride,
                        ResizeTaps vert, ResizeTaps horz, const float* d_norm_lut, float* d_out, uint32_t dw,
                        uint32_t dh, uint32_t count, hipStream_t s) {
  if (!count) return;
  // The tiled kernel when the windows fit its LDS budget.  Window sizes from the axis ratios (the tap tables are on the
  // device): the resampler takes ceil(2 * max(S/D, 1)) + 1 taps at most, a tile of TW outputs spans TW * S/D + that many columns.
  constexpr int tw = 128;  // (64 / 128 / 256 outputs per workgroup: 88.8 / 87.2 / 91.8 us for 1280x720 -> 640x480 at batch 16)
  const double ry = std::max((double)sh / dh, 1.0), rx = std::max((double)sw / dw, 1.0);
  const int max_rows = (int)std::ceil(2.0 * ry) + 2;
  const int max_cols = (int)std::ceil(tw * (double)sw / dw + 2.0 * rx) + 4;
  const int row_words = (max_cols * 3 + 3 + 3) / 4 + 1;
  const size_t lds = ((size_t)max_rows * row_words + (size_t)max_cols * 3) * 4;
  if (max_rows <= kTileMaxRows && max_cols <= kTileMaxCols && vert.stride <= kTileMaxRows && lds <= 60 * 1024) {
    dim3 grid((dw + tw - 1) / tw, dh, count);
    hipLaunchKernelGGL(k_resize_norm_tiled<tw>, grid, dim3(tw), lds, s, d_src, (int)sw, (int)sh, (int)pitch, src_stride, vert, horz,
                       d_norm_lut, d_out, (int)dw, (int)dh, row_words);
    return;
  }
  dim3 grid((dw + 255) / 256, dh, count);
  hipLaunchKernelGGL(k_resize_norm, grid, dim3(256), 0, s, d_src, (int)sw, (int)sh, (int)pitch, src_stride, vert, horz,
                     d_norm_lut, d_out, (int)dw, (int)dh);
}

void launch_norm_only(const uint8_t* d_src, uint32_t w, uint32_t h, uint32_t pitch, size_t src_stride,
                      const float* d_norm_lut, float* d_out, uint32_t count, hipStream_t s) {
  if (!count) return;
  dim3 grid((w + 255) / 256, h, count);
  hipLaunchKernelGGL(k_norm_only, grid, dim3(256), 0, s, d_src, (int)w, (int)h, (int)pitch, src_stride, d_norm_lut,
                     d_out);
}

}  // namespace ufd
