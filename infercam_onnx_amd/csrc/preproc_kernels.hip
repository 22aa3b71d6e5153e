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
constexpr int kTileRows = 4;       // output rows a workgroup walks (tap data of its columns, the table in LDS: loaded once)

// TW = output pixels per row and workgroup = threads.  Exact f32 arithmetic on bytes is instruction work as much as
// traffic (nothing may be fused: separate multiplies and adds, in the crate's order), so the passes are laid out for few
// instructions per value:
//   stage      the source rows of the output row's vertical window, as WORDS: each lane loads two neighbouring aligned
//              words and funnel-shifts them by the row's misalignment (v_alignbyte), so that every row sits in LDS from
//              byte 0 of the tile whatever its address -- coalesced 4-byte loads, no byte traffic;
//   vertical   one lane per 4 source bytes: per row ONE LDS word, four v_cvt_f32_ubyteN, four multiply-add pairs; the
//              four sums (the crate's unrounded f32 intermediate) go back to LDS as one 16-byte store;
//   horizontal one lane per output pixel: taps x 3 channels from that row, clamp, round half away, table, three
//              coalesced f32 stores.
template <int TW>
__global__ __launch_bounds__(TW) void k_resize_norm_tiled(const uint8_t* __restrict__ src, int sw, int sh, int pitch,
                                                          size_t src_stride, ResizeTaps vt, ResizeTaps ht,
                                                          const float* __restrict__ lut, float* __restrict__ out, int dw,
                                                          int dh, int row_words, int max_rows) {
  extern __shared__ uint32_t s_raw[];  // [max_rows][row_words] source bytes, tile-aligned | [4 * row_words] f32 vertical sums
  __shared__ float s_lut[768];
  const int frame = blockIdx.z, tid = threadIdx.x;
  const int ox0 = blockIdx.x * TW, ox1 = min(ox0 + TW, dw) - 1;
  const uint8_t* img = src + (size_t)frame * src_stride;
  const uint8_t* img_last = img + (size_t)sh * pitch - 1;  // last byte of the frame
  const int c_lo = ht.left[ox0], c_hi = ht.left[ox1] + ht.cnt[ox1];  // source columns [c_lo, c_hi): windows move right with ox
  const int nb = (c_hi - c_lo) * 3;                                   // bytes per source row under the tile
  const int nq = (nb + 3) >> 2;                                       // ... in words
  float* const s_t = reinterpret_cast<float*>(s_raw + (size_t)max_rows * row_words);  // (a fixed place: the row count varies between output rows)
  for (int i = tid; i < 768; i += TW) s_lut[i] = lut[i];
  // this lane's output column: window and weights (the same for every row)
  const int ox = ox0 + tid;
  const bool has_px = ox < dw;
  const int hl = has_px ? ht.left[ox] : c_lo, hn = has_px ? ht.cnt[ox] : 0;
  const float* hw = ht.w + (size_t)(has_px ? ox : ox0) * ht.stride;
  const size_t hwp = (size_t)dw * dh;
  const int oy0 = blockIdx.y * kTileRows;
  for (int r = 0; r < kTileRows; r++) {
    const int oy = oy0 + r;
    if (oy >= dh) break;  // (uniform)
    const int vl = vt.left[oy], vn = vt.cnt[oy];
    const float* vw = vt.w + (size_t)oy * vt.stride;
    // ---- stage
    for (int i = 0; i < vn; i++) {
      const uint8_t* rp = img + (size_t)(vl + i) * pitch + (size_t)c_lo * 3;
      const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(rp) & 3);
      const uint32_t* wp = reinterpret_cast<const uint32_t*>(rp - mis);
      // (never read past the word that holds the frame's last byte: garbage there is never used, a fault would be)
      const int wmax = (int)((reinterpret_cast<uintptr_t>(img_last) & ~(uintptr_t)3) - reinterpret_cast<uintptr_t>(wp)) >> 2;
      for (int w = tid; w < nq; w += TW) {
        const uint32_t a = wp[min(w, wmax)], b2 = wp[min(w + 1, wmax)];
        s_raw[(size_t)i * row_words + w] = __builtin_amdgcn_alignbyte(b2, a, mis);
      }
    }
    __syncthreads();
    // ---- vertical pass: t += px * w in row order (sample.rs vertical_sample), four byte positions per lane
    for (int q = tid; q < nq; q += TW) {
      float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
      for (int i = 0; i < vn; i++) {
        const uint32_t u = s_raw[(size_t)i * row_words + q];
        const float w = vw[i];
        t0 = __fadd_rn(t0, __fmul_rn((float)(u & 0xFFu), w));
        t1 = __fadd_rn(t1, __fmul_rn((float)((u >> 8) & 0xFFu), w));
        t2 = __fadd_rn(t2, __fmul_rn((float)((u >> 16) & 0xFFu), w));
        t3 = __fadd_rn(t3, __fmul_rn((float)(u >> 24), w));
      }
      *reinterpret_cast<float4*>(s_t + 4 * q) = make_float4(t0, t1, t2, t3);
    }
    __syncthreads();
    // ---- horizontal pass (horizontal_sample), clamp, round half away, normalise
    if (has_px) {
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
      float* o = out + (size_t)frame * 3 * hwp + (size_t)oy * dw + ox;
      o[0] = s_lut[v0];
      o[hwp] = s_lut[256 + v1];
      o[2 * hwp] = s_lut[512 + v2];
    }
    // (the next row's staging writes s_raw, which this row's vertical pass has finished reading: barrier 2 above; its
    // vertical pass writes s_t behind barrier 1 of the next iteration, after every lane's horizontal reads)
  }
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
  const int row_words = ((max_cols * 3 + 3) / 4 + 4) & ~3;  // (a multiple of 4 words: the f32 row behind the staged rows stays 16-byte aligned)
  const size_t lds = ((size_t)max_rows * row_words + 4 * (size_t)row_words) * 4;
  if (max_rows <= kTileMaxRows && max_cols <= kTileMaxCols && vert.stride <= kTileMaxRows && lds <= 56 * 1024) {
    dim3 grid((dw + tw - 1) / tw, (dh + kTileRows - 1) / kTileRows, count);
    ufd_launch(k_resize_norm_tiled<tw>, grid, dim3(tw), lds, s, d_src, (int)sw, (int)sh, (int)pitch, src_stride, vert, horz,
                       d_norm_lut, d_out, (int)dw, (int)dh, row_words, max_rows);
    return;
  }
  dim3 grid((dw + 255) / 256, dh, count);
  ufd_launch(k_resize_norm, grid, dim3(256), 0, s, d_src, (int)sw, (int)sh, (int)pitch, src_stride, vert, horz,
                     d_norm_lut, d_out, (int)dw, (int)dh);
}

void launch_norm_only(const uint8_t* d_src, uint32_t w, uint32_t h, uint32_t pitch, size_t src_stride,
                      const float* d_norm_lut, float* d_out, uint32_t count, hipStream_t s) {
  if (!count) return;
  dim3 grid((w + 255) / 256, h, count);
  ufd_launch(k_norm_only, grid, dim3(256), 0, s, d_src, (int)w, (int)h, (int)pitch, src_stride, d_norm_lut,
                     d_out);
}

}  // namespace ufd
