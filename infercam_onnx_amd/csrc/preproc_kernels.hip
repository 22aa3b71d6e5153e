// preproc_kernels.hip -- rows A2-A4: UltrafaceModel::preproc (infer_server/src/nn.rs:70-94):
//   image::imageops::resize(input, W, H, FilterType::Triangle)   -- image 0.24.5 sample.rs
//   (v as f32 / 255.0 - mean[c]) / std[c]  -> NCHW f32
// fused into one pass.  The two-pass structure of the crate (vertical pass into an unrounded
// f32 image, then horizontal pass, clamp, round-half-away, u8) is kept per output pixel so the
// u8 value is bit-identical: f32 multiply and add are never contracted (__fmul_rn/__fadd_rn),
// window weights come from the host (IEEE, same formulas), and the normalisation is a 3x256
// table computed on the host with IEEE division.  HBM-bound byte work: each workgroup covers
// one output row segment so source rows are re-used from L1/L2; writes are coalesced f32 rows.
#include "kernels.hpp"

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
