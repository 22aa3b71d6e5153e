// conv_kernels.hip -- row A6: the 52 convolutions of UltraFace-RFB (what tract's SimplePlan::run
// executes for `self.model.run(tvec!(input))`, infer_server/src/nn.rs:181; topology SURVEY 8.1).
// Activations are planar NCHW f32 ([frame][channel][y][x]) so that consecutive lanes hold
// consecutive pixels: depthwise stencils read rows, and the pointwise/implicit-GEMM kernels put
// PIXELS on the MFMA column (lane) dimension and OUTPUT CHANNELS on rows -- every global access
// is a 16-byte-per-lane row segment, no im2col buffer, no transposes.
//
//   k_conv_pointwise_mfma : 1x1 convs (66 % of the MACs) on v_mfma_f32_32x32x2_f32.  Exact fp32:
//                           per output, acc = bias, then fma over input channels in order.
//   k_conv_direct         : reference-order direct convolution for the remaining layer shapes
//                           (stem, depthwise, dilated RFB 3x3, 3x3 heads).
#include "kernels.hpp"

namespace ufd {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// Pointwise 1x1:  out[co][p] = act(bias[co] + sum_ci W[co][ci] * in[ci][p])
// One wave computes a tile of 128 pixels x (CT*32) output channels:
//   B operand (k x 32 pixels): lane l loads float4 in[ci = 2*ks + (l>>5)][p0 + 4*(l&31) .. +3];
//                              component j feeds MFMA j, so lane column (l&31) <-> pixel 4*(l&31)+j
//   A operand (32 couts x k):  pre-packed so that lane l reads W[ct*32 + (l&31)][2*ks + (l>>5)]
//   D (32 couts x 32 pixels):  reg r, lane l -> cout (r&3) + 8*(r>>2) + 4*(l>>5), pixel column l&31
// After the K loop the 4 MFMA results of a register form a float4 of 4 consecutive pixels.
template <int CT>
__global__ __launch_bounds__(256) void k_conv_pointwise_mfma(ConvArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int frame = blockIdx.z;
  const int hw = a.oh * a.ow;
  const int p = (blockIdx.x * 4 + wave) * 128 + 4 * (lane & 31);
  const int ct0 = blockIdx.y * CT;
  const int half = lane >> 5;
  const int ksteps = a.cin >> 1;
  const bool live = p < hw;

  floatx16 acc[CT][4];
#pragma unroll
  for (int ct = 0; ct < CT; ct++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int co = (ct0 + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float b = co < a.cout ? a.bias[co] : 0.0f;
#pragma unroll
      for (int j = 0; j < 4; j++) acc[ct][j][r] = b;
    }
  }
  const float* in = a.in + ((size_t)frame * a.in_ctotal + half) * hw + (live ? p : 0);
  const float* wp = a.w + (size_t)ct0 * ksteps * 64 + lane;
  const size_t in_step = (size_t)2 * hw;
#pragma unroll 4
  for (int ks = 0; ks < ksteps; ks++) {
    float4 b = live ? *reinterpret_cast<const float4*>(in + ks * in_step) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
      const float w = wp[((size_t)ct * ksteps + ks) * 64];
      acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b.x, acc[ct][0], 0, 0, 0);
      acc[ct][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b.y, acc[ct][1], 0, 0, 0);
      acc[ct][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b.z, acc[ct][2], 0, 0, 0);
      acc[ct][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b.w, acc[ct][3], 0, 0, 0);
    }
  }
  if (!live) return;
#pragma unroll
  for (int ct = 0; ct < CT; ct++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int co = (ct0 + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < a.cout) {
        float4 v = make_float4(acc[ct][0][r], acc[ct][1][r], acc[ct][2][r], acc[ct][3][r]);
        if (a.res) {
          const float4 q = *reinterpret_cast<const float4*>(a.res + ((size_t)frame * a.cout + co) * hw + p);
          v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
        }
        if (a.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        *reinterpret_cast<float4*>(a.out + ((size_t)frame * a.out_ctotal + a.out_coff + co) * hw + p) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Direct convolution, one output pixel x COB output channels per thread.  Per output element:
// acc = bias; for ci, ky, kx: acc = fma(w, x, acc) -- out-of-image taps skipped (they add 0).
// Weight addresses are wave-uniform (scalar loads); activation loads are coalesced along x.
template <int COB, bool DW>
__global__ __launch_bounds__(256) void k_conv_direct(ConvArgs a) {
  const int frame = blockIdx.z;
  const int co0 = blockIdx.y * COB;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int ohw = a.oh * a.ow;
  if (pix >= ohw) return;
  const int oy = pix / a.ow, ox = pix - oy * a.ow;
  const int kk = a.k * a.k;
  const int cin_g = DW ? 1 : a.cin;
  float acc[COB];
  int cow[COB];
#pragma unroll
  for (int j = 0; j < COB; j++) {
    cow[j] = min(co0 + j, a.cout - 1);
    acc[j] = a.bias[cow[j]];
  }
  const size_t ihw = (size_t)a.ih * a.iw;
  const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
  for (int ci = 0; ci < cin_g; ci++) {
    const float* ip = a.in + ((size_t)frame * a.in_ctotal + (DW ? co0 : ci)) * ihw;
    for (int ky = 0; ky < a.k; ky++) {
      const int iy = iy0 + ky * a.dil;
      if (iy < 0 || iy >= a.ih) continue;
      for (int kx = 0; kx < a.k; kx++) {
        const int ix = ix0 + kx * a.dil;
        if (ix < 0 || ix >= a.iw) continue;
        const float x = ip[(size_t)iy * a.iw + ix];
#pragma unroll
        for (int j = 0; j < COB; j++) acc[j] = fmaf(a.w[((size_t)cow[j] * cin_g + ci) * kk + ky * a.k + kx], x, acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < COB; j++) {
    const int co = co0 + j;
    if (co < a.cout) {
      float v = acc[j];
      if (a.res) v += a.res[((size_t)frame * a.cout + co) * ohw + pix];
      if (a.relu) v = fmaxf(v, 0.0f);
      a.out[((size_t)frame * a.out_ctotal + a.out_coff + co) * ohw + pix] = v;
    }
  }
}

}  // namespace

size_t pointwise_packed_floats(int cin, int cout) { return (size_t)((cout + 31) / 32) * (cin / 2) * 64; }

void pack_pointwise_weights(const float* w, int cin, int cout, float* packed) {
  const int cts = (cout + 31) / 32, ksteps = cin / 2;
  for (int ct = 0; ct < cts; ct++)
    for (int ks = 0; ks < ksteps; ks++)
      for (int lane = 0; lane < 64; lane++) {
        const int co = ct * 32 + (lane & 31), ci = 2 * ks + (lane >> 5);
        packed[((size_t)ct * ksteps + ks) * 64 + lane] = co < cout ? w[(size_t)co * cin + ci] : 0.0f;
      }
}

void launch_conv_pointwise_mfma(const ConvArgs& a, hipStream_t s) {
  const int hw = a.oh * a.ow;
  const int cts = (a.cout + 31) / 32;
  const unsigned gx = (hw + 511) / 512;
  if (cts % 2 == 0) {
    hipLaunchKernelGGL(k_conv_pointwise_mfma<2>, dim3(gx, cts / 2, a.B), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(k_conv_pointwise_mfma<1>, dim3(gx, cts, a.B), dim3(256), 0, s, a);
  }
}

void launch_conv_direct(const ConvArgs& a, hipStream_t s) {
  const unsigned gx = (a.oh * a.ow + 255) / 256;
  if (a.depthwise) {
    hipLaunchKernelGGL((k_conv_direct<1, true>), dim3(gx, a.cout, a.B), dim3(256), 0, s, a);
  } else if (a.cout >= 16) {
    hipLaunchKernelGGL((k_conv_direct<16, false>), dim3(gx, (a.cout + 15) / 16, a.B), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((k_conv_direct<4, false>), dim3(gx, (a.cout + 3) / 4, a.B), dim3(256), 0, s, a);
  }
}

}  // namespace ufd
