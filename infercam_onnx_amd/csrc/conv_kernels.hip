// conv_kernels.hip -- row A6: the 52 convolutions of UltraFace-RFB (what tract's SimplePlan::run
// executes for `self.model.run(tvec!(input))`, infer_server/src/nn.rs:181; topology SURVEY 8.1).
//
// Activations are planar NCHW f32 ([frame][channel][y][x]).  All MFMA kernels put PIXELS on the
// matrix column (lane) dimension and OUTPUT CHANNELS on rows, so activation loads/stores are
// row segments of the planes (16 B per lane where the shape allows), there is no im2col buffer
// and no layout transform between layers.  The batch is flattened into the pixel dimension
// (pixel groups are numbered over all frames), so small feature maps still fill whole waves.
//
//   k_pw_mfma        1x1 convs on v_mfma_f32_32x32x2_f32 (128 pixels x 32*CT couts per wave)
//   k_dwpw_mfma      depthwise 3x3 (+bias, ReLU) computed per lane straight into the MFMA B
//                    operand of the following 1x1 conv: the depthwise output never exists in
//                    memory (18 of the network's dw->pw pairs)
//   k_conv3x3_mfma   dense 3x3 (stem, dilated RFB convs, 3x3 heads) as implicit GEMM on
//                    v_mfma_f32_16x16x4_f32: K runs over (channel, tap), 4 taps per instruction
//   k_conv_direct    reference-order VALU fallback for shapes the MFMA kernels do not take
//
// Numerics: fp32 MFMA is an exact k-ordered fma chain, and every kernel keeps the order
// "acc = bias; for ci, ky, kx: acc = fma(w, x, acc)" (zero taps add exactly 0), so results are
// bit-identical to a plain fmaf loop nest in that order.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "kernels.hpp"

#include <type_traits>
#include <vector>

namespace ufd {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// Neighbouring lane's value through the DPP crossbar (one VALU instruction; __shfl_up/down
// compile to ds_bpermute_b32, an LDS-pipe instruction plus an address register).  Lane 0 of
// lane_prev / lane 63 of lane_next get 0 (bound_ctrl: no "old" value to materialise first): those
// lanes are halo providers only.
__device__ __forceinline__ float lane_prev(float x) {  // value of lane - 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
}
// Shift inside the 16-lane DPP rows (CTRL: 0x110 + n = row_shr:n, the value of lane - n; 0x100 + n = row_shl:n, lane + n);
// lanes whose source lies outside their row get 0.
template <int CTRL>
__device__ __forceinline__ float row_shift(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_next(float x) {  // value of lane + 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
}
// ReLU of a value that comes out of an MFMA: fmaxf() costs a canonicalising v_max first (the
// compiler cannot know the accumulator is not a signalling NaN); for non-NaN floats the integer
// maximum with 0 is the same function (negative floats, -0 included, are negative integers).
// a * b + c on the 24-bit multiplier (v_mad_i32_i24, full rate): exact for |a|, |b| < 2^23
// (as an instruction: the compiler turns __mul24 of values it knows to be small back into a generic multiply and then
// selects the quarter-rate v_mul_lo_u32 for some of them)
__device__ __forceinline__ int mad24(int a, int b, int c) {
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float relu_acc(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// ... and capped from above in the same instruction: min(max(x, 0), top) as integers (top = INT_MAX: ReLU; top = 0: always 0).
// (v_med3_i32 by hand: the compiler only forms it for constant bounds)
__device__ __forceinline__ float relu_below(float x, int top) {
  int d;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(d) : "v"(__float_as_int(x)), "v"(top));
  return __int_as_float(d);
}

// ------------------------------------------------------------------------------------------------
// Shared tail of the 32x32x2 kernels.  D layout: reg r, lane l -> cout (r&3) + 8*(r>>2) + 4*(l>>5),
// pixel column l&31; MFMA j of a k-step used component j of the lane's float4, so register r of
// the four accumulators is the float4 of 4 consecutive pixels.
// Everything the 16 * CT stores of a lane share is computed ONCE: tensor bases, channel step and flags pinned in scalar
// registers (left to itself the compiler re-read them from the kernel arguments in front of every store, a scalar-memory
// round trip apiece), the lane's 32-bit byte offset of its first channel (tensors stay below 4 GiB: ufd_create checks),
// and whether the tile is a full one (wave-uniform: no per-store bound test then).  Before, every store carried its own
// 64-bit (frame * channels + co) * hw + pix: a dozen quarter-rate integer multiplies each, 190 per wave.
template <int CT>
__device__ __forceinline__ void store_tiles(const ConvArgs& a, floatx16 (&acc)[CT][4], int ct0, int half, size_t frame,
                                            int pix, int hw) {
  typedef __attribute__((address_space(1))) char gchar;  // (global: a pointer that went through the asm below is generic, flat_store)
  char* out = reinterpret_cast<char*>(a.out);
  const char* res = reinterpret_cast<const char*>(a.res);
  int cout = a.cout, relu = a.relu;
  const int co0 = ct0 * 32 + 4 * half;  // the lane's first output channel
  const uint32_t fr = (uint32_t)frame;
  const uint32_t off = 4u * (__umul24(__umul24(fr, (uint32_t)a.out_ctotal) + (uint32_t)(a.out_coff + co0), (uint32_t)hw) + (uint32_t)pix);
  const uint32_t roff = 4u * (__umul24(__umul24(fr, (uint32_t)cout) + (uint32_t)co0, (uint32_t)hw) + (uint32_t)pix);
  uint32_t step = 4u * (uint32_t)hw;  // bytes between output channels
  uint32_t step5 = 5u * step;         // ... and from channel 4q + 3 to 4q + 8 (the other half's channels lie between)
  asm volatile("" : "+s"(out), "+s"(res), "+s"(cout), "+s"(relu), "+s"(step), "+s"(step5));
  gchar* gout = (gchar*)out;
  const gchar* gres = (const gchar*)res;
  // ReLU without a test: the integer maximum with 0 is ReLU for non-NaN floats (relu_acc), with INT_MIN the identity
  int lowest = relu ? 0 : (int)0x80000000;
  asm volatile("" : "+s"(lowest));
  auto emit = [&](auto guarded) {
    // the offsets walk from channel to channel by one vector add of a scalar each (opaque to the compiler, which otherwise
    // turns base + r * step back into a 64-bit multiply-add per store)
    uint32_t o = off, ro = roff;
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int cr = ct * 32 + (r & 3) + 8 * (r >> 2);
        if (!decltype(guarded)::value || co0 + cr < cout) {
          float4 v = make_float4(acc[ct][0][r], acc[ct][1][r], acc[ct][2][r], acc[ct][3][r]);
          if (res) {  // (no launch of the product plan: the RFB's shortcut is summed as a second input of its 1x1 conv)
            const floatx4 q = *reinterpret_cast<const __attribute__((address_space(1))) floatx4*>(gres + ro);
            v.x += q[0], v.y += q[1], v.z += q[2], v.w += q[3];
          }
          const floatx4 vv = {__int_as_float(max(__float_as_int(v.x), lowest)), __int_as_float(max(__float_as_int(v.y), lowest)),
                              __int_as_float(max(__float_as_int(v.z), lowest)), __int_as_float(max(__float_as_int(v.w), lowest))};
          *reinterpret_cast<__attribute__((address_space(1))) floatx4*>(gout + o) = vv;
        }
        const uint32_t d = (r & 3) == 3 ? step5 : step;
        o += d, ro += d;
        asm volatile("" : "+v"(o), "+v"(ro));
      }
    }
  };
  if ((ct0 + CT) * 32 <= cout) emit(std::false_type{});  // full tile (wave-uniform): no bound test per store
  else emit(std::true_type{});
}

// XCD-aware block order.  Hardware deals consecutive workgroup ids round-robin over the 8 XCDs
// (each with a private L2).  Blocks that read the same activation tile but produce different
// 32-cout tiles get ids 8 apart inside a group of 8*cts ids, so they run on the same XCD close in
// time and the second..cts-th read of the tile hits that XCD's L2.  And every XCD gets a CONTIGUOUS range of pixel tiles
// (round 4): neighbouring tiles of the kernels with a halo (depthwise windows, 3x3 rows) share input rows, and dealt
// round-robin they sat on different XCDs, each fetching the shared rows from memory for itself -- rocprofv3 FETCH_SIZE
// 1.4x (k_dwpw_mfma, 60x80 maps) to 2.1x (k_dwpw_coop, 30x40 maps) and 3.1x (the dilated 3x3 launch) of the input.
// Grids are padded to 8 * ceil(tiles / 8) * cts blocks by the launchers.  Placement only changes speed.
__device__ __forceinline__ bool remap_block(const ConvArgs& a, int bx, int* tile, int* ct) {
  const int per = 8 * a.cts;
  const int grp = bx / per, r = bx - grp * per;
  *ct = r >> 3;
  *tile = (r & 7) * ((a.tiles + 7) >> 3) + grp;
  return *tile < a.tiles;
}
// The same for a grid along x that is a multiple of 8 blocks with no cout tiles: XCD x takes blocks [x * n/8, (x+1) * n/8).
__device__ __forceinline__ int xcd_contiguous(int bx, int grid_x) { return (bx & 7) * (grid_x >> 3) + (bx >> 3); }

// Bias of the block's cout tiles through LDS: the first 32 * NT threads place bias[co] (0 past cout) in the order the
// accumulator rows want it -- [tile][half][16 rows] -- beside the weight tables, in front of the barrier those need anyway;
// a lane then takes its 16 values with four 16-byte LDS reads.  (They used to be 16 conditional global loads per lane,
// each in a branch of its own: 2.5 k of a wave's 56 k cycles on the 60x80 layers.  Unconditional global loads are no
// remedy: hipcc then keeps the 64 accumulators in vector registers through the prologue -- 204 VGPRs for 98, one wave
// per SIMD instead of three.)
constexpr int kBiasLds = 32;  // floats per cout tile
// (in two steps, so that the load can sit in front of the block's table copies and the store behind them: one memory round
// trip for everything a block stages, see copy_tables16)
template <int NT>
__device__ __forceinline__ float bias_for_lds(const ConvArgs& a, int tile0) {
  const int t = threadIdx.x & (32 * NT - 1), r = t & 15;  // (NT is a power of two; threads past 32 * NT repeat a load)
  const int co = (tile0 + (t >> 5)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * ((t >> 4) & 1);
  const float b = a.bias[min(co, a.cout - 1)];
  return co < a.cout ? b : 0.0f;
}
template <int NT>
__device__ __forceinline__ void put_bias_lds(float* s_bias, float b) {
  if (threadIdx.x < 32 * NT) s_bias[threadIdx.x] = b;
}

// Tables global -> LDS in 16-byte pieces, ALL of a thread's loads in flight before its first store: up to U pieces per
// thread and trip, from two concatenated sources (piece i < nA: srcA[i] -> putA(i, v); else srcB[i - nA] -> putB(i - nA, v)).
// Written as `for (i = tid; i < n; i += 256) dst[i] = src[i]` hipcc emits load - s_waitcnt vmcnt(0) - store per trip, and
// a conditional source (`i < nA ? a[i] : b[i - nA]`) a branch with a wait of its own per piece: the prologue of every block
// was 5-11 dependent memory round trips (~1 us each under load) on waves that live 17-30 us.  Loads past the end repeat
// the last piece (no branch around a load: behind one the wait counters are not statically known and every wait is for all).
template <int U, int NTHR = 256, typename PutA, typename PutB>
__device__ __forceinline__ void copy_tables16(const float* __restrict__ srcA, int nA, PutA&& putA, const float* __restrict__ srcB, int nB, PutB&& putB) {
  const float4* a4 = reinterpret_cast<const float4*>(srcA);
  const float4* b4 = reinterpret_cast<const float4*>(srcB);
  const int n = nA + nB;
  auto trip = [&](int i0) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = min(i0 + NTHR * u, n - 1);
      v[u] = *(i < nA ? a4 + i : b4 + (i - nA));
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = i0 + NTHR * u;
      if (i < nA) putA(i, v[u]);
      else if (i < n) putB(i - nA, v[u]);
    }
  };
  // (the first trip -- the only one for most tables -- outside the loop: in front of a loop hipcc waits for every load in
  // flight, the caller's prefetched input windows included, before the loop's own loads are even requested)
  trip((int)threadIdx.x);
  for (int i0 = threadIdx.x + NTHR * U; i0 < n + (int)threadIdx.x; i0 += NTHR * U) trip(i0);  // (block-uniform trip count)
}
template <int U>
__device__ __forceinline__ void copy_table16(float* dst, const float* __restrict__ src, int n4) {
  float4* d4 = reinterpret_cast<float4*>(dst);
  auto put = [&](int i, float4 v) { d4[i] = v; };
  copy_tables16<U>(src, n4, put, src, 0, put);
}
template <int CT>
__device__ __forceinline__ void init_acc(const float* s_bias, floatx16 (&acc)[CT][4], int half) {
#pragma unroll
  for (int ct = 0; ct < CT; ct++) {
    const float4* b4 = reinterpret_cast<const float4*>(s_bias + (ct * 2 + half) * 16);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 b = b4[q];
#pragma unroll
      for (int j = 0; j < 4; j++) acc[ct][j][4 * q] = b.x, acc[ct][j][4 * q + 1] = b.y, acc[ct][j][4 * q + 2] = b.z, acc[ct][j][4 * q + 3] = b.w;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) asm volatile("" : "+a"(acc[ct][j]));  // (the values are in the accumulation registers from here on)
  }
}

// Split-K (SK = 4): the four waves of a block share ONE 128-pixel tile and each takes a quarter of
// the input channels; partial accumulators of waves 1..3 go through LDS and wave 0 adds them in
// fixed order (deterministic) before the epilogue.  Used where a launch has too few wave tiles to
// fill the chip (feature maps <= 30x40), so the serial k-chain per wave is 4x shorter.
// SK = 8 (round 6, a frame or a few at a time: eight waves per tile, the chain an eighth): a tree in fixed order -- wave w adds
// wave w + 4's partial tile (w = 0..3), then w + 2's (w = 0, 1), then wave 0 adds wave 1's -- through four tile buffers (64 KB):
// three barrier pairs instead of seven.  The sum's order differs from SK = 4's (fp32 rounding, as between SK = 1 and 4).
template <int CT, int SK = 4>
__device__ __forceinline__ void splitk_reduce(floatx16 (&acc)[CT][4], float* red, int wave, int lane) {
  if (SK == 8) {
    constexpr int kTile = CT * 4 * 16 * 64;  // floats of one partial tile
#pragma unroll 1
    for (int half = SK / 2; half >= 1; half >>= 1) {
      if (wave >= half && wave < 2 * half) {
        float* dst = red + (wave - half) * kTile;
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
          for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) dst[((ct * 4 + j) * 16 + r) * 64 + lane] = acc[ct][j][r];
      }
      __syncthreads();
      if (wave < half) {
        const float* src = red + wave * kTile;
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
          for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[ct][j][r] += src[((ct * 4 + j) * 16 + r) * 64 + lane];
      }
      __syncthreads();
    }
    return;
  }
  // one partial tile at a time through a single 16 KiB buffer: fixed order w = 1, 2, 3
  // (deterministic), barriers keep the compiler from hoisting 192 LDS reads into registers
#pragma unroll 1
  for (int w = 1; w < 4; w++) {
    if (wave == w) {
#pragma unroll
      for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) red[((ct * 4 + j) * 16 + r) * 64 + lane] = acc[ct][j][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[ct][j][r] += red[((ct * 4 + j) * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Pointwise 1x1:  out[co][p] = act(bias[co] + sum_ci W[co][ci] * in[ci][p])
//   B operand (k x 32 pixel columns): lane l loads float4 in[ci = 2*ks + (l>>5)][4 pixels of group g]
//   A operand (32 couts x k): pre-packed so that lane l reads W[ct*32 + (l&31)][2*ks + (l>>5)]
// Pixel groups (4 consecutive pixels of one plane) are numbered over the whole batch.
// The k-loop keeps D k-steps of activations and weights in flight in a register ring (loads are
// unconditional: dead lanes read a valid dummy address), so a wave does not pay one memory round
// trip per k-step.  Needs ksteps % D == 0.
// (bodies take the conv and the block id along x as arguments: a block of a DUAL launch -- two independent convs of the
// graph in one grid, below -- runs one of them with its own numbering)
template <int CT, int D, int SK>
__device__ __forceinline__ void pw_mfma_body(const ConvArgs& a, int bx) {
  // LDS: weights of this cout tile [ksteps][64] (one vector-memory instruction per k-step is left:
  // the activation load) | split-K reduction buffer
  extern __shared__ float s_mem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int tile, ctile;
  if (!remap_block(a, bx, &tile, &ctile)) return;
  const int ct0 = ctile * CT, half = lane >> 5, ksteps = a.cin >> 1;
  float* s_bias = s_mem;  // [CT][2][16]
  float* s_w = s_mem + CT * kBiasLds;
  float* s_red = s_w + CT * ksteps * 64;
  {
    const float bv = bias_for_lds<CT>(a, ct0);
    copy_table16<8>(s_w, a.w + (size_t)ct0 * ksteps * 64, CT * ksteps * 16);
    put_bias_lds<CT>(s_bias, bv);
  }
  __syncthreads();
  const int hw = a.oh * a.ow, gpf = hw >> 2;  // pixel groups per frame
  // (32-bit index arithmetic: a tensor below 4 GiB has fewer than 2^28 pixel groups; the 64-bit division this used to be
  // is a ~100-instruction routine full of quarter-rate multiplies)
  const int g = (SK == 1 ? (tile * 4 + wave) : tile) * 32 + (lane & 31);
  const bool live = g < a.B * gpf;
  const uint32_t frame32 = live ? (uint32_t)g / (uint32_t)gpf : 0u;
  const size_t frame = frame32;
  const int pix = live ? (g - (int)frame32 * gpf) * 4 : 0;

  floatx16 acc[CT][4];
  init_acc<CT>(s_bias, acc, half);
  const int kper = ksteps / SK, kbeg = (SK == 1) ? 0 : wave * kper, kend = kbeg + kper;
  if (SK > 1 && wave > 0) {
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[ct][j][r] = 0.0f;
  }
  // wave-uniform base pointer + 32-bit per-lane element offsets (tensors are < 2^32 bytes)
  const float* __restrict__ in = a.in;
  const uint32_t in_off = (frame32 * (uint32_t)a.in_ctotal + (uint32_t)half) * (uint32_t)hw + (uint32_t)pix;
  const uint32_t in_step = 2u * (uint32_t)hw;
  // two 1x1 convs over different tensors summed as one (RFB: ConvLinear(cat) + shortcut(x)):
  // k-steps from ksplit on read the second tensor (wave-uniform choice)
  const float* __restrict__ in2 = a.in2;
  const uint32_t in2_off = in2 ? (frame32 * (uint32_t)a.in2_ctotal + (uint32_t)half) * (uint32_t)hw + (uint32_t)pix : 0u;
  const int ksplit = in2 ? a.ksplit : 0x7FFFFFFF;
  // (ONE load from a selected address: as two loads in the arms of a branch every wait of the k-loop was for all loads in
  // flight -- vmcnt(0) per k-step, the D-deep queue waited for its newest entry)
  auto load_b = [&](int ks) -> float4 {
    const float* p = ks >= ksplit ? in2 + (in2_off + (uint32_t)(ks - ksplit) * in_step) : in + (in_off + (uint32_t)ks * in_step);
    return *reinterpret_cast<const float4*>(p);
  };
  float4 bq[D];
#pragma unroll
  for (int d = 0; d < D; d++) {
    bq[d] = load_b(min(kbeg + d, kend - 1));
    // (in THIS order: hipcc requested them last slot first, so entering the loop its first wait was for the newest load --
    // and a loop's waits are those of its worst entry: vmcnt(0) at the top of every iteration)
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int ks = kbeg; ks < kend; ks += D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      const float4 b = bq[d];
#pragma unroll
      for (int ct = 0; ct < CT; ct++) {
        const float w = s_w[(ct * ksteps + ks + d) * 64 + lane];
        acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b.x, acc[ct][0], 0, 0, 0);
        acc[ct][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b.y, acc[ct][1], 0, 0, 0);
        acc[ct][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b.z, acc[ct][2], 0, 0, 0);
        acc[ct][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b.w, acc[ct][3], 0, 0, 0);
      }
      // Refill the slot BEHIND the MFMAs that read it (the load lands in the same registers: requested in front of them it
      // needed other registers and a copy back at the loop's end, which waited for the newest load), and pinned: to save
      // registers hipcc kept the D ADDRESSES ahead instead and issued every load one k-step before its use -- vmcnt(0)
      // per k-step, a queue of depth one.
      const int kn = min(ks + D + d, kend - 1);  // (tail: harmless re-read)
      bq[d] = load_b(kn);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (SK > 1) {
    splitk_reduce<CT>(acc, s_red, wave, lane);
    if (wave > 0) return;
  }
  if (live) store_tiles<CT>(a, acc, ct0, half, frame, pix, hw);
}
template <int CT, int D, int SK>
__global__ __launch_bounds__(256) void k_pw_mfma(ConvArgs3 p3) {
  const ConvArgs a = p3.a[blockIdx.y];  // (a copy: see k_dwpw_mfma)
  pw_mfma_body<CT, D, SK>(a, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Fused depthwise 3x3 (pad 1, stride S, +bias, ReLU) -> pointwise 1x1 (+bias, optional ReLU).
// Lane l owns 4 consecutive output pixels (oy, ox..ox+3) and, per k-step, input channel
// ci = 2*ks + (l>>5): it computes the depthwise result of that channel for its 4 pixels from a
// 3 x (4*S+2) input window (aligned float4 row segments + edge scalars) and feeds it directly as
// the B operand.  Needs ow % 4 == 0 and iw % 4 == 0.  a.w2 = depthwise weights packed [cin][12]
// (9 taps, bias, pad; pack_depthwise_weights); a.w/a.bias = packed pointwise weights and bias.
template <int S>
struct DwWindow {  // raw 3 x 4*S input window of one channel for 4 output pixels
  float4 m0[3];
  float4 m1[3];  // stride 2 only
};

// Each 32-lane half of the wave holds 32 consecutive pixel groups of which the first and the last
// are halo providers only: tiles advance by kDwGroups = 30 groups and overlap by two, so the
// column left / right of a lane's group always comes from the neighbouring lane by shuffle and the
// kernel issues exactly three 16-byte loads per k-step (stride 1) -- measured on MI355X, the
// per-lane scalar gathers for the halo columns cost more than all arithmetic of this kernel.
constexpr int kDwGroups = 30;

// Depthwise weights in LDS: three copies of the packed [cin][12] table (9 taps ky-major, bias, 2 pad).
// Copy 1 has tap row 0 zeroed, copy 2 tap row 2: a lane whose window hangs over the top / bottom
// image border reads its taps from that copy instead of masking nine weights per k-step
// (fma(0, x, t) == t for the finite x loaded from the clamped row address).
__device__ __forceinline__ void put_dw_variants(float4* dst, int n4, int i, float4 v) {  // 16 bytes i of the table
  const int q = i % 3;  // which 16 bytes of the record: taps 0-3 | taps 4-7 | tap 8, bias, pad
  float4 top = v, bot = v;
  if (q == 0) top.x = top.y = top.z = 0.f;
  if (q == 1) bot.z = bot.w = 0.f;
  if (q == 2) bot.x = 0.f;
  dst[i] = v, dst[n4 + i] = top, dst[2 * n4 + i] = bot;
}
__device__ __forceinline__ int dw_variant(bool row0ok, bool row2ok) { return !row0ok ? 1 : (!row2ok ? 2 : 0); }

template <int CT, int S, int D, int SK>
__device__ __forceinline__ void dwpw_mfma_body(const ConvArgs& a, int bx) {
  // LDS: depthwise weights [cin][12] | pointwise weights of this cout tile [ksteps][64] | split-K buffer
  extern __shared__ float s_mem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int tile, ctile;
  if (!remap_block(a, bx, &tile, &ctile)) return;  // whole block, before the barrier
  const int ct0 = ctile * CT, half = lane >> 5, ksteps = a.cin >> 1;
  float* s_bias = s_mem;  // [CT][2][16]
  float* s_dw = s_mem + CT * kBiasLds;
  float* s_w = s_dw + 3 * a.cin * 12;
  float* s_red = s_w + CT * ksteps * 64;
  const int ohw = a.oh * a.ow, gpf = ohw >> 2, gpr = a.ow >> 2;
  const int j32 = lane & 31;
  const int total = a.B * gpf;  // (32-bit index arithmetic, as in k_pw_mfma)
  const int g = (SK == 1 ? (tile * 4 + wave) : tile) * kDwGroups + j32 - 1;
  const bool inrange = g >= 0 && g < total;
  const bool live = inrange && j32 >= 1 && j32 <= kDwGroups;  // lanes that own an output group
  const uint32_t frame32 = inrange ? (uint32_t)g / (uint32_t)gpf : 0u;
  const size_t frame = frame32;
  const int rem = inrange ? g - (int)frame32 * gpf : 0;
  const int oy = rem / gpr, ox = (rem - oy * gpr) * 4;
  const int ihw = a.ih * a.iw;

  const int kper = ksteps / SK, kbeg = (SK == 1) ? 0 : wave * kper, kend = kbeg + kper;

  // input window: rows iy0..iy0+2, columns ix0 .. ix0+4*S-1 (ix0 = ox*S, a multiple of 4).
  // Loads are unconditional from clamped (always valid) addresses and zeroed by select; D k-steps
  // of windows are kept in flight in a register ring.
  const int iy0 = oy * S - 1, ix0 = ox * S;
  uint32_t rowoff[3];  // 32-bit BYTE offsets from the wave-uniform base a.in (< 2^32 bytes, launcher-checked)
  const uint32_t lane_base = (frame32 * (uint32_t)a.in_ctotal + (uint32_t)half) * (uint32_t)ihw;
#pragma unroll
  for (int r = 0; r < 3; r++) rowoff[r] = 4u * (lane_base + (uint32_t)(min(max(iy0 + r, 0), a.ih - 1) * a.iw + ix0));
  // rows outside the image are zero padding: the lane reads its taps from the LDS copy that has
  // that tap row zeroed (the loads come from clamped, valid addresses)
  const float* s_dw_lane = s_dw + (dw_variant(iy0 >= 0, iy0 + 2 < a.ih) * a.cin + half) * 12;
  const bool leftok = ix0 > 0;                        // else: image border, the tap is zero padding
  const bool rightok = (S == 1) && (ix0 + 4 < a.iw);  // stride 2 never needs column ix0+8
  const char* __restrict__ in = reinterpret_cast<const char*>(a.in);
  const size_t chan_step = 8u * (size_t)ihw;  // bytes between the channels of consecutive k-steps

  // (wave-uniform channel base + 32-bit lane offset: the scalar-base form of global_load, no
  // per-load address arithmetic on the vector ALU)
  auto load_window = [&](int ks, DwWindow<S>& win) {
    const char* base = in + (size_t)ks * chan_step;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      win.m0[r] = *reinterpret_cast<const float4*>(base + rowoff[r]);
      if (S == 2) win.m1[r] = *reinterpret_cast<const float4*>(base + rowoff[r] + 16);
    }
  };

  // depthwise result of one k-step (4 pixels of channel 2*ks + half), from a loaded window
  // Rows outside the image are zero padding: their three weights are zeroed instead of the 4 or 8
  // loaded pixels (fma(0, x, t) == t for finite x; the loads come from clamped, valid addresses).
  auto dw_compute = [&](const DwWindow<S>& win, int ks, float (&t)[4]) {
    const float4* wq = reinterpret_cast<const float4*>(s_dw_lane + 2 * ks * 12);
    const float4 q0 = wq[0], q1 = wq[1], q2 = wq[2];
    const float wd[10] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y};
    float t0 = wd[9], t1 = t0, t2 = t0, t3 = t0;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const float w0 = wd[3 * r], w1 = wd[3 * r + 1], w2 = wd[3 * r + 2];
      if (S == 1) {
        const float4 m = win.m0[r];
        const float from_prev = lane_prev(m.w), from_next = lane_next(m.x);
        const float l = leftok ? from_prev : 0.f;
        const float rr = rightok ? from_next : 0.f;
        // per pixel, taps in kx order: x-1, x, x+1
        t0 = fmaf(w0, l, t0), t0 = fmaf(w1, m.x, t0), t0 = fmaf(w2, m.y, t0);
        t1 = fmaf(w0, m.x, t1), t1 = fmaf(w1, m.y, t1), t1 = fmaf(w2, m.z, t1);
        t2 = fmaf(w0, m.y, t2), t2 = fmaf(w1, m.z, t2), t2 = fmaf(w2, m.w, t2);
        t3 = fmaf(w0, m.z, t3), t3 = fmaf(w1, m.w, t3), t3 = fmaf(w2, rr, t3);
      } else {
        const float4 m0 = win.m0[r], m1 = win.m1[r];
        const float from_prev = lane_prev(m1.w);
        const float l = leftok ? from_prev : 0.f;
        // output pixel j reads columns 2j-1, 2j, 2j+1 of the window
        t0 = fmaf(w0, l, t0), t0 = fmaf(w1, m0.x, t0), t0 = fmaf(w2, m0.y, t0);
        t1 = fmaf(w0, m0.y, t1), t1 = fmaf(w1, m0.z, t1), t1 = fmaf(w2, m0.w, t1);
        t2 = fmaf(w0, m0.w, t2), t2 = fmaf(w1, m1.x, t2), t2 = fmaf(w2, m1.y, t2);
        t3 = fmaf(w0, m1.y, t3), t3 = fmaf(w1, m1.z, t3), t3 = fmaf(w2, m1.w, t3);
      }
    }
    t[0] = fmaxf(t0, 0.f), t[1] = fmaxf(t1, 0.f), t[2] = fmaxf(t2, 0.f), t[3] = fmaxf(t3, 0.f);
  };

  // Software pipeline: while the matrix pipe works through the MFMAs of k-step ks, the vector ALU
  // computes the depthwise result of k-step ks+1 (sched_group_barrier pins "1 MFMA, then a slice
  // of VALU work"); input windows are prefetched D steps ahead in a register ring.
  DwWindow<S> ring[D];
#pragma unroll
  for (int d = 0; d < D; d++) load_window(min(kbeg + d, kend - 1), ring[d]);
  // the first windows are in flight while the weights go to LDS
  {  // (a.w2 is pre-packed [cin][12], a.w [cout tile][k-step][64]; both tables and the bias in one memory round trip)
    float4* ddst = reinterpret_cast<float4*>(s_dw);
    float4* wdst = reinterpret_cast<float4*>(s_w);
    const int n4 = a.cin * 3;
    const float bv = bias_for_lds<CT>(a, ct0);
    copy_tables16<3, (SK == 8 ? 512 : 256)>(a.w2, n4, [&](int i, float4 v) { put_dw_variants(ddst, n4, i, v); },
                                            a.w + (size_t)ct0 * ksteps * 64, CT * ksteps * 16, [&](int i, float4 v) { wdst[i] = v; });
    put_bias_lds<CT>(s_bias, bv);
  }
  __syncthreads();
  floatx16 acc[CT][4];
  init_acc<CT>(s_bias, acc, half);
  if (SK > 1 && wave > 0) {
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[ct][j][r] = 0.0f;
  }
  float tcur[4];
  dw_compute(ring[0], kbeg, tcur);
  for (int ks0 = kbeg; ks0 < kend; ks0 += D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      const int ks = ks0 + d;
      // slot d held step ks (already consumed into tcur): refill it with step ks + D
      load_window(min(ks + D, kend - 1), ring[d]);
      float wcur[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ct++) wcur[ct] = s_w[(ct * ksteps + ks) * 64 + lane];
      // next step's depthwise values from slot (d+1)%D, beside this step's MFMAs
      float tnext[4];
      dw_compute(ring[(d + 1) % D], min(ks + 1, kend - 1), tnext);
#pragma unroll
      for (int ct = 0; ct < CT; ct++) {
        acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[ct], tcur[0], acc[ct][0], 0, 0, 0);
        acc[ct][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[ct], tcur[1], acc[ct][1], 0, 0, 0);
        acc[ct][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[ct], tcur[2], acc[ct][2], 0, 0, 0);
        acc[ct][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[ct], tcur[3], acc[ct][3], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4 * CT; i++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);  // then a slice of the VALU work
      }
#pragma unroll
      for (int j = 0; j < 4; j++) tcur[j] = tnext[j];
    }
  }
  if (SK > 1) {
    splitk_reduce<CT, SK>(acc, s_red, wave, lane);
    if (wave > 0) return;
  }
  if (live) store_tiles<CT>(a, acc, ct0, half, frame, oy * a.ow + ox, ohw);
}
template <int CT, int S, int D, int SK>
__global__ __launch_bounds__(256) void k_dwpw_mfma(ConvArgs3 p3) {
  const ConvArgs a = p3.a[blockIdx.y];  // (a copy: every field the body reads is requested in one go at the top)
  dwpw_mfma_body<CT, S, D, SK>(a, (int)blockIdx.x);
}
// Split-K over EIGHT waves (512 threads) for launches of a few dozen blocks -- a frame or a few at a time, where a launch
// lasts as long as one wave's walk of its chain (0.45 us per k-step: issue of 4 MFMAs + the depthwise arithmetic on a
// SIMD the wave has to itself) and most CUs are empty anyway.  One conv per launch (n = 1).
template <int S>
__global__ __launch_bounds__(512) void k_dwpw_mfma_sk8(ConvArgs3 p3) {
  const ConvArgs a = p3.a[0];
  dwpw_mfma_body<1, S, 2, 8>(a, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Fused depthwise 3x3 -> pointwise 1x1 for layers with several 32-cout tiles (cout >= 64).
// In k_dwpw_mfma every cout tile is an independent wave that repeats the depthwise arithmetic and
// the input loads of its pixels: 4x (cout 128) or 8x (cout 256) redundant VALU work and vector
// loads, at about one wave per SIMD, so every load round trip is exposed (measured on the
// 128-channel 30x40 layers: 82 k cycles per wave for 16 k cycles of MFMA and 24 k of VALU).
// Here the four waves of a block are CTW cout tiles of the SAME pixel tile (x 4/CTW pixel tiles).
// The k-loop runs in chunks of 8 k-steps: each wave computes the depthwise result of 8/CTW steps
// of the next chunk and publishes it in LDS (one float4 per lane and k-step); after a barrier all
// waves of the pixel tile read every step's B operand back and run their own cout tile's MFMAs.
// Depthwise arithmetic and loads per wave drop by CTW, and a wave's own loads are a whole chunk
// (~2 k cycles) ahead of their use.  A operand (pointwise weights) straight from global / L2 with
// one chunk of prefetch.  fma order per output as k_dwpw_mfma.  Needs (cin/2) % 8 == 0,
// cts % CTW == 0.
template <int S, int CTW>
__device__ __forceinline__ void dwpw_coop_body(const ConvArgs& a, int bx) {
  constexpr int PT = 4 / CTW;   // pixel tiles per block
  constexpr int CH = 8;         // k-steps per chunk
  constexpr int NS = CH / CTW;  // depthwise steps per wave and chunk
  extern __shared__ float s_mem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pt = wave / CTW, cw = wave - pt * CTW;
  // block -> (pixel-tile group, cout-tile group); groups of one pixel tile are 8 ids apart (same XCD)
  const int cgroups = a.cts / CTW;
  int tgrp, cgrp;
  {
    const int per = 8 * cgroups;
    const int grp = bx / per, r = bx - grp * per;
    cgrp = r >> 3;
    tgrp = (r & 7) * ((a.tiles + 7) >> 3) + grp;  // a contiguous range of pixel tiles per XCD: remap_block
  }
  if (tgrp >= a.tiles) return;  // whole block, before any barrier
  const int half = lane >> 5, j32 = lane & 31, ksteps = a.cin >> 1;
  float* s_bias = s_mem;                                            // [CTW][2][16]
  float* s_dw = s_mem + CTW * kBiasLds;                             // [3][cin][12] (put_dw_variants)
  float4* s_t = reinterpret_cast<float4*>(s_dw + 3 * a.cin * 12);   // [PT][3][CH][64]
  const int ohw = a.oh * a.ow, gpf = ohw >> 2, gpr = a.ow >> 2;
  const int total = a.B * gpf;
  const int g = (tgrp * PT + pt) * kDwGroups + j32 - 1;
  const bool inrange = g >= 0 && g < total;
  const bool live = inrange && j32 >= 1 && j32 <= kDwGroups;
  const uint32_t frame32 = inrange ? (uint32_t)g / (uint32_t)gpf : 0u;
  const size_t frame = frame32;
  const int rem = inrange ? g - (int)frame32 * gpf : 0;
  const int oy = rem / gpr, ox = (rem - oy * gpr) * 4;
  const int ihw = a.ih * a.iw;
  const int ct = cgrp * CTW + cw;

  floatx16 acc[1][4];  // (initialised behind the prologue's barrier, below)

  // (addressing and border handling as in k_dwpw_mfma: byte offsets from a wave-uniform channel
  // base, tap rows over the image border zeroed in the lane's copy of the LDS table)
  const int iy0 = oy * S - 1, ix0 = ox * S;
  uint32_t rowoff[3];
  const uint32_t lane_base = (frame32 * (uint32_t)a.in_ctotal + (uint32_t)half) * (uint32_t)ihw;
#pragma unroll
  for (int r = 0; r < 3; r++) rowoff[r] = 4u * (lane_base + (uint32_t)(min(max(iy0 + r, 0), a.ih - 1) * a.iw + ix0));
  const float* s_dw_lane = s_dw + (dw_variant(iy0 >= 0, iy0 + 2 < a.ih) * a.cin + half) * 12;
  const bool leftok = ix0 > 0;
  const bool rightok = (S == 1) && (ix0 + 4 < a.iw);
  const char* __restrict__ in = reinterpret_cast<const char*>(a.in);
  const size_t chan_step = 8u * (size_t)ihw;

  auto load_window = [&](int ks, DwWindow<S>& win) {
    const char* base = in + (size_t)ks * chan_step;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      win.m0[r] = *reinterpret_cast<const float4*>(base + rowoff[r]);
      if (S == 2) win.m1[r] = *reinterpret_cast<const float4*>(base + rowoff[r] + 16);
    }
  };
  auto dw_compute = [&](const DwWindow<S>& win, int ks) -> float4 {
    const float4* wq = reinterpret_cast<const float4*>(s_dw_lane + 2 * ks * 12);
    const float4 q0 = wq[0], q1 = wq[1], q2 = wq[2];
    const float wd[10] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y};
    float t0 = wd[9], t1 = t0, t2 = t0, t3 = t0;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const float w0 = wd[3 * r], w1 = wd[3 * r + 1], w2 = wd[3 * r + 2];
      if (S == 1) {
        const float4 m = win.m0[r];
        const float from_prev = lane_prev(m.w), from_next = lane_next(m.x);
        const float l = leftok ? from_prev : 0.f;
        const float rr = rightok ? from_next : 0.f;
        t0 = fmaf(w0, l, t0), t0 = fmaf(w1, m.x, t0), t0 = fmaf(w2, m.y, t0);
        t1 = fmaf(w0, m.x, t1), t1 = fmaf(w1, m.y, t1), t1 = fmaf(w2, m.z, t1);
        t2 = fmaf(w0, m.y, t2), t2 = fmaf(w1, m.z, t2), t2 = fmaf(w2, m.w, t2);
        t3 = fmaf(w0, m.z, t3), t3 = fmaf(w1, m.w, t3), t3 = fmaf(w2, rr, t3);
      } else {
        const float4 m0 = win.m0[r], m1 = win.m1[r];
        const float from_prev = lane_prev(m1.w);
        const float l = leftok ? from_prev : 0.f;
        t0 = fmaf(w0, l, t0), t0 = fmaf(w1, m0.x, t0), t0 = fmaf(w2, m0.y, t0);
        t1 = fmaf(w0, m0.y, t1), t1 = fmaf(w1, m0.z, t1), t1 = fmaf(w2, m0.w, t1);
        t2 = fmaf(w0, m0.w, t2), t2 = fmaf(w1, m1.x, t2), t2 = fmaf(w2, m1.y, t2);
        t3 = fmaf(w0, m1.y, t3), t3 = fmaf(w1, m1.z, t3), t3 = fmaf(w2, m1.w, t3);
      }
    }
    return make_float4(fmaxf(t0, 0.f), fmaxf(t1, 0.f), fmaxf(t2, 0.f), fmaxf(t3, 0.f));
  };

  const int nchunks = ksteps / CH;
  float4* my_t = s_t + (size_t)pt * 3 * CH * 64;  // this pixel tile's three chunk buffers
  const float* __restrict__ wsrc = a.w + (size_t)ct * ksteps * 64 + lane;  // A operand of step ks: wsrc[ks * 64]
  // Schedule of iteration c (one barrier each): issue the loads of the weights of chunk c+1 (a whole iteration before their
  // use), compute this wave's depthwise steps of chunk c+2 into LDS buffer (c+2)%3, request the windows of chunk c+3, then
  // multiply chunk c out of buffer c%3 -- which every wave finished writing two iterations ago, so the MFMA phase never
  // waits for a neighbour's depthwise phase of the same iteration.
  auto load_windows = [&](int chunk, DwWindow<S> (&w)[NS]) {
    const int cc = min(chunk, nchunks - 1);  // past the end: harmless re-read
#pragma unroll
    for (int i = 0; i < NS; i++) load_window(cc * CH + cw + i * CTW, w[i]);
  };
  auto load_weights = [&](int chunk, float (&w)[CH]) {
    const int cc = min(chunk, nchunks - 1);
#pragma unroll
    for (int kk = 0; kk < CH; kk++) w[kk] = wsrc[(cc * CH + kk) * 64];
  };
  auto publish = [&](int chunk, const DwWindow<S> (&w)[NS]) {
    if (chunk >= nchunks) return;
    float4* tn = my_t + (size_t)(chunk % 3) * CH * 64;
#pragma unroll
    for (int i = 0; i < NS; i++) tn[(cw + i * CTW) * 64 + lane] = dw_compute(w[i], chunk * CH + cw + i * CTW);
  };
  auto multiply = [&](int chunk, const float (&w)[CH]) {
    const float4* tb = my_t + (size_t)(chunk % 3) * CH * 64;
#pragma unroll
    for (int kk = 0; kk < CH; kk++) {
      const float4 b = tb[kk * 64 + lane];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kk], b.x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kk], b.y, acc[0][1], 0, 0, 0);
      acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kk], b.z, acc[0][2], 0, 0, 0);
      acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kk], b.w, acc[0][3], 0, 0, 0);
    }
  };
  // ONE set of input windows in flight (round 3; two sets before): the windows of chunk c+3 are requested right behind the
  // publication of chunk c+2 and land during the MFMA phase of chunk c and the barrier.  A whole iteration more of run-ahead
  // cost 48 (stride 1) / 96 (stride 2: 3 rows x 8 floats per window) registers and bought nothing: k_dwpw_coop<1, 4> 29.7 ->
  // 29.0 us, the stride-2 dual launches 41.7 -> 37.5 us alone at three waves per SIMD instead of two.
  // Prologue: everything the block needs first -- the windows and the weights of chunk 0, the depthwise table, the bias --
  // is requested in ONE go in front of the first barrier (it used to be bias, barrier, windows + table in a load - wait -
  // store loop, barrier, ...: five to six memory round trips one after the other on waves that live ~29 us; a second set of
  // windows up front as well costs 28-60 registers, a wave per SIMD on the stride-2 instances).
  float wA[CH], wB[CH];
  DwWindow<S> win[NS];
  load_windows(0, win);
  load_weights(0, wA);
  {
    const float bv = bias_for_lds<CTW>(a, cgrp * CTW);
    float4* ddst = reinterpret_cast<float4*>(s_dw);
    const int n4 = a.cin * 3;
    auto put = [&](int i, float4 v) { put_dw_variants(ddst, n4, i, v); };
    copy_tables16<2>(a.w2, n4, put, a.w2, 0, put);
    put_bias_lds<CTW>(s_bias, bv);
  }
  __syncthreads();
  init_acc<1>(s_bias + cw * kBiasLds, acc, half);
  publish(0, win);
  load_windows(1, win);
  publish(1, win);
  load_windows(2, win);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < nchunks; c += 2) {
    // even iteration: weights of chunk c in wA
    load_weights(c + 1, wB);
    publish(c + 2, win);
    load_windows(c + 3, win);
    multiply(c, wA);
    __syncthreads();
    if (c + 1 >= nchunks) break;
    // odd iteration: weights of chunk c+1 in wB
    load_weights(c + 2, wA);
    publish(c + 3, win);
    load_windows(c + 4, win);
    multiply(c + 1, wB);
    __syncthreads();
  }
  if (live) store_tiles<1>(a, acc, ct, half, frame, oy * a.ow + ox, ohw);
}
template <int S, int CTW>
__global__ __launch_bounds__(256) void k_dwpw_coop(ConvArgs3 p3) {
  const ConvArgs a = p3.a[blockIdx.y];  // (a copy: see k_dwpw_mfma)
  dwpw_coop_body<S, CTW>(a, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// DUAL launches: two convolutions of the graph that do not depend on each other -- a cls/reg head pair and the next
// backbone block, which both read the tensor produced just before -- as ONE grid: the first ax * ay blocks run conv A
// (numbered x + ax * y like its own launch), the rest conv B.  The small-map layers are latency-bound at one wave per
// SIMD or less; side by side they fill each other's stalls, the dependent chain of a batch is one launch shorter per
// pair, and no extra HIP stream is needed (streams beyond the runtime's four hardware queues cost more than they
// bring: DESIGN.md, host pipeline).  Results are those of the separate launches bit for bit: the bodies are the same.
struct DualArgs {
  ConvArgs3 a, b;
  int ax, ay;  // grid of conv A
};
template <int SA, int SKA, int SB>
__global__ __launch_bounds__(256) void k_dual_dwpw_coop(DualArgs q) {  // A: k_dwpw_mfma<1, SA, 2, SKA> (merged head pair); B: k_dwpw_coop<SB, 4>
  const int bid = (int)blockIdx.x, na = q.ax * q.ay;
  if (bid < na) {
    const int y = bid / q.ax;
    const ConvArgs a = q.a.a[y];  // (copies: see k_dwpw_mfma)
    dwpw_mfma_body<1, SA, 2, SKA>(a, bid - y * q.ax);
  } else {
    const ConvArgs b = q.b.a[0];
    dwpw_coop_body<SB, 4>(b, bid - na);
  }
}
template <int SA, int SKA, int SKB>
__global__ __launch_bounds__(256) void k_dual_dwpw_pw(DualArgs q) {  // A as above; B: k_pw_mfma<1, 4, SKB> (one 1x1 conv)
  const int bid = (int)blockIdx.x, na = q.ax * q.ay;
  if (bid < na) {
    const int y = bid / q.ax;
    const ConvArgs a = q.a.a[y];  // (copies: see k_dwpw_mfma)
    dwpw_mfma_body<1, SA, 2, SKA>(a, bid - y * q.ax);
  } else {
    const ConvArgs b = q.b.a[0];
    pw_mfma_body<1, 4, SKB>(b, bid - na);
  }
}

// ------------------------------------------------------------------------------------------------
// Two depthwise->pointwise blocks in one launch:
//   [dw 3x3 s1 (C1 ch, +bias, ReLU) -> pw C1->32 (+bias, act)] -> [dw 3x3 s2 (32 ch, +bias, ReLU) -> pw 32->cout<=64]
// The tensor between the blocks (X1: 32 channels at full resolution, 315 MB per 32-frame batch for
// m1 -> m2 of the 640x480 model, written once and read once) never exists in memory.
// A group of 4 output pixels (oy, ox..ox+3) of the second block reads rows 2oy-1..2oy+1 and columns
// 2ox-1..2ox+7 of X1.  Two neighbouring lane columns share an output group: column parity s = 0
// owns X1 columns 2ox..2ox+3 and output pixels 0,1; s = 1 owns X1 columns 2ox+4..2ox+7 and output
// pixels 2,3.  So the lane columns of a wave are 32 consecutive 4-pixel groups of an X1 row, and
// the first block is exactly the stride-1 k_dwpw_mfma inner loop (three 16-byte loads per k-step,
// halo columns by shuffle, 4 MFMAs), run once per X1 row.  The D layout leaves the lane with 16
// of the 32 X1 channels of its 4 pixels (the other 16 sit in lane l^32); the second depthwise
// conv is per channel, so each row is folded straight into the lane's 16 x 2 second-block sums:
//   pixel A: taps (left lane's x3, x0, x1)      pixel B: taps (x1, x2, x3)
// for both parities.  The fma order is the unfused one (bias; rows top to bottom; taps left to
// right).
// ROW ROLLING: a wave owns a BAND of a2.band output rows of its column strip and walks down the
// 2 * band + 1 X1 rows under it.  X1 row 2(oy0 + j) - 1 is the bottom tap row of output row
// oy0 + j - 1 and the top tap row of output row oy0 + j: it is computed once and folded into both
// sums (two live sets), then the finished output row goes through the second pointwise conv.  Every
// X1 row of a band is computed once (band = 1: 3 X1 rows per output row; band = 5: 2.2) and every
// input row read once per band plus a two-row halo.  Lanes are numbered over (frame, band, column
// group), so a wave's columns may straddle bands, rows and frames: a lane's left / right neighbour
// is either the same row's neighbouring group or beyond an image border (zero padding).
// Before the second pointwise conv the halves exchange channels so that k-step s multiplies
// channels (2s, 2s+1) in ascending order like k_dwpw_mfma.
// Tiles advance by 30 lane columns = 15 output groups; columns 0 and 31 only provide halos.
// a[0]: first block (in, w2 = dw [C1][12], w = packed pw, bias, relu); a[1]: second block
// (w2 = dw [32][12], w = packed pw, bias, relu, out...).  Needs a[0].iw % 8 == 0, a[1].ow % 4 == 0,
// a[1].oh % a[1].band == 0.
// ROW RING (RING = true, the 16-channel instance): consecutive X1 rows of a band share two of their three input
// rows.  Each wave keeps those two rows of its strip in LDS for all channels (2 x C1/2 k-steps x 64 lanes x 16 B =
// 16 KB per wave; a lane only ever reads back what it wrote itself, so there is no barrier), and per k-step loads
// only the NEW bottom row from global memory: one 16-byte global load and two LDS reads instead of three global
// loads, and every input row crosses the memory system once per band instead of three times.
// ISSUE DIET (round 5).  On gfx950 fp32 MFMAs and the wave's own vector instructions share one issue path, so this kernel is
// bound by their SUM (rule 1); `tools/isa_mix.py` counted 65 vector instructions per k-step of the first
// block for the 36 multiply-adds and 4 ReLUs its arithmetic needs.  What went:
//   * halo columns: the neighbouring lane's pixel used to be a v_mov_b32_dpp, the image border a v_cndmask on it (12 per
//     k-step).  Now the cross-lane read happens INSIDE the multiply-add (v_fmac_f32_dpp, dw_row below: five of the six), and the
//     border is a per-lane WEIGHT: a lane whose window hangs over the left / right image border reads its first / last tap
//     column from an all-zero copy of the edge-tap table in LDS (fma(0, x, t) == t for the finite x of the neighbouring row);
//   * the bias of the first pointwise conv used to be 64 v_mov into the accumulators per X1 row: now 16 LDS reads;
//   * the exchange of channels between the wave's halves in front of the second pointwise conv was select + ds_bpermute +
//     select (48 v_cndmask per output row): now 16 v_permlane32_swap;
//   * an X1 row outside the image used to switch the second depthwise conv to an all-zero tap table: now the ReLU of the
//     accumulators is a v_med3_i32 against a per-lane upper bound (INT_MAX, or 0 for such a row), same instruction count;
//   * the k-loop is fully unrolled: LDS and channel offsets are immediates, no address arithmetic on the vector ALU.
// The order of every output's multiply-adds is unchanged: results are bit-identical to the unfused pair (tested).
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx2 __attribute__((ext_vector_type(2)));

// One tap row (weights w0 w1 w2) of a stride-1 depthwise 3x3 for the lane's 4 consecutive pixels m.x .. m.w of one channel:
//   t0 += wl * [lane - 1].m.w + w1 * m.x + w2 * m.y        t2 += w0 * m.y + w1 * m.z + w2 * m.w
//   t1 += w0 * m.x + w1 * m.y + w2 * m.z                   t3 += w0 * m.z + w1 * m.w + wr * [lane + 1].m.x
// (each sum in this order: taps left to right).  wl / wr: w0 / w2, or 0 in the lane at the left / right image border.
// FIRST: t = bias + ... (the first tap row of a channel).  Written as the 13 instructions it is: the compiler does not fold
// a DPP move into the multiply-add that uses it.  s_nop 4: a DPP source must not have been written by the vector ALU in the two
// cycles before, and a vector-ALU write of EXEC (v_cmpx) needs FIVE wait states in front of a DPP op -- the pixels come from
// loads and the kernels branch on scalar conditions, but the compiler's hazard recogniser does not look inside inline asm, so
// the asm carries the worst case itself (3 cycles more per 12-13 vector instructions).  wave_shr / wave_shl are GFX9 DPP modes.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "dw_row / wave_inscan use GFX9-only DPP modes (wave_shr, wave_shl, row_bcast): this library is built for gfx950 only"
#endif
template <bool FIRST>
__device__ __forceinline__ void dw_row(float (&t)[4], const float4 m, float w0, float w1, float w2, float wl, float wr, float bias) {
  if (FIRST) {
    float tmp;
    asm("s_nop 4\n\t"
        "v_mov_b32_dpp %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fma_f32 %0, %12, %4, %14\n\t"
        "v_fma_f32 %1, %9, %5, %14\n\t"
        "v_fma_f32 %2, %9, %6, %14\n\t"
        "v_fma_f32 %3, %9, %7, %14\n\t"
        "v_fmac_f32 %0, %10, %5\n\t"
        "v_fmac_f32 %1, %10, %6\n\t"
        "v_fmac_f32 %2, %10, %7\n\t"
        "v_fmac_f32 %3, %10, %8\n\t"
        "v_fmac_f32 %0, %11, %6\n\t"
        "v_fmac_f32 %1, %11, %7\n\t"
        "v_fmac_f32 %2, %11, %8\n\t"
        "v_fmac_f32_dpp %3, %5, %13 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(tmp)
        : "v"(m.x), "v"(m.y), "v"(m.z), "v"(m.w), "v"(w0), "v"(w1), "v"(w2), "v"(wl), "v"(wr), "v"(bias));
  } else {
    asm("s_nop 4\n\t"
        "v_fmac_f32_dpp %0, %7, %11 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32 %1, %8, %4\n\t"
        "v_fmac_f32 %2, %8, %5\n\t"
        "v_fmac_f32 %3, %8, %6\n\t"
        "v_fmac_f32 %0, %9, %4\n\t"
        "v_fmac_f32 %1, %9, %5\n\t"
        "v_fmac_f32 %2, %9, %6\n\t"
        "v_fmac_f32 %3, %9, %7\n\t"
        "v_fmac_f32 %0, %10, %5\n\t"
        "v_fmac_f32 %1, %10, %6\n\t"
        "v_fmac_f32 %2, %10, %7\n\t"
        "v_fmac_f32_dpp %3, %4, %12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])
        : "v"(m.x), "v"(m.y), "v"(m.z), "v"(m.w), "v"(w0), "v"(w1), "v"(w2), "v"(wl), "v"(wr));
  }
}
// The three tap rows of one channel: `wq` -> its 12-float record (9 taps ky-major, bias), `el` / `er` -> the lane's edge
// taps of the channel (first / last tap column of the three rows; the zero copy for a lane at that image border).
__device__ __forceinline__ void dw_taps(const float4 (&m3)[3], const float* wq, const float* el, const float* er, float (&t)[4]) {
  const float4 q0 = reinterpret_cast<const float4*>(wq)[0], q1 = reinterpret_cast<const float4*>(wq)[1];
  const float2 q2 = reinterpret_cast<const float2*>(wq)[4];
  const float4 l = *reinterpret_cast<const float4*>(el), r = *reinterpret_cast<const float4*>(er);
  dw_row<true>(t, m3[0], q0.x, q0.y, q0.z, l.x, r.x, q2.y);
  dw_row<false>(t, m3[1], q0.w, q1.x, q1.y, l.y, r.y, 0.f);
  dw_row<false>(t, m3[2], q1.z, q1.w, q2.x, l.z, r.z, 0.f);
  // (ReLU as the integer maximum with 0: behind inline asm fmaxf() would cost a canonicalising v_max_f32 of its own first)
  t[0] = relu_acc(t[0]), t[1] = relu_acc(t[1]), t[2] = relu_acc(t[2]), t[3] = relu_acc(t[3]);
}
template <int C1, int CT2, bool RING>  // channels of the first block's input (16 / 32), 32-cout tiles of the second block (1 / 2)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_dwpw2_mfma(ConvArgs3 p3) {
  const ConvArgs& a1 = p3.a[0];
  const ConvArgs& a2 = p3.a[1];
  constexpr int KS1 = C1 / 2, KS2 = 16;
  extern __shared__ float s_mem[];
  float* s_dw1 = s_mem;                 // [3][C1][12] (put_dw_variants)
  float* s_e1 = s_dw1 + 3 * C1 * 12;    // [4][C1][8] (fill_edge_taps)
  float* s_w1 = s_e1 + 4 * C1 * 8;      // [KS1][64]
  float* s_f2 = s_w1 + KS1 * 64;        // [2][16 channel pairs (c, c + 1)][32] second depthwise conv: per tap row k eight floats (wk0' wk0'
                                        // | wk0 wk0 | wk1 wk1 | wk2 wk2) of (c, c + 1), then (bias bias 0 ...); wk0' = wk0 in copy 0, 0 in
                                        // copy 1 (the lanes at the left image border read that one)
  float* s_w2 = s_f2 + 2 * 16 * 32;     // [CT2][KS2][64]
  float* s_b1 = s_w2 + CT2 * KS2 * 64;  // [2 halves][16] bias of the first pointwise conv in the order of the accumulator rows
  float* s_b2 = s_b1 + 32;              // [CT2][2][16] bias of the second (0 beyond cout)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // RING: [wave][2 row slots][KS1][64 lanes] float4 behind the tables (a multiple of 16 bytes in)
  float4* const s_ring = reinterpret_cast<float4*>(s_b2 + 32 * CT2) + (size_t)wave * 2 * KS1 * 64 + lane;
  // Workgroup ids are dealt round-robin over the 8 XCDs: give every XCD a contiguous range of
  // tiles, so that neighbouring tiles (which share input rows) meet in the same L2 close in time.
  const int per_xcd = (int)gridDim.x >> 3;  // the grid is a multiple of 8
  const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (tile >= a2.tiles) return;  // whole block, before the barrier
  {
    // The five tables and the two biases in ONE memory round trip: every thread requests its 16-byte pieces of the
    // concatenation (depthwise records | edge taps | first 1x1 | fold records | second 1x1 -- sizes are compile-time) and its
    // bias value, then stores them.  (As five copy loops one after the other, each `load - wait - store` per trip, a block's
    // prologue was a dozen dependent round trips.)  Edge taps and fold records come packed by pack_depthwise_weights.
    constexpr int nD = C1 * 3, nE = 4 * C1 * 2, nW1 = KS1 * 16, nF = 2 * 16 * 8, nW2 = CT2 * KS2 * 16;
    constexpr int o1 = nD, o2 = o1 + nE, o3 = o2 + nW1, o4 = o3 + nF, nAll = o4 + nW2, U = (nAll + 255) / 256;
    const float4* gD = reinterpret_cast<const float4*>(a1.w2);
    const float4* gE = reinterpret_cast<const float4*>(a1.w2 + C1 * 12);
    const float4* gW1 = reinterpret_cast<const float4*>(a1.w);
    const float4* gF = reinterpret_cast<const float4*>(a2.w2 + 32 * 12 + 4 * 32 * 8);
    const float4* gW2 = reinterpret_cast<const float4*>(a2.w);
    const int t = threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = min(t + 256 * u, nAll - 1);
      v[u] = *(i < o1 ? gD + i : i < o2 ? gE + (i - o1) : i < o3 ? gW1 + (i - o2) : i < o4 ? gF + (i - o3) : gW2 + (i - o4));
    }
    // accumulator row r of half h holds channel (r & 3) + 8 * (r >> 2) + 4 * h of its tile
    const int tb = t & (32 * (1 + CT2) - 1 < 64 ? 63 : 127), r = tb & 15, co = (tb >> 5) * 32 + (r & 3) + 8 * (r >> 2) + 4 * ((tb >> 4) & 1);
    const float bias1 = a1.bias[co & 31];
    const float bias2 = a2.bias[min(max(co - 32, 0), a2.cout - 1)];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = t + 256 * u;
      if (i < o1) put_dw_variants(reinterpret_cast<float4*>(s_dw1), nD, i, v[u]);
      else if (i < o2) reinterpret_cast<float4*>(s_e1)[i - o1] = v[u];
      else if (i < o3) reinterpret_cast<float4*>(s_w1)[i - o2] = v[u];
      else if (i < o4) reinterpret_cast<float4*>(s_f2)[i - o3] = v[u];
      else if (i < nAll) reinterpret_cast<float4*>(s_w2)[i - o4] = v[u];
    }
    if (t < 32) s_b1[t] = bias1;
    else if (t < 32 * (1 + CT2)) s_b2[t - 32] = co - 32 < a2.cout ? bias2 : 0.0f;
  }
  __syncthreads();
  const int half = lane >> 5, j32 = lane & 31;
  const int R = a2.band, bands = a2.oh / R;
  const int ohw = a2.oh * a2.ow, gpr = a2.ow >> 2, gpb = bands * gpr;  // column groups per row / per frame of bands
  const long total = 2L * a2.B * gpb;  // X1 half-groups over (frame, band, group)
  const long q = ((long)tile * 4 + wave) * kDwGroups + j32 - 1;
  const bool inrange = q >= 0 && q < total;
  const bool live = inrange && j32 >= 1 && j32 <= kDwGroups;
  const int sub = inrange ? (int)(q & 1) : 0;
  const long g = inrange ? (q >> 1) : 0;
  const size_t frame = (size_t)(g / gpb);
  const int rem = (int)(g - (long)frame * gpb);
  const int band = rem / gpr, ox = (rem - band * gpr) * 4, oy0 = band * R;
  const int H1 = a1.ih, W1 = a1.iw, ihw = H1 * W1;
  const int x0 = 2 * ox + 4 * sub;            // first X1 / input column of the lane
  const bool leftok = x0 > 0;                 // column x0 - 1 exists (else zero padding)
  const bool rightok = x0 + 4 < W1;           // column x0 + 4 exists
  // Input windows come by BUFFER loads: descriptor of the input tensor in scalar registers, the lane's 32-bit byte offset of
  // the row in a vector register, the channel of the k-step as the instruction's scalar offset -- no address arithmetic on the
  // vector ALU at all (global_load with a 64-bit address cost a v_lshl_add_u64 per load).  Tensors stay below 4 GiB (ufd_create).
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a1.in), 0, 0xFFFFFFFF, 0x00020000);
  const uint32_t lane_base = (uint32_t)((frame * a1.in_ctotal + half) * ihw) + (uint32_t)x0;
  const uint32_t chan_step = 8u * (uint32_t)ihw;  // bytes between the channels of consecutive k-steps
  auto load_row = [&](uint32_t off, int ks) -> float4 {
    const uintx4 v = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, (uint32_t)ks * chan_step, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
  };

  // second block: depthwise sums of the lane's 16 channels x 2 output pixels -- of the output row being
  // finished (t2) and, across an X1 row that both need, of the one below it (t2n)
  // Packed over CHANNEL PAIRS: accumulator rows (2p, 2p + 1) of the lane are channels (c, c + 1), c = 2(p & 1) + 8(p >> 1) + 4 half;
  // t2[j][p] = output pixel j (A, B) of both.  Every tap is then one v_pk_fma_f32 whose operands are plain register pairs.
  floatx2 t2[2][8], t2n[2][8];
  // the lane's records of the second depthwise conv: channel pair p at float offset rec2(p)
  const float* const f2 = s_f2 + (leftok ? 0 : 16 * 32) + 2 * half * 32;
  auto rec2 = [](int p) { return ((p & 1) + 4 * (p >> 1)) * 32; };
#pragma unroll
  for (int p = 0; p < 8; p++) t2[0][p] = t2[1][p] = *reinterpret_cast<const floatx2*>(f2 + rec2(p) + 24);

  // input windows: 3 rows x 4 columns of channel 2*ks + half.  Rows that come from memory are requested PD k-steps ahead,
  // ACROSS the X1 rows (the queue never drains: the first windows of the next row are loaded while the current row is
  // folded into the second depthwise conv).  RING: one row per k-step comes from memory (4 registers per step in flight),
  // the two rows from the LDS ring are read two k-steps ahead.  Measured alone on MI355X (round 5, docs/EXPERIMENTS.md):
  // RING PD 2 / 4 / 8: 86.7 / 81.5 / 114 us (8: 20 spilled dwords, scratch traffic inside the k-loop); three rows per
  // k-step, PD 2 / 4: 64.1 / 66.1 us (spills again).
  constexpr int PD = RING ? 4 : 2;
  static_assert(KS1 % PD == 0 && PD >= 2, "the queue slot of a k-step is ks % PD in every X1 row");
  float4 win[PD][RING ? 1 : 3];
  float4 up[2][2];  // RING: top and middle row of k-steps (even, odd) from the LDS ring
  // (wave-uniform channel base + the lane's 32-bit byte offset of the row; the rows of an X1 row y1 are
  // input rows y1 - 1 .. y1 + 1, clamped into the image: rows outside are zero padding through the taps)
  auto row_offsets = [&](int y1, uint32_t (&ro)[3]) {
#pragma unroll
    for (int k = 0; k < 3; k++) ro[k] = 4u * (lane_base + (uint32_t)(min(max(y1 - 1 + k, 0), H1 - 1) * W1));
  };
  // RING: input row y of k-step ks sits in slot (y & 1) once it has been the bottom row of an X1 row
  auto ring_at = [&](int y, int ks) -> float4& { return s_ring[(((y + 2) & 1) * KS1 + ks) * 64]; };
  auto load_window = [&](const uint32_t (&ro)[3], int ks) {  // k-step ks of the X1 row with input rows `ro`, from memory
    if (RING) {
      win[ks % PD][0] = load_row(ro[2], ks);
    } else {
#pragma unroll
      for (int k = 0; k < 3; k++) win[ks % PD][k] = load_row(ro[k], ks);
    }
  };
  auto load_ring = [&](int y1, int ks) { up[ks & 1][0] = ring_at(y1 - 1, ks), up[ks & 1][1] = ring_at(y1, ks); };
  auto taps_of = [&](int ks, const float* wl, const float* el, const float* er, float (&t)[4]) {
    if (RING) {
      const float4 m3[3] = {up[ks & 1][0], up[ks & 1][1], win[ks % PD][0]};
      dw_taps(m3, wl + 2 * ks * 12, el + 2 * ks * 8, er + 2 * ks * 8, t);
    } else {
      const float4 m3[3] = {win[ks % PD][0], win[ks % PD][RING ? 0 : 1], win[ks % PD][RING ? 0 : 2]};
      dw_taps(m3, wl + 2 * ks * 12, el + 2 * ks * 8, er + 2 * ks * 8, t);
    }
  };
  // one X1 row of the lane's 16 channels (in the accumulators) -> the second depthwise sums.  ReLU of the first block and the
  // row's existence in one v_med3_i32: the integer maximum with 0 is ReLU for non-NaN floats (relu_acc), and `xmax` is INT_MAX or,
  // for an X1 row outside the image (zero padding), 0.  Output pixel A of the lane reads X1 columns (left lane's x3, x0, x1), B
  // (x1, x2, x3), taps in this order.  Odd X1 rows are the middle tap row of one output row (mode 0: tap row 1 into t2); even
  // ones the bottom tap row of the output row above and the top tap row of the one below: both sums are updated in one pass
  // over the channels (tap row 2 into t2, tap row 0 into t2n after its bias), so an accumulator register is dead as soon as
  // its channel is folded.
  auto fold = [&](const floatx16 (&acc)[4], int xmax, int mode /* 0: middle tap row; 1: up + down; 2: down only; 3: up only */) {
#pragma unroll
    for (int p = 0; p < 8; p++) {
      const float4* rec = reinterpret_cast<const float4*>(f2 + rec2(p));
      floatx2 x[4];
#pragma unroll
      for (int i = 0; i < 4; i++)  // (the first block always ends in a ReLU: dwpw2_supported)
        x[i] = floatx2{relu_below(acc[i][2 * p], xmax), relu_below(acc[i][2 * p + 1], xmax)};
      const floatx2 l = {lane_prev(x[3][0]), lane_prev(x[3][1])};
      auto taps = [&](floatx2& sa, floatx2& sb, int k) {
        const float4 w0 = rec[2 * k], w12 = rec[2 * k + 1];
        const floatx2 wl = {w0.x, w0.y}, wk0 = {w0.z, w0.w}, wk1 = {w12.x, w12.y}, wk2 = {w12.z, w12.w};
        sa = __builtin_elementwise_fma(wl, l, sa), sa = __builtin_elementwise_fma(wk1, x[0], sa), sa = __builtin_elementwise_fma(wk2, x[1], sa);
        sb = __builtin_elementwise_fma(wk0, x[1], sb), sb = __builtin_elementwise_fma(wk1, x[2], sb), sb = __builtin_elementwise_fma(wk2, x[3], sb);
      };
      if (mode == 0) {
        taps(t2[0][p], t2[1][p], 1);
      } else {
        if (mode != 2) taps(t2[0][p], t2[1][p], 2);  // (the band's first X1 row has no output row above, its last none below)
        if (mode != 3) {
          const float4 b = rec[6];
          t2n[0][p] = t2n[1][p] = floatx2{b.x, b.y};
          taps(t2n[0][p], t2n[1][p], 0);
        }
      }
      // (the scheduler would otherwise issue all the table reads of the row first: registers this kernel does not have)
      if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  };
  // second pointwise conv of the finished sums t2 -> output row oy.  The lane holds channels qq + 8b + 4*half
  // (qq = r&3, b = r>>2); k-step s = 4b + u needs channel 8b + 2u from the half-0 lanes and 8b + 2u + 1 from the
  // half-1 lanes.  v_permlane32_swap(A, B) leaves A = (A's lower half, B's lower half), B = (A's upper half, B's upper
  // half); with A = the lane's qq = 0, B = qq = 1:  A -> channels 8b | 8b + 1 = k-step u = 0,  B -> 8b + 4 | 8b + 5 = u = 2;
  // with qq = 2, 3: u = 1 and u = 3.
  const int lowest2 = a2.relu ? 0 : (int)0x80000000;
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a2.out + (size_t)a2.out_coff * ohw, 0, 0xFFFFFFFF, 0x00020000);
  const uint32_t out_chan_step = 4u * (uint32_t)ohw;
  auto finish = [&](int oy) {
    floatx2 bop[KS2];  // B operands of the 16 k-steps (2 output pixels each), after the exchange between the halves
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int qq = 0; qq < 4; qq += 2) {
          // (accumulator rows 4b + qq, 4b + qq + 1 = the two channels of pair 2b + qq / 2; ReLU as the integer maximum)
          const floatx2 own = t2[j][2 * b + (qq >> 1)];
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(relu_acc(own[0])), __float_as_uint(relu_acc(own[1])), false, false);
          bop[4 * b + (qq >> 1)][j] = __uint_as_float(sw[0]), bop[4 * b + 2 + (qq >> 1)][j] = __uint_as_float(sw[1]);
        }
    // (buffer stores: the lane's 32-bit byte offset in a vector register, the channel as the scalar offset)
    const uint32_t out_off = 4u * (uint32_t)((frame * a2.out_ctotal + 4 * half) * ohw + (size_t)(oy * a2.ow + ox + 2 * sub));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < CT2; ct++) {
      floatx16 acc2[2];
#pragma unroll
      for (int j = 0; j < 2; j++) {  // (16-byte LDS reads per accumulator, not moves: the LDS pipe is idle, the vector ALU is not)
        int hb = half * 16;  // (an index the compiler cannot merge with the other accumulator's: one read each, no copies)
        asm volatile("" : "+v"(hb));
        const float* b2 = s_b2 + ct * 32 + hb;
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
          const float4 b = reinterpret_cast<const float4*>(b2)[r4];
          acc2[j][4 * r4] = b.x, acc2[j][4 * r4 + 1] = b.y, acc2[j][4 * r4 + 2] = b.z, acc2[j][4 * r4 + 3] = b.w;
        }
      }
      const float* w2l = s_w2 + ct * KS2 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS2; ks++) {
        const float w = w2l[ks * 64];
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, bop[ks][0], acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, bop[ks][1], acc2[1], 0, 0, 0);
      }
      if (live) {
        // (full cout tiles only: dwpw2_supported; ReLU as the integer maximum with 0 -- relu_acc -- or with INT_MIN, the
        // identity: no bound or flag test per store)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const uintx2 v = {(uint32_t)max(__float_as_int(acc2[0][r]), lowest2), (uint32_t)max(__float_as_int(acc2[1][r]), lowest2)};
          __builtin_amdgcn_raw_buffer_store_b64(v, out_rsrc, out_off, (uint32_t)(ct * 32 + (r & 3) + 8 * (r >> 2)) * out_chan_step, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int nrows = 2 * R + 1;
  {
    uint32_t ro[3];
    row_offsets(2 * oy0 - 1, ro);
    if (RING) {  // the two upper input rows of the band's first X1 row, all channels
#pragma unroll
      for (int ks = 0; ks < KS1; ks++) {
        ring_at(2 * oy0 - 2, ks) = load_row(ro[0], ks);
        ring_at(2 * oy0 - 1, ks) = load_row(ro[1], ks);
      }
    }
#pragma unroll
    for (int ks = 0; ks < PD; ks++) load_window(ro, ks);
    if (RING) load_ring(2 * oy0 - 1, 0), load_ring(2 * oy0 - 1, 1);
  }
#pragma unroll 1
  for (int row = 0; row < nrows; row++) {
    const int y1 = 2 * oy0 - 1 + row;  // X1 row
    const bool row1ok = y1 >= 0 && y1 < H1;
    const bool last = row == nrows - 1;
    uint32_t ro[3], ro_next[3];  // this X1 row's input rows, the next one's (the last row: its own)
    row_offsets(y1, ro);
    row_offsets(last ? y1 : y1 + 1, ro_next);
    // (an X1 row outside the image has two padding input rows; it is computed from whatever the
    // clamped addresses hold and dropped below by the v_med3 of the fold)
    // Input rows outside the image are zero padding: `wl` points into the copy of the LDS table that has those taps zeroed,
    // `el` / `er` into the matching copy of the edge taps -- or the all-zero one for a lane at the left / right image border.
    const int tbv = dw_variant(y1 - 1 >= 0, y1 + 1 < H1);
    const float* wl = s_dw1 + (tbv * C1 + half) * 12;
    const float* el = s_e1 + ((leftok ? tbv : 3) * C1 + half) * 8;
    const float* er = s_e1 + ((rightok ? tbv : 3) * C1 + half) * 8 + 4;
    floatx16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {  // bias: four 16-byte LDS reads per accumulator (an address the compiler cannot merge)
      int hb = half * 16;
      asm volatile("" : "+v"(hb));
      const float* b1 = s_b1 + hb;
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        const float4 b = reinterpret_cast<const float4*>(b1)[r4];
        acc[p][4 * r4] = b.x, acc[p][4 * r4 + 1] = b.y, acc[p][4 * r4 + 2] = b.z, acc[p][4 * r4 + 3] = b.w;
      }
    }
    // software pipeline: the MFMAs of k-step ks run beside the depthwise arithmetic of ks + 1
    float tcur[4];
    taps_of(0, wl, el, er, tcur);
#pragma unroll
    for (int ks = 0; ks < KS1; ks++) {
      // the queue slots of step ks are consumed (into tcur): refill them with step ks + PD / ks + 2 -- of the next X1 row
      // once this one runs out
      if (RING) ring_at(y1 + 1, ks) = win[ks % PD][0];  // this step's bottom row: the middle / top row of the next two X1 rows
      // (the next X1 row's first windows are requested on the band's last row too -- ro_next is that row's own address then,
      // a harmless re-read: behind `if (!last)` the requests sat in a conditionally executed block and every wait of the
      // row's last PD - 1 k-steps was computed as if they had not been issued -- vmcnt(2) (1) (0) where (3) (3) (3) was meant)
      if (ks + PD < KS1) load_window(ro, ks + PD);
      else load_window(ro_next, ks + PD - KS1);
      if (RING) {
        if (ks + 2 < KS1) load_ring(y1, ks + 2);
        else load_ring(y1 + 1, ks + 2 - KS1);
      }
      // (memory instructions stay on their side of this line, everything else may cross: in the unrolled loop hipcc sank
      // the requests of an even k-step down to those of the odd one behind it -- six loads in a burst, the first three
      // waited for at once: a queue of depth zero for every other k-step)
      __builtin_amdgcn_sched_barrier(0);
      const float w = s_w1[ks * 64 + lane];
      float tnext[4];
      if (ks + 1 < KS1) taps_of(ks + 1, wl, el, er, tnext);
      // (no sched_group_barrier here: fp32 MFMAs and the wave's own vector instructions do not
      // overlap on gfx950 -- tools/ubench/mfma_f32_cost.hip -- and the compiler's own order
      // measured 3 % faster than "one MFMA, fourteen VALU")
#pragma unroll
      for (int p = 0; p < 4; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, tcur[p], acc[p], 0, 0, 0);
      if (ks + 1 < KS1) {
#pragma unroll
        for (int p = 0; p < 4; p++) tcur[p] = tnext[p];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int xmax = row1ok ? 0x7FFFFFFF : 0;
    if (row & 1) {  // X1 row 2(oy0 + j): the middle tap row of output row oy0 + j
      fold(acc, xmax, 0);
    } else {        // X1 row 2(oy0 + j) - 1: bottom tap row of output row oy0 + j - 1, top tap row of oy0 + j
      if (row == 0) fold(acc, xmax, 2);
      else if (last) fold(acc, xmax, 3);
      else fold(acc, xmax, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (row > 0) finish(oy0 + (row >> 1) - 1);
#pragma unroll
      for (int p = 0; p < 8; p++) t2[0][p] = t2n[0][p], t2[1][p] = t2n[1][p];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Dense 3x3 (any stride / dilation, pad = dilation, cout <= 16) as implicit GEMM on
// v_mfma_f32_16x16x4_f32.  K runs over (input channel, tap) with the 9 taps of a channel padded
// to 12 = 3 instructions of 4 taps: lane l supplies tap 4*s + (l>>4) of channel ci for pixel
// column l&15 (a gather from the input plane, zero outside the image), so the accumulation order
// is exactly ci-major / tap-minor.  A wave owns PG groups of 16 consecutive output pixels
// (numbered over the whole batch); D: reg r, lane l -> cout 4*(l>>4) + r, pixel column l&15.
// a.w: packed [cin][3][64] (lane l of slot s: W[l&15][ci][4s + (l>>4)], 0 for taps >= 9 / cout pad).
template <int PG, int U, int SK = 1>
__global__ __launch_bounds__(256) void k_conv3x3_mfma(ConvArgs3 p3) {
  const ConvArgs a = p3.a[blockIdx.y];  // (a copy: see k_dwpw_mfma)
  __shared__ float s_red[SK > 1 ? 3 * 4 * PG * 64 : 1];  // split-K: partial tiles of waves 1..3
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, quad = lane >> 4;
  const int ohw = a.oh * a.ow, ihw = a.ih * a.iw;
  const long total = (long)a.B * ohw;
  const long p0 = ((SK == 1 ? ((long)blockIdx.x * 4 + wave) : (long)blockIdx.x) * PG) * 16 + col;

  // per pixel group: base pointer; per (slot, group): tap offset validity
  const float* base[PG];
  int off[3];
  unsigned valid = 0;  // bit (s*PG + g)
  bool live[PG];
  long pidx[PG];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    const int t = 4 * s + quad;
    const int ky = t / 3, kx = t - ky * 3;
    off[s] = (t < 9) ? ((ky - 1) * a.dil * a.iw + (kx - 1) * a.dil) : 0;
  }
  int goff[3][PG];  // per (slot, group) gather offset, 0 (the pixel itself: always valid) when masked
#pragma unroll
  for (int g = 0; g < PG; g++) {
    const long p = p0 + (long)g * 16;
    live[g] = p < total;
    pidx[g] = p;
    const long pc = live[g] ? p : 0;
    const size_t frame = pc / ohw;
    const int rem = (int)(pc - (long)frame * ohw);
    const int oy = rem / a.ow, ox = rem - oy * a.ow;
    const int iy = oy * a.stride, ix = ox * a.stride;
    base[g] = a.in + frame * a.in_ctotal * (size_t)ihw + (size_t)iy * a.iw + ix;
#pragma unroll
    for (int s = 0; s < 3; s++) {
      const int t = 4 * s + quad;
      const int ky = t / 3, kx = t - ky * 3;
      const int yy = iy + (ky - 1) * a.dil, xx = ix + (kx - 1) * a.dil;
      const bool ok = live[g] && t < 9 && yy >= 0 && yy < a.ih && xx >= 0 && xx < a.iw;
      valid |= (ok ? 1u : 0u) << (s * PG + g);
      goff[s][g] = ok ? off[s] : 0;
    }
  }
  floatx4 acc[PG];
#pragma unroll
  for (int g = 0; g < PG; g++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int co = 4 * quad + r;
      acc[g][r] = (co < a.cout && (SK == 1 || wave == 0)) ? a.bias[co] : 0.0f;
    }
  const float* wp = a.w + lane;
  // U channels per iteration: their gathers are issued together (deep layers have one short
  // dependent MFMA chain per wave and would otherwise pay a memory round trip per tap).
  // Gathers are unconditional from clamped addresses; masked taps are zeroed by select.
  // split-K: the block's four waves share the pixel groups and take a quarter of the channels each
  const int cper = a.cin / SK, cbeg = (SK == 1) ? 0 : wave * cper, cend = cbeg + cper;
  for (int c0 = cbeg; c0 < cend; c0 += U) {
    float x[U][3][PG], w[U][3];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const ptrdiff_t coff = (ptrdiff_t)(c0 + u) * ihw;
#pragma unroll
      for (int s = 0; s < 3; s++) {
        w[u][s] = wp[((c0 + u) * 3 + s) * 64];
#pragma unroll
        for (int g = 0; g < PG; g++) x[u][s][g] = base[g][coff + goff[s][g]];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int s = 0; s < 3; s++)
#pragma unroll
        for (int g = 0; g < PG; g++) {
          const float xv = ((valid >> (s * PG + g)) & 1u) ? x[u][s][g] : 0.0f;
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][s], xv, acc[g], 0, 0, 0);
        }
  }
  if (SK > 1) {  // fixed-order reduction (w = 1, 2, 3) into wave 0
    if (wave > 0) {
#pragma unroll
      for (int g = 0; g < PG; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) s_red[(((wave - 1) * PG + g) * 4 + r) * 64 + lane] = acc[g][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; w++)
#pragma unroll
      for (int g = 0; g < PG; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[g][r] += s_red[((w * PG + g) * 4 + r) * 64 + lane];
  }
#pragma unroll
  for (int g = 0; g < PG; g++) {
    if (!live[g]) continue;
    const size_t frame = pidx[g] / ohw;
    const int rem = (int)(pidx[g] - (long)frame * ohw);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int co = 4 * quad + r;
      if (co < a.cout) {
        float v = acc[g][r];
        if (a.relu) v = fmaxf(v, 0.0f);
        a.out[(frame * a.out_ctotal + a.out_coff + co) * ohw + rem] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Dense 3x3 "row" kernel (stem s2, RFB 3x3 with dilation 1/2/3/5): v_mfma_f32_16x16x4_f32 with
// 4 input channels per instruction.  Lane (q = l>>4, j = l&15) owns the 4 output pixels of pixel
// group j of its wave tile for input channel 4*kc + q; per channel chunk it loads its three input
// rows as 16-byte segments (one per row, two for stride 2) and every tap is a register select or
// a cross-lane shuffle of those rows -- no per-tap gathers (the gather kernel above issues one
// scalar load per tap, which is what bounds it).  The first / last HL lanes of each 16-lane
// quad are halo providers (tiles overlap by 2*HL groups).  Needs ow % 4 == 0, iw % 4 == 0,
// cout <= 16.  Accumulation order: channel chunks of 4, taps inside (not the oracle's order;
// the difference is fp32 rounding, ~1e-7 relative).
// a.w: packed [ceil(cin/4)][9][64] (lane l: W[l&15][4*kc + (l>>4)][tap], zero padded).
template <int S, int DIL>
__device__ __forceinline__ void conv3x3_rows_body(const ConvArgs& a) {
  const int dil = DIL == 0 ? a.dil : DIL;
  // DIL = 0: the dilation is a.dil (<= 8) at run time: two halo lanes and run-time tap offsets instead of compile-time ones
  constexpr int HL = (S == 1) ? (DIL == 0 ? 2 : (DIL + 3) / 4) : 1;  // halo lanes on each side of a quad
  constexpr int NG = 16 - 2 * HL;                   // output pixel groups per wave tile
  extern __shared__ float s_w[];                    // packed weights when they fit (a.dbg = 1)
  // (a merged launch is sized for the widest halo among its convs: blocks past this conv's last tile leave, whole
  // blocks and before the barrier)
  // (a contiguous range of tiles per XCD -- remap_block: the tiles above and below a tile share its input rows, three
  // dilations deep in the RFB's merged launch; the grid is a multiple of 8 along x)
  const int bx = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  if ((long)bx * 4 * NG - HL >= (long)a.B * ((a.oh * a.ow) >> 2)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, j16 = lane & 15;
  const int cin4 = (a.cin + 3) >> 2;
  const bool w_in_lds = a.dbg != 0;
  // (The k-loop of these convs is two to four channel chunks long: what surrounds it counts as much.  32-bit index
  // arithmetic -- a tensor below 4 GiB has fewer than 2^28 pixel groups, and the 64-bit division this used to be is a
  // ~100-instruction routine of quarter-rate multiplies; the lane's four bias values as one 16-byte load; the store
  // offsets from one 32-bit base, below.)
  const int ohw = a.oh * a.ow, gpf = ohw >> 2, gpr = a.ow >> 2;
  const int total = a.B * gpf;
  const int g = (bx * 4 + wave) * NG + j16 - HL;
  const bool inrange = g >= 0 && g < total;
  const bool live = inrange && j16 >= HL && j16 < 16 - HL;
  const uint32_t frame32 = inrange ? (uint32_t)g / (uint32_t)gpf : 0u;
  const int rem = inrange ? g - (int)frame32 * gpf : 0;
  const int oy = rem / gpr, ox = (rem - oy * gpr) * 4;
  const int ihw = a.ih * a.iw;
  const int cout = a.cout;

  floatx4 acc[4];
  {
    // (unconditional loads from clamped indices, zeroed by select: a load in a lane-dependent branch is a wait of its own)
    float4 b4;
    if ((cout & 3) == 0) {  // (wave-uniform) the lane's channels 4q .. 4q+3 exist together or not at all
      b4 = *reinterpret_cast<const float4*>(a.bias + min(4 * q, cout - 4));
      if (4 * q >= cout) b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const float* bias = a.bias;
      const float b0 = bias[min(4 * q + 0, cout - 1)], b1 = bias[min(4 * q + 1, cout - 1)], b2 = bias[min(4 * q + 2, cout - 1)], b3 = bias[min(4 * q + 3, cout - 1)];
      b4.x = 4 * q + 0 < cout ? b0 : 0.0f, b4.y = 4 * q + 1 < cout ? b1 : 0.0f;
      b4.z = 4 * q + 2 < cout ? b2 : 0.0f, b4.w = 4 * q + 3 < cout ? b3 : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j][0] = b4.x, acc[j][1] = b4.y, acc[j][2] = b4.z, acc[j][3] = b4.w;
  }

  const int ix0 = ox * S;
  bool rowok[3];
  uint32_t rowoff[3];
  const uint32_t frame_off = __umul24(__umul24(frame32, (uint32_t)a.in_ctotal), (uint32_t)ihw);
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int iy = oy * S + (r - 1) * dil;
    rowok[r] = iy >= 0 && iy < a.ih;
    rowoff[r] = frame_off + (uint32_t)(min(max(iy, 0), a.ih - 1) * a.iw + ix0);
  }
  // column validity of the shifted taps (stride 1: x + j -/+ DIL inside the row; stride 2: left edge)
  bool lok[4], rok[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    lok[j] = S == 1 ? (ox + j - dil >= 0) : (j > 0 || ix0 > 0);
    rok[j] = S == 1 ? (ox + j + dil < a.ow) : true;
  }
  const float* __restrict__ in = a.in;
  const float* wg = a.w + lane;

  struct Rows {
    float4 m0[3], m1[3];
  };
  auto load_rows = [&](int kc, Rows& w) {
    const uint32_t c = (uint32_t)min(4 * kc + q, a.cin - 1) * (uint32_t)ihw;  // padded channels: weight is 0
#pragma unroll
    for (int r = 0; r < 3; r++) {
      w.m0[r] = *reinterpret_cast<const float4*>(in + (rowoff[r] + c));
      if (S == 2) w.m1[r] = *reinterpret_cast<const float4*>(in + (rowoff[r] + c + 4u));
    }
  };
  // value of this lane's row segment at column offset c (relative to its first pixel), c in [-8, 11]
  auto col = [&](const float4& m, int c) -> float {
    if (DIL == 0) {  // run-time offset: floor division by shift, unconditional shuffle (lane + 0 = itself)
      const int o = c >> 2, k = c & 3;
      const float v = k == 0 ? m.x : (k == 1 ? m.y : (k == 2 ? m.z : m.w));
      return __shfl(v, lane + o);
    }
    const int o = (c >= 0) ? (c >> 2) : -((3 - c) >> 2);  // lane offset, floor(c / 4)
    const int k = c - 4 * o;
    const float v = k == 0 ? m.x : (k == 1 ? m.y : (k == 2 ? m.z : m.w));
    // a quad's 16 lanes are one DPP row: the neighbour's value is a row shift on the vector ALU (one instruction; __shfl is
    // a ds_bpermute -- an LDS-pipe round trip the MFMAs behind it wait for).  Lanes past the row's ends get 0: they are halo
    // providers, their own outputs are not stored.
    switch (o) {
      case 0: return v;
      case -1: return row_shift<0x111>(v);  // row_shr:1 = the value of lane - 1
      case -2: return row_shift<0x112>(v);
      case 1: return row_shift<0x101>(v);   // row_shl:1 = the value of lane + 1
      case 2: return row_shift<0x102>(v);
      default: return __shfl(v, lane + o);
    }
  };

  // The channel loop, once per source of the weights (wave-uniform: a run-time choice between an LDS and a global pointer
  // inside the loop made every weight read a FLAT load).  A tap row outside the image zeroes the lane's ROW SEGMENT once
  // (4 selects; the neighbours' halo values come out of their own zeroed segments -- same output row, same tap row --
  // or are masked by the column tests) instead of each of the 12 operands made from it, and the column tests are compiled
  // in only where they can fail: left of pixel j < DIL, right of pixel j > 3 - DIL.  54 selects per chunk became 18.
  // (as an integer AND with an all-ones / all-zero mask: `keep ? m : 0` became a BRANCH around four moves, with a wait for
  // the row's load inside it -- and behind a conditionally executed wait every later wait is for everything in flight, the
  // next chunk's rows included)
  auto zero_if = [](bool keep, const float4& m) {
    const int k = keep ? -1 : 0;
    return make_float4(__int_as_float(__float_as_int(m.x) & k), __int_as_float(__float_as_int(m.y) & k), __int_as_float(__float_as_int(m.z) & k),
                       __int_as_float(__float_as_int(m.w) & k));
  };
  // Rows (and, when they come from memory, the nine weights) of chunk kc + 1 are requested before chunk kc's MFMAs, into the
  // other half of a register ping-pong -- the loop runs two chunks per iteration so that no copy is needed, and the
  // sched_barrier keeps the requests where they are written (round 5: with `cur = nxt` copies at the loop's end hipcc sank the
  // loads down to the copies and waited for them there, one exposed memory round trip per 36 MFMAs).
  // The first chunk's rows are requested in front of the weight table's copy and its barrier (the block's tables, bias and
  // first rows in one memory round trip instead of three).
  Rows rows[2];
  load_rows(0, rows[0]);
  if (w_in_lds) {
    copy_table16<4>(s_w, a.w, cin4 * 9 * 16);
    __syncthreads();
  }
  auto run = [&](auto in_lds) {
    constexpr bool kLds = decltype(in_lds)::value;
    float wts[2][9];
    auto load_wts = [&](int kc, float (&w)[9]) {
      if (kLds) return;
#pragma unroll
      for (int t = 0; t < 9; t++) w[t] = wg[(kc * 9 + t) * 64];
    };
    auto chunk = [&](int kc, const Rows& cur, const float (&wcur)[9]) {
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const bool ok = rowok[r];
        float x[3][4];  // [kx][pixel]
        if (S == 1) {
          const float4 m = zero_if(ok, cur.m0[r]);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            // shuffles first, unconditionally: a cross-lane read inside a divergent branch would
            // see inactive source lanes
            const float vl = col(m, j - dil), vc = DIL == 0 ? (j == 0 ? m.x : (j == 1 ? m.y : (j == 2 ? m.z : m.w))) : col(m, j), vr = col(m, j + dil);
            x[0][j] = (DIL != 0 && j >= DIL) ? vl : (lok[j] ? vl : 0.0f);
            x[1][j] = vc;
            x[2][j] = (DIL != 0 && j + DIL <= 3) ? vr : (rok[j] ? vr : 0.0f);
          }
        } else {
          const float4 m0 = zero_if(ok, cur.m0[r]), m1 = zero_if(ok, cur.m1[r]);
          const float left = row_shift<0x111>(m1.w);
          x[0][0] = lok[0] ? left : 0.0f, x[0][1] = m0.y, x[0][2] = m0.w, x[0][3] = m1.y;
          x[1][0] = m0.x, x[1][1] = m0.z, x[1][2] = m1.x, x[1][3] = m1.z;
          x[2][0] = m0.y, x[2][1] = m0.w, x[2][2] = m1.y, x[2][3] = m1.w;
        }
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          const float w = kLds ? s_w[(kc * 9 + r * 3 + kx) * 64 + lane] : wcur[r * 3 + kx];
#pragma unroll
          for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x[kx][j], acc[j], 0, 0, 0);
        }
      }
    };
    load_wts(0, wts[0]);
    // (the requests are unconditional -- past the last chunk a harmless re-read of it: behind a branch around a load the
    // wait counters are not statically known either)
    for (int kc = 0; kc < cin4; kc += 2) {
      load_rows(min(kc + 1, cin4 - 1), rows[1]), load_wts(min(kc + 1, cin4 - 1), wts[1]);
      __builtin_amdgcn_sched_barrier(0);
      chunk(kc, rows[0], wts[0]);
      if (kc + 1 >= cin4) break;
      load_rows(min(kc + 2, cin4 - 1), rows[0]), load_wts(min(kc + 2, cin4 - 1), wts[0]);
      __builtin_amdgcn_sched_barrier(0);
      chunk(kc + 1, rows[1], wts[1]);
    }
  };
  if (w_in_lds) run(std::true_type{});
  else run(std::false_type{});
  if (!live) return;
  const int pix = oy * a.ow + ox;
  // byte offset of the lane's first channel from the tensor base, a channel step per store; ReLU as the integer maximum
  // with 0 (relu_acc), or with INT_MIN = the identity: no flag test, no canonicalising v_max
  char* out = reinterpret_cast<char*>(a.out);
  const uint32_t step = 4u * (uint32_t)ohw;
  uint32_t o = 4u * (__umul24(__umul24(frame32, (uint32_t)a.out_ctotal) + (uint32_t)(a.out_coff + 4 * q), (uint32_t)ohw) + (uint32_t)pix);
  const int lowest = a.relu ? 0 : (int)0x80000000;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (4 * q + r < cout) {
      const float4 v = make_float4(__int_as_float(max(__float_as_int(acc[0][r]), lowest)), __int_as_float(max(__float_as_int(acc[1][r]), lowest)),
                                   __int_as_float(max(__float_as_int(acc[2][r]), lowest)), __int_as_float(max(__float_as_int(acc[3][r]), lowest)));
      *reinterpret_cast<float4*>(out + o) = v;
    }
    o += step;
  }
}

// Merged launches (blockIdx.y selects the conv; same shapes, possibly different dilations -- the RFB's three dilated
// 3x3 convs): DIL = 0 dispatches, per block, to the body compiled for the conv's dilation (compile-time tap offsets:
// a register pick or one shuffle per tap, where the run-time form pays three selects and an LDS-pipe permute per tap).
template <int S, int DIL>
__global__ __launch_bounds__(256) void k_conv3x3_rows_mfma(ConvArgs3 p3) {
  const ConvArgs a = p3.a[blockIdx.y];  // (a copy: see k_dwpw_mfma)
  if (DIL != 0) {
    conv3x3_rows_body<S, DIL>(a);
    return;
  }
  switch (a.dil) {
    case 1: conv3x3_rows_body<S, 1>(a); break;
    case 2: conv3x3_rows_body<S, 2>(a); break;
    case 3: conv3x3_rows_body<S, 3>(a); break;
    case 5: conv3x3_rows_body<S, 5>(a); break;
    default: conv3x3_rows_body<S, 0>(a); break;
  }
}

// ------------------------------------------------------------------------------------------------
// RFB tail in ONE launch (round 4): the three dilated 3x3 convs of the branches (b0: d2, b1: d3, b2: d5; 16 -> 16 each, no
// activation) and relu(ConvLinear(cat48) + shortcut(x)) -- SURVEY 8.1 rows 15, 18, 22, 23, 24.  The 48-channel concat
// tensor (29.5 MB per 32-frame batch of the 640 model, written by one launch and read by the next) never exists: a wave
// computes the three convs for its tile exactly as k_conv3x3_rows_mfma does (same lane layout, same MFMA sequence, one
// conv after the other; since round 5 WITHOUT halo lanes -- rfb_dilated: the tile's outer neighbours come from memory, all 16
// lanes of a quad own outputs; the form with the halo of the widest dilation, 12 outputs per quad, was 66.6 us for 58.0 and
// is gone) and keeps the results in their accumulators; the D
// layout of v_mfma_f32_16x16x4_f32 -- register r of lane (q, j) = channel 4q + r of pixel group j -- IS a B operand of the
// same instruction for the k-chunk of channels {r, 4 + r, 8 + r, 12 + r}, so the 1x1 conv multiplies them straight out of
// the registers (weights packed in that channel order: pack_rfb_tail_weights), then runs over the 64 channels of x from
// memory (16-byte row segments, the row kernel's own load shape) and stores 64 output channels.
// Accumulation order of the 1x1: bias; the branch channels in the permuted chunk order; x's channels in chunks of 4 -- fp32
// rounding apart from the two-launch form (which itself differs from the oracle's order by the same kind of rounding).
// a3.a[b]: conv b as for launch_conv3x3_rows_mfma (w = row packing); fin: the summed 1x1 (in2 = x with in2_ctotal, w = tail
// packing [28 chunks][4 cout tiles][64], bias = both biases summed, out).
constexpr int kTailNG = 16;           // pixel groups (4 pixels each) per wave: every lane of a quad owns one
constexpr int kTailChunks = 12 + 16;  // 48 branch channels + 64 channels of x, 4 per MFMA
struct RfbTailArgs {
  ConvArgs3 a3;
  ConvArgs fin;
};

template <int DIL>
__device__ __forceinline__ void rfb_dilated(const ConvArgs& a, const float* __restrict__ wsrc, int lane, uint32_t frame32, int oy,
                                            int ox, floatx4 (&acc)[4]) {
  const int q = lane >> 4;
  const int ihw = a.ih * a.iw, cin4 = (a.cin + 3) >> 2;
  {
    const float4 b4 = *reinterpret_cast<const float4*>(a.bias + 4 * q);  // (cout = 16: rfb_tail_supported)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j][0] = b4.x, acc[j][1] = b4.y, acc[j][2] = b4.z, acc[j][3] = b4.w;
  }
  bool rowok[3];
  uint32_t rowoff[3];
  const uint32_t frame_off = __umul24(__umul24(frame32, (uint32_t)a.in_ctotal), (uint32_t)ihw);
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int iy = oy + (r - 1) * DIL;
    rowok[r] = iy >= 0 && iy < a.ih;
    rowoff[r] = frame_off + (uint32_t)(min(max(iy, 0), a.ih - 1) * a.iw + ox);
  }
  bool lok[4], rok[4];
#pragma unroll
  for (int j = 0; j < 4; j++) lok[j] = ox + j - DIL >= 0, rok[j] = ox + j + DIL < a.ow;
  const float* __restrict__ in = a.in;
  // value of this lane's row segment at column offset c (relative to its first pixel): a register pick or a DPP row shift
  // (a quad's 16 lanes are one DPP row; lanes past its ends get 0: they are halo providers)
  auto col = [&](const float4& m, int c) -> float {
    const int o = (c >= 0) ? (c >> 2) : -((3 - c) >> 2);
    const int k = c - 4 * o;
    const float v = k == 0 ? m.x : (k == 1 ? m.y : (k == 2 ? m.z : m.w));
    switch (o) {
      case 0: return v;
      case -1: return row_shift<0x111>(v);
      case -2: return row_shift<0x112>(v);
      case 1: return row_shift<0x101>(v);
      default: return row_shift<0x102>(v);
    }
  };
  auto load_rows = [&](int kc, float4 (&m)[3]) {
    const uint32_t c = (uint32_t)min(4 * kc + q, a.cin - 1) * (uint32_t)ihw;
#pragma unroll
    for (int r = 0; r < 3; r++) m[r] = *reinterpret_cast<const float4*>(in + (rowoff[r] + c));
  };
  // No halo lanes: all 16 lanes of a quad own outputs, and the two pixel groups left / right of the tile come
  // from memory -- ONE more row segment per lane and row: lanes 0, 1 read groups g - 2 (the tile's left neighbours), lanes
  // 14, 15 groups g + 2, the others their own again (a cache hit).  A neighbour value is then dpp(own row) | dpp(edge row):
  // the row shifts deliver 0 to lanes without a source, the edge rows are masked to 0 outside their two lanes.
  const int j16 = lane & 15;
  const int e_shift = j16 < 2 ? -8 : (j16 >= 14 ? 8 : 0);
  const uint32_t e_last = (uint32_t)a.B * (uint32_t)a.in_ctotal * (uint32_t)ihw - 4u;
  const int e_mask_l = j16 < 2 ? -1 : 0, e_mask_r = j16 >= 14 ? -1 : 0;
  auto load_edge = [&](int kc, int r) -> float4 {
    const uint32_t c = (uint32_t)min(4 * kc + q, a.cin - 1) * (uint32_t)ihw;
    const int off = (int)(rowoff[r] + c) + e_shift;
    return *reinterpret_cast<const float4*>(in + (uint32_t)min(max(off, 0), (int)e_last));
  };
  auto pick = [](const float4& m, int k) { return k == 0 ? m.x : (k == 1 ? m.y : (k == 2 ? m.z : m.w)); };
  auto masked = [](float v, int mask) { return __int_as_float(__float_as_int(v) & mask); };
  auto either = [](float a0, float b0) { return __int_as_float(__float_as_int(a0) | __float_as_int(b0)); };
  // as col(), with the tile's outer neighbours from the edge row e
  auto col_e = [&](const float4& m, const float4& e, int c) -> float {
    const int o = (c >= 0) ? (c >> 2) : -((3 - c) >> 2);
    const int k = c - 4 * o;
    const float v = pick(m, k);
    switch (o) {
      case 0: return v;
      case -1: return either(row_shift<0x111>(v), row_shift<0x101>(masked(pick(e, k), e_mask_l)));  // lane 0 <- lane 1's left edge
      case -2: return either(row_shift<0x112>(v), masked(pick(e, k), e_mask_l));
      case 1: return either(row_shift<0x101>(v), row_shift<0x111>(masked(pick(e, k), e_mask_r)));   // lane 15 <- lane 14's right edge
      default: return either(row_shift<0x102>(v), masked(pick(e, k), e_mask_r));
    }
  };
  // Rows AND weights of chunk kc + 1 are requested before chunk kc's MFMAs and held in the other half of a register ping-pong
  // (the loop is fully unrolled: cin = 16, rfb_tail_supported); the sched_barrier keeps the requests where they are written.
  // Round 5: with `cur = nxt` copies at the loop's end hipcc sank the next rows' loads down to the copies and waited for them
  // there, and the nine weights of a chunk were requested at the top of the chunk that multiplies them -- two exposed memory
  // round trips per 36 MFMAs, SQ_WAIT_INST_ANY 73 % of the kernel's wave-cycles.
  auto load_wts = [&](int kc, float (&w)[9]) {
#pragma unroll
    for (int t = 0; t < 9; t++) w[t] = wsrc[(kc * 9 + t) * 64 + lane];
  };
  constexpr int kChunks = 4;
  float4 rows[2][3], edges[3];  // (the edge rows in ONE set of registers: row r of the next chunk is requested as soon as row r
                                 // of this one has been read -- a chunk's MFMAs ahead of its use, 12 registers instead of 24)
  float wts[2][9];
  load_rows(0, rows[0]);
#pragma unroll
  for (int r = 0; r < 3; r++) edges[r] = load_edge(0, r);
  load_wts(0, wts[0]);
  (void)cin4;
#pragma unroll
  for (int kc = 0; kc < kChunks; kc++) {
    const float4 (&cur)[3] = rows[kc & 1];
    if (kc + 1 < kChunks) load_rows(kc + 1, rows[(kc + 1) & 1]), load_wts(kc + 1, wts[(kc + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const float4 m = rowok[r] ? cur[r] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 e = rowok[r] ? edges[r] : make_float4(0.f, 0.f, 0.f, 0.f);
      float x[3][4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float vl = col_e(m, e, j - DIL), vc = col(m, j), vr = col_e(m, e, j + DIL);
        x[0][j] = j >= DIL ? vl : (lok[j] ? vl : 0.0f);
        x[1][j] = vc;
        x[2][j] = j + DIL <= 3 ? vr : (rok[j] ? vr : 0.0f);
      }
      if (kc + 1 < kChunks) edges[r] = load_edge(kc + 1, r);
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const float w = wts[kc & 1][r * 3 + kx];
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x[kx][j], acc[j], 0, 0, 0);
      }
    }
  }
}

// (Register budget: with `amdgpu_waves_per_eu(2, 2)` -- "plan for two waves per SIMD, up to 256 registers" -- hipcc comes out
// at 159 unified registers, which the hardware still runs three waves of; left to itself it splits them 96 + 84 accumulation
// registers = 180, two waves.  Capped at 168 in the earlier form of the kernel: 69.9 -> 68.9 us alone.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_rfb_tail(RfbTailArgs t) {
  const ConvArgs& fin = t.fin;
  extern __shared__ float s_tail[];
  // (LDS holds the 1x1's weights only, 28 KB: with the dilated convs' 27 KB beside them a CU took two blocks, not the three
  // its registers allow; those are read from L1 / L2 like the row kernel's when they do not fit)
  float* s_wf = s_tail;                      // [kTailChunks][4][64]
  const int bx = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  const int ohw = fin.oh * fin.ow, gpf = ohw >> 2, gpr = fin.ow >> 2;
  const int total = fin.B * gpf;
  if ((long)bx * 4 * kTailNG >= (long)total) return;  // whole block, before the barrier
  // The 1x1's weight table is REQUESTED here and stored to LDS behind the first dilated conv, in front of the barrier its
  // first reader (fold_branch) needs: the table's round trip runs beside that conv's loads and MFMAs instead of in front of
  // everything (28 registers held meanwhile; the kernel has two waves per SIMD either way).
  constexpr int kTab4 = kTailChunks * 4 * 16, kTabU = (kTab4 + 255) / 256;
  static_assert(kTabU <= 8, "the table's registers");
  typedef float floatx32 __attribute__((ext_vector_type(32)));  // (a vector, not an array: an array of float4 went to scratch)
  floatx32 tab;
#pragma unroll
  for (int u = 0; u < kTabU; u++) {
    const float4 v = reinterpret_cast<const float4*>(fin.w)[min((int)threadIdx.x + 256 * u, kTab4 - 1)];
    tab[4 * u] = v.x, tab[4 * u + 1] = v.y, tab[4 * u + 2] = v.z, tab[4 * u + 3] = v.w;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, j16 = lane & 15;
  const int g = (bx * 4 + wave) * kTailNG + j16;
  const bool inrange = g < total;
  const bool live = inrange;
  const uint32_t frame32 = inrange ? (uint32_t)g / (uint32_t)gpf : 0u;
  const int rem = inrange ? g - (int)frame32 * gpf : 0;
  const int oy = rem / gpr, ox = (rem - oy * gpr) * 4;

  // 1x1 over [48 branch channels | 64 channels of x] -> 64 output channels: out[m][pixel][r] = channel 16 m + 4 q + r.
  // Every branch is multiplied into the 1x1's accumulators as soon as its conv is done (its 16 result registers are free
  // again for the next branch): 16 + 64 accumulators instead of 48 + 64.
  floatx4 out[4][4];
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const float4 b4 = *reinterpret_cast<const float4*>(fin.bias + 16 * m + 4 * q);
#pragma unroll
    for (int p = 0; p < 4; p++) out[m][p][0] = b4.x, out[m][p][1] = b4.y, out[m][p][2] = b4.z, out[m][p][3] = b4.w;
  }
  const float* wl = s_wf + lane;
  auto fold_branch = [&](int b, const floatx4 (&br)[4]) {  // br[pixel][r] = channel 16 b + 4 q + r
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int chunk = b * 4 + r;
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const float w = wl[(chunk * 4 + m) * 64];
#pragma unroll
        for (int p = 0; p < 4; p++) out[m][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, br[p][r], out[m][p], 0, 0, 0);
      }
    }
  };
  {
    floatx4 br[4];
    rfb_dilated<2>(t.a3.a[0], t.a3.a[0].w, lane, frame32, oy, ox, br);
#pragma unroll
    for (int u = 0; u < kTabU; u++)
      if ((int)threadIdx.x + 256 * u < kTab4)
        reinterpret_cast<float4*>(s_wf)[threadIdx.x + 256 * u] = make_float4(tab[4 * u], tab[4 * u + 1], tab[4 * u + 2], tab[4 * u + 3]);
    __syncthreads();
    fold_branch(0, br);
  }
  {
    floatx4 br[4];
    rfb_dilated<3>(t.a3.a[1], t.a3.a[1].w, lane, frame32, oy, ox, br);
    fold_branch(1, br);
  }
  // x's row segments: channel 4 kc + q of the lane's 4 pixels, sixteen chunks.  The first eight are requested in front of the
  // third branch's fold, the other eight behind it, and PINNED there: left to itself hipcc keeps the sixteen addresses instead
  // and requests every chunk one chunk -- 16 MFMAs, a third of a memory round trip -- before its use.
  // (Measured alone in round 3, 640 model at batch 32: requested behind the convs 76 us; before the convs with two chunks of rows
  // in flight, 184 registers = two waves per SIMD, 79 us; capped at 128 registers for four waves, 18 dwords spilled, 78 us;
  // with the convs' weights in LDS too, 55 KB = two blocks per CU, 80 us.  The two launches it replaces take 37 + 34 us: the
  // 1x1 here also multiplied the four halo columns of every 16 then, which the 32x32x2 kernel does not have.)
  const uint32_t x_off = (frame32 * (uint32_t)fin.in2_ctotal + (uint32_t)q) * (uint32_t)ohw + (uint32_t)(oy * fin.ow + ox);
  const uint32_t x_step = 4u * (uint32_t)ohw;
  const float* __restrict__ xin = fin.in2;
  float4 xa[8], xb[8];
  {
    floatx4 br[4];
    rfb_dilated<5>(t.a3.a[2], t.a3.a[2].w, lane, frame32, oy, ox, br);
#pragma unroll
    for (int kc = 0; kc < 8; kc++) xa[kc] = *reinterpret_cast<const float4*>(xin + (x_off + (uint32_t)kc * x_step));
    __builtin_amdgcn_sched_barrier(0);
    fold_branch(2, br);
  }
#pragma unroll
  for (int kc = 0; kc < 8; kc++) xb[kc] = *reinterpret_cast<const float4*>(xin + (x_off + (uint32_t)(8 + kc) * x_step));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int half = 0; half < 2; half++) {
#pragma unroll
    for (int kc = 0; kc < 8; kc++) {
      const float4 xv = half == 0 ? xa[kc] : xb[kc];
      const float xp[4] = {xv.x, xv.y, xv.z, xv.w};
      const int chunk = 12 + half * 8 + kc;
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const float w = wl[(chunk * 4 + m) * 64];
#pragma unroll
        for (int p = 0; p < 4; p++) out[m][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, xp[p], out[m][p], 0, 0, 0);
      }
    }
  }
  if (!live) return;
  char* optr = reinterpret_cast<char*>(fin.out);
  const uint32_t step = 4u * (uint32_t)ohw;
  uint32_t o = 4u * (__umul24(__umul24(frame32, (uint32_t)fin.out_ctotal) + (uint32_t)(fin.out_coff + 4 * q), (uint32_t)ohw) + (uint32_t)(oy * fin.ow + ox));
  const int lowest = fin.relu ? 0 : (int)0x80000000;
#pragma unroll
  for (int m = 0; m < 4; m++) {
    uint32_t om = o + (uint32_t)(16 * m) * step;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float4 v = make_float4(__int_as_float(max(__float_as_int(out[m][0][r]), lowest)), __int_as_float(max(__float_as_int(out[m][1][r]), lowest)),
                                   __int_as_float(max(__float_as_int(out[m][2][r]), lowest)), __int_as_float(max(__float_as_int(out[m][3][r]), lowest)));
      *reinterpret_cast<float4*>(optr + om) = v;
      om += step;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Direct convolution fallback, one output pixel x COB output channels per thread.
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

// ------------------------------------------------------------------------------------------------
// Stem conv (3 -> 16, 3x3, stride 2) reading the decoder's sample planes directly: the fancy chroma upsampling -- h2v2
// for 4:2:0 frames, h2v1 for 4:2:2 ones (the UVC-MJPG flavour the reference's sender captures, sensors.rs:18-68), chosen
// per FRAME from its descriptor --, the fixed-point YCbCr -> RGB conversion and the (v/255 - mean)/std table of
// k_upsample_norm[_420] run inside the kernel, so the normalised f32 input tensor (3.7 MB per frame, written by one kernel
// and read by the next) never exists.
// Same lane layout and MFMA sequence as k_conv3x3_rows_mfma<2, 1> (quad q = input channel R/G/B,
// lane = group of 4 output pixels = 8 input columns; the left halo column from the previous group),
// same integer formulas as jpeg_kernels.hip, hence bit-identical to the two-kernel path.
// The conversion itself is pixel-parallel (one lane = one pixel pair, all three channels) and reaches
// the MFMA layout through a per-wave LDS buffer: see "Conversion" below.
__global__ __launch_bounds__(256) void k_stem_planes_mfma(StemArgs sa) {
  const ConvArgs& a = sa.a;
  constexpr int HL = 1, NG = 16 - 2 * HL;
  extern __shared__ float s_sh[];
  float* s_w = s_sh;                 // packed weights [1][9][64]
  float* s_lut = s_sh + 9 * 64;      // 3 x 256 normalisation table, then 256 zeros (the "table" of a padding row)
  float* s_x = s_lut + 1024;         // per wave: exchange buffer [4 channels][16 groups][8 columns] of one input row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, j16 = lane & 15;
  // (the tables are copied to LDS further down, BEHIND the wave's descriptor reads and its first two rows' sample loads: three
  // dependent memory round trips in a row -- tables, barrier, descriptor, samples -- were a quarter of a wave's life)
  // Row rolling: a wave owns a band of a.band output rows of its 14 column groups.  Output row oy reads input rows
  // 2oy-1 .. 2oy+1, so consecutive output rows share one input row: its 8 upsampled, colour-converted, normalised
  // pixels stay in registers (2 * band + 1 input rows converted per band instead of 3 * band), and the tables above
  // are loaded once per band.  Groups are numbered over (frame, band, column group).
  const int R = a.band, bands = a.oh / R;
  const int ohw = a.oh * a.ow, gpr = a.ow >> 2, gpb = bands * gpr;
  const int total = a.B * gpb;  // (32-bit index arithmetic: fewer than 2^28 groups in a tensor below 4 GiB)
  const int g = ((int)blockIdx.x * 4 + wave) * NG + j16 - HL;
  const bool inrange = g >= 0 && g < total;
  const bool live = inrange && j16 >= HL && j16 < 16 - HL;
  const uint32_t frame32 = inrange ? (uint32_t)g / (uint32_t)gpb : 0u;
  const int rem = inrange ? g - (int)frame32 * gpb : 0;
  const int band = rem / gpr, ox = (rem - band * gpr) * 4, oy0 = band * R;
  const int ix0 = ox * 2;       // first input column of the lane's group (multiple of 8)
  // ---- Conversion, pixel-parallel (round 3; before, every MFMA lane converted the 8 pixels of ITS channel: the chroma
  // upsampling of a pixel was done three times over and the pad quad converted for nothing).  The wave's 16 groups x 8
  // input columns are 64 pixel pairs: lane L converts pair cp = L & 3 of group cg = L >> 2 -- two luma samples, ONE chroma
  // column of both planes, R, G and B of both pixels -- and leaves the six normalised values in the wave's exchange buffer
  // xb[channel][group][8]; every MFMA lane then reads the 8 values of its (channel, group) and the column to their left
  // back (two 16-byte LDS reads + one; channel 3 of the buffer stays zero for the pad quad).  Same integer formulas as
  // jpeg_kernels.hip, hence bit-identical; ~66 vector instructions per lane and input row instead of ~105.
  // The conversion lane takes its group's coordinates from the MFMA lane that owns the group (lanes 0..15: quad 0).
  const int cg = lane >> 2, cp = lane & 3;
  const bool c_inrange = __shfl((int)inrange, cg) != 0;
  const uint32_t c_frame = (uint32_t)__shfl((int)frame32, cg);
  const int c_oy0 = __shfl(oy0, cg);
  const int c_ix = __shfl(ix0, cg) + 2 * cp;  // the lane's two input columns: c_ix, c_ix + 1
  const int cc = c_ix >> 1;                    // ... and its chroma column
  // (every descriptor field the lane needs is loaded UNCONDITIONALLY into a local first -- c_frame is a valid index for every
  // lane -- and selected afterwards: written as `ok && d.width == ...` / `ok ? d.wblk[0] : 0` each field was a load, a wait
  // and a branch of its own, eleven dependent memory round trips in every wave's prologue)
  const JpegFrameDesc& dcv = sa.descs[c_frame];
  const int d_width = dcv.width, d_height = dcv.height, d_v0 = dcv.v[0], d_wblk0 = dcv.wblk[0], d_wblk1 = dcv.wblk[1];
  const int c_dw = dcv.dw[1], c_dh = dcv.dh[1];
  const uint32_t d_off0 = dcv.plane_off[0], d_off1 = dcv.plane_off[1], d_off2 = dcv.plane_off[2];
  const bool c_frame_ok = c_inrange & (d_width == a.iw) & (d_height == a.ih);  // failed frames: zero input
  const uint8_t* cfp = sa.planes + (size_t)c_frame * sa.plane_stride;
  // (a failed frame's lanes load the first bytes of the plane buffer -- its descriptor may hold anything -- and convert_row
  // turns whatever they read into zeros: the loads themselves stay unconditional, see load_samples)
  const int c_ypitch = c_frame_ok ? d_wblk0 * 8 : 0, c_cpitch = c_frame_ok ? d_wblk1 * 8 : 0;
  const uint8_t* c_y = c_frame_ok ? cfp + d_off0 + c_ix : sa.planes;
  const uint8_t* c_cb = c_frame_ok ? cfp + d_off1 + cc : sa.planes;
  const uint8_t* c_cr = c_frame_ok ? cfp + d_off2 + cc : sa.planes;
  // neighbour chroma columns come from the adjacent lanes; at the image's edges jdsample.c repeats the column itself.
  // (Lane 0's left and lane 63's right neighbour lie outside the wave: they only enter pixels of the two halo groups
  // that nobody reads -- group 0 provides its LAST column, group 15 nothing.)
  const bool has_prev = cc > 0, has_next = cc + 1 <= c_dw - 1;
  // h2v1 (4:2:2: chroma at full height) through the h2v2 arithmetic: near row = far row = the pixel's own chroma row makes
  // the column sums 4c, and jdsample.c's h2v1 outputs (3c + l + 1) >> 2 | (3c + r + 2) >> 2 are (3 * 4c + 4l + 4) >> 4 |
  // (3 * 4c + 4r + 8) >> 4: only the two rounding terms differ from h2v2's 8 | 7.
  const bool c_v2 = d_v0 == 2;
  const int bias_l = c_v2 ? 8 : 4, bias_r = c_v2 ? 7 : 8;
  // (Round 5 tried [channel][columns 0-3 | 4-7][group][4], which makes the MFMA lanes' 16-byte reads conflict-free: the
  // kernel's SQ_LDS_BANK_CONFLICT share went from 106 % to 133 % of its LDS-active cycles and its time did not move, 60.4 ->
  // 60.5 us -- the conflicts that count are the six table look-ups per lane and row below, whose addresses are the pixels.)
  float* xb = s_x + wave * 512;                        // [4][16][8]
  const float* xr = xb + q * 128 + j16 * 8;            // what this MFMA lane reads back
  const int xl = max(q * 128 + j16 * 8 - 1, 0);        // the column left of them (group j16 - 1's last)
  const bool left_ok = j16 > 0 && ix0 > 0;             // else: zero padding at the row start (or a halo lane)
  *reinterpret_cast<float2*>(xb + 384 + 2 * lane) = make_float2(0.f, 0.f);  // channel 3: zeros, written once
  auto dpp_prev = [](int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138 /*wave_shr:1*/, 0xf, 0xf, true); };
  auto dpp_next = [](int x) { return __builtin_amdgcn_update_dpp(0, x, 0x130 /*wave_shl:1*/, 0xf, 0xf, true); };
  // out = y + ((ku * (cb - 128) + kv * (cr - 128) + 32768) >> 16) as ku * cb + kv * cr + kc on the 24-bit multiplier
  // (v_mad_i32_i24, full rate): the same integer, |ku|, |kv| < 2^17 and cb, cr < 2^8
  constexpr int kRv = 91881, kGu = -22554, kGv = -46802, kBu = 116130;
  constexpr int kRc = 32768 - 128 * kRv, kGc = 32768 - 128 * (kGu + kGv), kBc = 32768 - 128 * kBu;

  // input row 2 * oy0 + k of the wave's groups: v[0..7] = columns ix0 .. ix0+7 of the lane's channel, v[8] = column ix0 - 1
  // (0 at the row start); a row outside the image is zero padding
  // (the five sample loads of a row are issued ONE ROW AHEAD of their conversion -- load_samples / convert_row -- into a
  // register ping-pong: issued where they are used, every row of a wave's band waited a memory round trip, ~1.5 us x 9 rows of
  // a 17 us wave for 3.3 us of issue work)
  struct Samples {
    uint32_t yy, cbn, cbf, crn, crf;
  };
  auto load_samples = [&](int k, Samples& sm) {
    const int yc = min(max(2 * c_oy0 + k, 0), a.ih - 1);
    const int cy = c_v2 ? yc >> 1 : yc;  // h2v2 fancy upsampling: near / far row; h2v1: both the pixel's own
    const int ny = c_v2 ? max(0, min(c_dh - 1, (yc & 1) ? cy + 1 : cy - 1)) : cy;
    // (no branch around the loads: behind one the compiler cannot count what is in flight and waits for ALL of it -- the
    // row just requested included -- in front of the previous row's conversion)
    const uint32_t ro = (uint32_t)__mul24(cy, c_cpitch), fo = (uint32_t)__mul24(ny, c_cpitch);
    sm.yy = *reinterpret_cast<const uint16_t*>(c_y + (uint32_t)__mul24(yc, c_ypitch));
    sm.cbn = c_cb[ro], sm.cbf = c_cb[fo], sm.crn = c_cr[ro], sm.crf = c_cr[fo];
  };
  auto convert_row = [&](int k, const Samples& sm, float (&v)[9]) {
    const int iy = 2 * c_oy0 + k;
    const bool ok = c_frame_ok && iy >= 0 && iy < a.ih;
    const uint32_t yy = sm.yy, cbn = sm.cbn, cbf = sm.cbf, crn = sm.crn, crf = sm.crf;
    const int scb = mad24(3, (int)cbn, (int)cbf), scr = mad24(3, (int)crn, (int)crf);  // column sums 3 * near + far
    const int pcb = dpp_prev(scb), ncb = dpp_next(scb), pcr = dpp_prev(scr), ncr = dpp_next(scr);
    const int lcb = has_prev ? pcb : scb, rcb = has_next ? ncb : scb, lcr = has_prev ? pcr : scr, rcr = has_next ? ncr : scr;
    const int cb0 = (mad24(scb, 3, lcb) + bias_l) >> 4, cb1 = (mad24(scb, 3, rcb) + bias_r) >> 4;
    const int cr0 = (mad24(scr, 3, lcr) + bias_l) >> 4, cr1 = (mad24(scr, 3, rcr) + bias_r) >> 4;
    const int y0 = (int)(yy & 255u), y1 = (int)(yy >> 8);
    auto clamp255 = [](int x) { return min(255, max(0, x)); };
    const int r0 = clamp255(y0 + (mad24(kRv, cr0, kRc) >> 16)), r1 = clamp255(y1 + (mad24(kRv, cr1, kRc) >> 16));
    const int g0 = clamp255(y0 + (mad24(kGv, cr0, mad24(kGu, cb0, kGc)) >> 16)), g1 = clamp255(y1 + (mad24(kGv, cr1, mad24(kGu, cb1, kGc)) >> 16));
    const int b0 = clamp255(y0 + (mad24(kBu, cb0, kBc) >> 16)), b1 = clamp255(y1 + (mad24(kBu, cb1, kBc) >> 16));
    // (a padding row reads the table's row of zeros: two selects per row instead of a branch around the six reads)
    const float* l0 = s_lut + (ok ? 0 : 768);
    const int lstep = ok ? 256 : 0;
    float* xw = xb + cg * 8 + 2 * cp;
    *reinterpret_cast<float2*>(xw) = make_float2(l0[r0], l0[r1]);
    *reinterpret_cast<float2*>(xw + 128) = make_float2(l0[lstep + g0], l0[lstep + g1]);
    *reinterpret_cast<float2*>(xw + 256) = make_float2(l0[2 * lstep + b0], l0[2 * lstep + b1]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the wave's stores before the wave's loads (in-order LDS)
    const float4 va = *reinterpret_cast<const float4*>(xr), vb = *reinterpret_cast<const float4*>(xr + 4);
    const float left = xb[xl];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // ... and these loads before the next row's stores
    v[0] = va.x, v[1] = va.y, v[2] = va.z, v[3] = va.w, v[4] = vb.x, v[5] = vb.y, v[6] = vb.z, v[7] = vb.w;
    v[8] = left_ok ? left : 0.0f;
  };
  floatx4 acc[4];
  // tap row r of the 3x3 kernel applied to a converted input row: output pixel j reads columns 2j-1, 2j, 2j+1
  auto mac_row = [&](int r, const float (&v)[9]) {
    const float x[3][4] = {{v[8], v[1], v[3], v[5]}, {v[0], v[2], v[4], v[6]}, {v[1], v[3], v[5], v[7]}};
#pragma unroll
    for (int kx = 0; kx < 3; kx++) {
      const float w = s_w[(r * 3 + kx) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x[kx][j], acc[j], 0, 0, 0);
    }
  };
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {  // (unconditional loads from a clamped index: no branch, no wait per element)
    const float b = a.bias[min(4 * q + r, a.cout - 1)];
    bias[r] = 4 * q + r < a.cout ? b : 0.0f;
  }

  // stores: 32-bit byte offset of (frame, channel 4q, row oy0, column ox) from the tensor base, a channel step per store;
  // ReLU as the integer maximum with 0 (relu_acc), or with INT_MIN = the identity
  char* out = reinterpret_cast<char*>(a.out);
  const int cout = a.cout;
  const uint32_t out_step = 4u * (uint32_t)ohw;
  const uint32_t out_off = 4u * (__umul24(__umul24(frame32, (uint32_t)a.out_ctotal) + (uint32_t)(a.out_coff + 4 * q), (uint32_t)ohw) + (uint32_t)(oy0 * a.ow + ox));
  const int lowest = a.relu ? 0 : (int)0x80000000;

  float top[9], mid[9], bot[9];
  Samples even, odd;  // the samples of the next even / odd input row of the band
  load_samples(-1, odd);
  load_samples(0, even);
  {  // 144 + 192 16-byte pieces, at most one of each per thread, both loads in flight together (as `for (i = tid; ...) s[i] =
    // g[i]` loops these were six load - wait - store round trips one after the other)
    const int tid = threadIdx.x;
    const float4 wv = reinterpret_cast<const float4*>(a.w)[min(tid, 143)];
    const float4 lv = reinterpret_cast<const float4*>(sa.lut)[min(tid, 191)];
    if (tid < 144) reinterpret_cast<float4*>(s_w)[tid] = wv;
    if (tid < 192) reinterpret_cast<float4*>(s_lut)[tid] = lv;
    s_lut[768 + tid] = 0.0f;
  }
  __syncthreads();
  convert_row(-1, odd, top);
  load_samples(1, odd);
#pragma unroll 1
  for (int i = 0; i < R; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[j][r] = bias[r];
    mac_row(0, top);
    convert_row(2 * i, even, mid);
    load_samples(2 * i + 2, even);  // (past the band's last row: a clamped, valid address, never converted)
    __builtin_amdgcn_sched_barrier(0);  // (the odd row's samples are not touched before here: hoisted, their first use waits)
    mac_row(1, mid);
    convert_row(2 * i + 1, odd, bot);
    load_samples(2 * i + 3, odd);
    __builtin_amdgcn_sched_barrier(0);
    mac_row(2, bot);
    if (live) {
      uint32_t o = out_off + 4u * (uint32_t)(i * a.ow);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (4 * q + r < cout) {
          const float4 v4 = make_float4(__int_as_float(max(__float_as_int(acc[0][r]), lowest)), __int_as_float(max(__float_as_int(acc[1][r]), lowest)),
                                        __int_as_float(max(__float_as_int(acc[2][r]), lowest)), __int_as_float(max(__float_as_int(acc[3][r]), lowest)));
          *reinterpret_cast<float4*>(out + o) = v4;
        }
        o += out_step;
      }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) top[k] = bot[k];  // input row 2oy + 1 = 2(oy + 1) - 1
  }
}

}  // namespace

// ---------------------------------------------------------------- host side
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

// [c][12] records (9 taps ky-major, bias, 2 pad), then the two derived tables of the chained kernel (k_dwpw2_mfma copies them
// to LDS as they are: built per block from the records they cost every block ~1500 dependent scalar loads in its prologue):
//   edge taps [4][c][8] = (w00 w10 w20 0 | w02 w12 w22 0), copies 0 .. 2 as dw_variant (tap row 0 / 2 zeroed for a window over
//   the top / bottom border), copy 3 all zero;
//   fold records [2][c / 2][32]: per channel pair (i, i + 1) and tap row k eight floats (wk0' wk0' | wk0 wk0 | wk1 wk1 | wk2 wk2),
//   then (bias bias 0 ...); wk0' = wk0 in copy 0, 0 in copy 1 (lanes at the left image border).
size_t depthwise_packed_floats(int c) { return (size_t)c * (12 + 32 + 32); }

void pack_depthwise_weights(const float* w, const float* bias, int c, float* packed) {
  for (int i = 0; i < c; i++) {
    for (int t = 0; t < 9; t++) packed[i * 12 + t] = w[i * 9 + t];
    packed[i * 12 + 9] = bias[i];
    packed[i * 12 + 10] = packed[i * 12 + 11] = 0.0f;
  }
  float* e = packed + (size_t)c * 12;
  for (int v = 0; v < 4; v++)
    for (int i = 0; i < c; i++)
      for (int j = 0; j < 8; j++) {
        const int k = j & 3;
        const bool zero = v == 3 || k == 3 || (v == 1 && k == 0) || (v == 2 && k == 2);
        e[((size_t)v * c + i) * 8 + j] = zero ? 0.0f : w[i * 9 + 3 * k + (j >> 2) * 2];
      }
  float* f = e + (size_t)4 * c * 8;
  for (int v = 0; v < 2; v++)
    for (int pr = 0; pr < c / 2; pr++)
      for (int idx = 0; idx < 32; idx++) {
        const int i = 2 * pr + (idx & 1), k = idx >> 3, el = (idx >> 1) & 3;
        float x;
        if (k < 3) x = el == 0 ? (v ? 0.0f : w[i * 9 + 3 * k]) : w[i * 9 + 3 * k + el - 1];
        else x = el == 0 ? bias[i] : 0.0f;
        f[((size_t)v * (c / 2) + pr) * 32 + idx] = x;
      }
}

size_t conv3x3_packed_floats(int cin) { return (size_t)cin * 3 * 64; }

void pack_conv3x3_weights(const float* w, int cin, int cout, float* packed) {
  for (int ci = 0; ci < cin; ci++)
    for (int s = 0; s < 3; s++)
      for (int lane = 0; lane < 64; lane++) {
        const int co = lane & 15, t = 4 * s + (lane >> 4);
        packed[((size_t)ci * 3 + s) * 64 + lane] = (co < cout && t < 9) ? w[((size_t)co * cin + ci) * 9 + t] : 0.0f;
      }
}

size_t conv3x3_rows_packed_floats(int cin) { return (size_t)((cin + 3) / 4) * 9 * 64; }

void pack_conv3x3_rows_weights(const float* w, int cin, int cout, float* packed) {
  const int cin4 = (cin + 3) / 4;
  for (int kc = 0; kc < cin4; kc++)
    for (int t = 0; t < 9; t++)
      for (int lane = 0; lane < 64; lane++) {
        const int co = lane & 15, ci = 4 * kc + (lane >> 4);
        packed[((size_t)kc * 9 + t) * 64 + lane] = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * 9 + t] : 0.0f;
      }
}

bool conv3x3_rows_supported(const ConvArgs& a) {
  if (a.k != 3 || a.cout > 16 || a.depthwise || a.res) return false;
  if ((a.ow & 3) || (a.iw & 3) || a.pad != a.dil) return false;
  if (a.stride == 1) return (a.dil == 1 || a.dil == 2 || a.dil == 3 || a.dil == 5) && a.iw == a.ow && a.ih == a.oh;
  return a.stride == 2 && a.dil == 1 && a.iw == 2 * a.ow;
}

bool stem_planes_supported(const ConvArgs& a) {
  return a.k == 3 && a.stride == 2 && a.dil == 1 && a.pad == 1 && a.cin == 3 && a.cout <= 16 && !a.depthwise && !a.res &&
         a.iw == 2 * a.ow && a.ih == 2 * a.oh && (a.ow & 3) == 0 && (a.iw & 7) == 0 &&
         (reinterpret_cast<uintptr_t>(a.w) & 15) == 0;  // (the kernel copies the packed weights in 16-byte pieces)
}

void launch_stem_planes_mfma(const StemArgs& sa0, hipStream_t s) {
  StemArgs sa = sa0;
  ConvArgs& a = sa.a;
  // rows per band: the largest divisor of the output height <= 8 that still gives the GPU a full round of waves (8 per
  // SIMD).  Measured at batch 32 of the 640 model (rocprofv3, alone): band 1 85 us, 2 72, 3 69, 4 67, 6 68, 8 73, 12 68.
  int band = 1;
  for (int r = 2; r <= 8; r++)
    if (a.oh % r == 0 && (long)a.B * (a.oh / r) * (a.ow / 4) / 14 >= 8192) band = r;
  a.band = band;
  const long groups = (long)a.B * ((a.oh / band) * a.ow / 4);
  const size_t shmem = (9 * 64 + 768 + 256 + 4 * 512) * sizeof(float);  // weights, normalisation table, a row of zeros, exchange buffers
  ufd_launch(k_stem_planes_mfma, dim3((unsigned)((groups + 4L * 14 - 1) / (4L * 14))), dim3(256), shmem, s, sa);
}

void launch_conv3x3_rows_mfma(const ConvArgs* args, int n, hipStream_t s) {
  ConvArgs3 p{};
  const ConvArgs& a = args[0];
  const long groups = (long)a.B * (a.oh * a.ow / 4);
  const size_t wbytes = conv3x3_rows_packed_floats(a.cin) * sizeof(float);
  const bool lds = wbytes <= 40 * 1024;
  for (int i = 0; i < n; i++) {
    p.a[i] = args[i];
    p.a[i].dbg = lds ? 1 : 0;  // weights staged in LDS
  }
  const size_t shmem = lds ? wbytes : 0;
  // (a multiple of 8 blocks along x: the kernel gives every XCD a contiguous range of them)
  auto grid = [&](int hl) { return dim3((unsigned)(((groups + 4L * (16 - 2 * hl) - 1) / (4L * (16 - 2 * hl)) + 7) / 8 * 8), (unsigned)n); };
  bool mixed = false;
  for (int i = 1; i < n; i++) mixed = mixed || args[i].dil != a.dil;
  if (mixed) {  // (stride 1, dilations <= 8: conv3x3_rows_supported)
    ufd_launch((k_conv3x3_rows_mfma<1, 0>), grid(2), dim3(256), shmem, s, p);
    return;
  }
  if (a.stride == 2) {
    ufd_launch((k_conv3x3_rows_mfma<2, 1>), grid(1), dim3(256), shmem, s, p);
  } else if (a.dil == 1) {
    ufd_launch((k_conv3x3_rows_mfma<1, 1>), grid(1), dim3(256), shmem, s, p);
  } else if (a.dil == 2) {
    ufd_launch((k_conv3x3_rows_mfma<1, 2>), grid(1), dim3(256), shmem, s, p);
  } else if (a.dil == 3) {
    ufd_launch((k_conv3x3_rows_mfma<1, 3>), grid(1), dim3(256), shmem, s, p);
  } else {
    ufd_launch((k_conv3x3_rows_mfma<1, 5>), grid(2), dim3(256), shmem, s, p);
  }
}

// Template instance launch_conv3x3_rows_mfma picks (profile labels: bench.py matches them with rocprofv3's kernel names).
const char* conv3x3_rows_instance(const ConvArgs* args, int n) {
  const ConvArgs& a = args[0];
  for (int i = 1; i < n; i++)
    if (args[i].dil != a.dil) return "<1, 0>";
  if (a.stride == 2) return "<2, 1>";
  return a.dil == 1 ? "<1, 1>" : (a.dil == 2 ? "<1, 2>" : (a.dil == 3 ? "<1, 3>" : "<1, 5>"));
}

// ---- RFB tail (k_rfb_tail)
size_t rfb_tail_packed_floats() { return (size_t)kTailChunks * 4 * 64; }

// w_lin [64][48] (ConvLinear over the concat: branch b's channels at 16 b), w_short [64][64] (shortcut over x):
// packed[(chunk * 4 + m) * 64 + lane] = W[cout 16 m + (lane & 15)][channel of (chunk, k = lane >> 4)], where the branch chunks
// follow the accumulator layout of the dilated convs (chunk 4 b + r holds channels 16 b + 4 k + r) and x's are 4 chunk' + k.
void pack_rfb_tail_weights(const float* w_lin, const float* w_short, float* packed) {
  for (int chunk = 0; chunk < kTailChunks; chunk++)
    for (int m = 0; m < 4; m++)
      for (int lane = 0; lane < 64; lane++) {
        const int co = 16 * m + (lane & 15), k = lane >> 4;
        float v;
        if (chunk < 12) {
          const int b = chunk >> 2, r = chunk & 3;
          v = w_lin[(size_t)co * 48 + 16 * b + 4 * k + r];
        } else {
          v = w_short[(size_t)co * 64 + 4 * (chunk - 12) + k];
        }
        packed[((size_t)chunk * 4 + m) * 64 + lane] = v;
      }
}

// dil3[0..2]: the dilated convs in concat order (dilations 2, 3, 5; 16 -> 16, stride 1, same maps); fin: the summed 1x1
// (cin = 48 + 64, cout 64, in2 = x).
bool rfb_tail_supported(const ConvArgs* dil3, const ConvArgs& fin) {
  static const int kDil[3] = {2, 3, 5};
  for (int b = 0; b < 3; b++) {
    const ConvArgs& a = dil3[b];
    if (a.k != 3 || a.stride != 1 || a.dil != kDil[b] || a.pad != a.dil || a.cin != 16 || a.cout != 16 || a.depthwise || a.res || a.relu) return false;
    if (a.ih != fin.oh || a.iw != fin.ow || a.oh != fin.oh || a.ow != fin.ow) return false;
  }
  // ow % 8: lanes 0 / 1 and 14 / 15 of a quad fetch each other's outer neighbours (rfb_dilated, load_edge), so a pair of pixel
  // groups must never straddle a row or a frame: an even number of groups per row (20 for the 640 model, 10 for the 320 one)
  return fin.k == 1 && fin.cout == 64 && fin.cin == 48 + 64 && fin.in2_ctotal >= 64 && (fin.ow & 7) == 0 && !fin.res;  // (shapes only: also asked at plan time)
}

void launch_rfb_tail(const ConvArgs* dil3, const ConvArgs& fin, hipStream_t s) {
  RfbTailArgs t{};
  for (int b = 0; b < 3; b++) t.a3.a[b] = dil3[b];
  t.fin = fin;
  const long groups = (long)fin.B * (fin.oh * fin.ow / 4);
  const unsigned blocks = (unsigned)(((groups + 4L * kTailNG - 1) / (4L * kTailNG) + 7) / 8 * 8);
  const size_t lds = rfb_tail_packed_floats() * sizeof(float);  // 28 KB
  ufd_launch(k_rfb_tail, dim3(blocks), dim3(256), lds, s, t);
}

bool dwpw_supported(const ConvArgs& a, int stride) {
  return (a.ow % 4 == 0) && (a.iw % 4 == 0) && (a.cin % 2 == 0) && (stride == 1 || stride == 2) &&
         (stride == 1 ? (a.iw == a.ow && a.ih == a.oh) : (a.iw == 2 * a.ow));
}

// One 32-cout tile per wave (64 accumulator registers, 3 waves/SIMD) measured faster than two
// tiles (128 accumulators, 1 wave/SIMD) on every layer of this network at batch 32, so the
// launchers use CT = 1: activations of wider layers are re-read per cout tile from L2.
// Split-K when a launch would have fewer wave tiles than ~2 per SIMD.
static bool want_splitk(long wave_tiles, int cts, int ksteps) {
  if (ksteps % 16 != 0) return false;
  // measured on MI355X at batch 32: pays when the launch has fewer than ~6 waves per k-step of
  // chain length (15x20 maps, 64->4/8 heads at 30x40, the 256-channel layers); costs otherwise
  return wave_tiles * cts < 6L * ksteps;
}
// The 1x1 kernel also splits chains that are a multiple of 8 k-steps (the RFB's summed conv: 48 + 64 channels = 56 k-steps,
// 23 us for ONE frame unsplit): k_pw_mfma<1, 2, 4>.
static bool pw_wants_splitk(long wave_tiles, int cts, int ksteps) {
  if (ksteps % 8 != 0) return false;
  return wave_tiles * cts < 6L * ksteps;
}
constexpr size_t kMaxLdsBytes = 160 * 1024;  // LDS of a gfx950 CU
// Kernels that ask for more than the default 64 KB of dynamic LDS: raised once per kernel and host thread.
static void allow_large_lds(const void* kernel) {
  thread_local std::vector<const void*> done;
  for (const void* k : done)
    if (k == kernel) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes);
  done.push_back(kernel);
}
constexpr size_t kSplitKBytes = 64 * 64 * sizeof(float);  // CT = 1: one partial tile at a time

// Launch configuration of k_pw_mfma for n merged convs.
struct PwConfig {
  ConvArgs3 p;
  unsigned gx, gy;
  size_t lds;
  bool sk;
  int ksteps;
};
static PwConfig pw_config(const ConvArgs* args, int n) {
  PwConfig c{};
  const ConvArgs& r = args[0];
  const long groups = (long)r.B * (r.oh * r.ow / 4);
  const long wave_tiles = (groups + 31) / 32;
  c.ksteps = r.cin >> 1;
  int max_cts = 1;
  for (int i = 0; i < n; i++) max_cts = std::max(max_cts, (args[i].cout + 31) / 32);
  c.sk = pw_wants_splitk(wave_tiles, max_cts * n, c.ksteps);
  c.lds = ((size_t)kBiasLds + (size_t)c.ksteps * 64) * sizeof(float) + (c.sk ? kSplitKBytes : 0);
  unsigned grid = 1;
  for (int i = 0; i < n; i++) {
    c.p.a[i] = args[i];
    c.p.a[i].cts = (args[i].cout + 31) / 32;
    c.p.a[i].tiles = c.sk ? (int)wave_tiles : (int)((groups + 127) / 128);
    grid = std::max(grid, (unsigned)((c.p.a[i].tiles + 7) / 8) * 8 * c.p.a[i].cts);
  }
  c.gx = grid, c.gy = (unsigned)n;
  return c;
}

// n (<= 3) convolutions with identical shapes except cout / weights / outputs
void launch_conv_pointwise_mfma(const ConvArgs* args, int n, hipStream_t s) {
  const PwConfig c = pw_config(args, n);
  if (c.sk && c.ksteps % 16 != 0)  // (a quarter of the chain is not a multiple of four k-steps: ring depth 2)
    ufd_launch((k_pw_mfma<1, 2, 4>), dim3(c.gx, c.gy), dim3(256), c.lds, s, c.p);
  else if (c.sk)
    ufd_launch((k_pw_mfma<1, 4, 4>), dim3(c.gx, c.gy), dim3(256), c.lds, s, c.p);
  else if (c.ksteps % 4 == 0)
    ufd_launch((k_pw_mfma<1, 4, 1>), dim3(c.gx, c.gy), dim3(256), c.lds, s, c.p);
  else
    ufd_launch((k_pw_mfma<1, 1, 1>), dim3(c.gx, c.gy), dim3(256), c.lds, s, c.p);
}

// Template instance the launchers above / below pick for a launch, as rocprofv3 prints it behind the kernel name
// (bench.py reports the roofline per instance and checks it against profiles/<round>/kernel_stats.csv).
const char* conv_pointwise_instance(const ConvArgs* args, int n) {
  const ConvArgs& r = args[0];
  const long groups = (long)r.B * (r.oh * r.ow / 4);
  const int ksteps = r.cin >> 1;
  int max_cts = 1;
  for (int i = 0; i < n; i++) max_cts = std::max(max_cts, (args[i].cout + 31) / 32);
  if (pw_wants_splitk((groups + 31) / 32, max_cts * n, ksteps)) return ksteps % 16 ? "<1, 2, 4>" : "<1, 4, 4>";
  return ksteps % 4 == 0 ? "<1, 4, 1>" : "<1, 1, 1>";
}

bool dwpw_uses_coop(const ConvArgs* args, int n) {
  // (4 cout tiles per block; the 2-tile form measured slower than k_dwpw_mfma)
  const int cts = (args[0].cout + 31) / 32, ksteps = args[0].cin >> 1;
  if (!(n == 1 && ksteps % 8 == 0 && cts % 4 == 0)) return false;
  // A frame or a few at a time (the reference's operating point: one stream, one frame, inferer.rs:23,29-50): the launch
  // has a handful of pixel tiles, and a cooperative wave walks the WHOLE channel chain of its tile -- the kernel then
  // lasts as long for one frame as for 32 (k_dwpw_coop<1, 4> on the 30x40 maps: 16-28 us at batch 1, 29 us at batch 32).
  // Split-K waves of k_dwpw_mfma walk a quarter of it each: the one to use while the cooperative launch would have fewer
  // than 256 waves (64 blocks, a quarter of the CUs) -- NOT wherever that kernel would split: m12 at batch 32 (15x20 maps,
  // 80 pixel tiles x 8 cout tiles) is inside want_splitk's range and takes 55 us split, 29 us cooperative.
  const long groups = (long)args[0].B * (args[0].oh * args[0].ow / 4);
  const long wave_tiles = (groups + kDwGroups - 1) / kDwGroups;
  return !(wave_tiles * cts < 256 && want_splitk(wave_tiles, cts, ksteps));
}

// Launch configuration of k_dwpw_mfma for n merged convs (the non-cooperative form): arguments with tiles / cts set,
// grid, dynamic LDS, split-K or not.
struct DwpwConfig {
  ConvArgs3 p;
  unsigned gx, gy;
  size_t lds;
  bool sk, deep;
  bool sk8;  // the split over eight waves (k_dwpw_mfma_sk8): one conv, at most kSk8MaxBlocks blocks
};
constexpr long kSk8MaxBlocks = 128;  // half the CUs: beyond that the 64 KB of reduction LDS per block starts to cost other work its place
// Measured, one 640x480 frame (profiles/r6c/batch1_timeline_640.txt against r5d's): chains of 64 and 128 k-steps gain -- m9 / m10
// 12.1 -> 10.7 us, m12 18.3 -> 15.4 -- chains of 32 (m5, m6: 9.0 -> 9.3) and the stride-2 blocks (m8 9.4 -> 10.0, m11 12.7 -> 13.3)
// do not: below ~8 k-steps per wave a launch is its 4.6 us floor, its table copy and the reduction.  Hence: stride 1, >= 64 k-steps.
static bool want_sk8(bool sk, int n, long blocks, int ksteps, int stride) {
  return sk && n == 1 && stride == 1 && ksteps >= 64 && blocks <= kSk8MaxBlocks;
}
static DwpwConfig dwpw_config(const ConvArgs* args, int n, int sk8_stride = 0 /* the depthwise stride of a launch that may take the eight-wave split; 0: it may not */) {
  DwpwConfig c{};
  const ConvArgs& r = args[0];
  const long groups = (long)r.B * (r.oh * r.ow / 4);
  const long wave_tiles = (groups + kDwGroups - 1) / kDwGroups;
  const int ksteps = r.cin >> 1;
  int max_cts = 1;
  for (int i = 0; i < n; i++) max_cts = std::max(max_cts, (args[i].cout + 31) / 32);
  c.sk = want_splitk(wave_tiles, max_cts * n, ksteps);
  c.sk8 = sk8_stride > 0 && want_sk8(c.sk, n, wave_tiles * max_cts, ksteps, sk8_stride);
  c.deep = ksteps % 4 == 0;
  c.lds = ((size_t)kBiasLds + (size_t)r.cin * 36 + (size_t)ksteps * 64) * sizeof(float) + (c.sk8 ? 4 * kSplitKBytes : (c.sk ? kSplitKBytes : 0));
  unsigned grid = 1;
  for (int i = 0; i < n; i++) {
    c.p.a[i] = args[i];
    c.p.a[i].cts = (args[i].cout + 31) / 32;
    c.p.a[i].tiles = c.sk ? (int)wave_tiles : (int)((wave_tiles + 3) / 4);
    grid = std::max(grid, (unsigned)((c.p.a[i].tiles + 7) / 8) * 8 * c.p.a[i].cts);
  }
  c.gx = grid, c.gy = (unsigned)n;
  return c;
}
// ... of k_dwpw_coop<S, 4> for one conv with a multiple of four cout tiles
struct CoopConfig {
  ConvArgs3 p;
  unsigned blocks;
  size_t lds;
  int ctw;
};
static CoopConfig coop_config(const ConvArgs* args) {
  CoopConfig c{};
  const ConvArgs& r = args[0];
  const long groups = (long)r.B * (r.oh * r.ow / 4);
  const long wave_tiles = (groups + kDwGroups - 1) / kDwGroups;
  const int max_cts = (r.cout + 31) / 32;
  c.ctw = (max_cts % 4 == 0) ? 4 : 2;
  const int ptiles = 4 / c.ctw;
  c.p.a[0] = r;
  c.p.a[0].cts = max_cts;
  c.p.a[0].tiles = (int)((wave_tiles + ptiles - 1) / ptiles);  // pixel-tile groups
  c.blocks = (unsigned)((c.p.a[0].tiles + 7) / 8) * 8 * (max_cts / c.ctw);
  c.lds = ((size_t)c.ctw * kBiasLds + (size_t)r.cin * 36) * sizeof(float) + (size_t)ptiles * 3 * 8 * 64 * sizeof(float4);
  return c;
}

void launch_conv_dwpw_mfma(const ConvArgs* args, int n, int stride, hipStream_t s) {
  // several cout tiles: the cooperative kernel shares the depthwise work among the waves of a block
  if (dwpw_uses_coop(args, n)) {
    const CoopConfig c = coop_config(args);
    void (*kernel)(ConvArgs3) = stride == 1 ? (c.ctw == 4 ? k_dwpw_coop<1, 4> : k_dwpw_coop<1, 2>)
                                             : (c.ctw == 4 ? k_dwpw_coop<2, 4> : k_dwpw_coop<2, 2>);
    if (c.lds > 64 * 1024) allow_large_lds(reinterpret_cast<const void*>(kernel));
    ufd_launch(kernel, dim3(c.blocks, 1), dim3(256), c.lds, s, c.p);
    return;
  }
  const DwpwConfig c = dwpw_config(args, n, stride);
  const dim3 g(c.gx, c.gy);
  // (the three copies of the depthwise table of a 256-channel layer need more than the default
  // 64 KB of dynamic LDS: raised once per instantiation)
  auto launch = [&](void (*kernel)(ConvArgs3)) {
    if (c.lds > 64 * 1024) allow_large_lds(reinterpret_cast<const void*>(kernel));
    ufd_launch(kernel, g, dim3(256), c.lds, s, c.p);
  };
  if (c.sk8) {
    void (*kernel)(ConvArgs3) = k_dwpw_mfma_sk8<1>;
    if (c.lds > 64 * 1024) allow_large_lds(reinterpret_cast<const void*>(kernel));
    ufd_launch(kernel, dim3(c.gx, 1), dim3(512), c.lds, s, c.p);
    return;
  }
  if (c.sk) {
    if (stride == 1) launch(k_dwpw_mfma<1, 1, 2, 4>);
    else launch(k_dwpw_mfma<1, 2, 2, 4>);
    return;
  }
  if (stride == 1) {
    // (four k-steps of windows in flight instead of two, round 5: 31.1 -> 30.6 us alone, frame rate unchanged; again with the
    // requests pinned -- twelve loads in flight in the ISA, where hipcc had left six --: 29.6 -> 29.1 us, the dual launches
    // that share the body 35.1 -> 36.4: these layers do not wait for their windows.  Not kept.)
    if (c.deep) launch(k_dwpw_mfma<1, 1, 2, 1>);
    else launch(k_dwpw_mfma<1, 1, 1, 1>);
  } else {
    if (c.deep) launch(k_dwpw_mfma<1, 2, 2, 1>);
    else launch(k_dwpw_mfma<1, 2, 1, 1>);
  }
}

// Dual launch (k_dual_*): conv group A = a merged cls/reg head pair on k_dwpw_mfma<1, 1, 2, SK>, conv B = one dw->pw block
// on k_dwpw_coop<S, 4> (b_stride > 0) or one 1x1 conv on k_pw_mfma<1, 4, SK> (b_stride == 0).  False: this combination
// of instances is not compiled -- the caller issues the two launches one after the other.
namespace {
struct DualChoice {
  void (*kernel)(DualArgs) = nullptr;
  const char* label = nullptr;  // profile label: device function (as bench.py's KERNEL_FUNCS maps it) + template instance
  DualArgs q{};
  unsigned blocks = 0;
  size_t lds = 0;
};
bool choose_dual(const ConvArgs* a, int na, int a_stride, const ConvArgs* b, int b_stride, DualChoice* out) {
  if (experiment_env("UFD_NO_DUAL")) return false;
  if (na < 1 || na > 3 || a_stride != 1 || dwpw_uses_coop(a, na)) return false;
  const DwpwConfig ca = dwpw_config(a, na);
  if (!ca.deep) return false;
  DualChoice& c = *out;
  c.q.a = ca.p;
  c.q.ax = (int)ca.gx, c.q.ay = (int)ca.gy;
  const unsigned blocks_a = ca.gx * ca.gy;
  c.lds = ca.lds;
  unsigned blocks_b = 0;
  if (b_stride > 0) {
    if (!dwpw_uses_coop(b, 1)) return false;
    const CoopConfig cb = coop_config(b);
    if (cb.ctw != 4) return false;
    c.q.b = cb.p;
    blocks_b = cb.blocks;
    c.lds = std::max(c.lds, cb.lds);
    if (b_stride == 2) {
      c.kernel = ca.sk ? k_dual_dwpw_coop<1, 4, 2> : k_dual_dwpw_coop<1, 1, 2>;
      c.label = ca.sk ? "conv_dual_coop<1, 4, 2>" : "conv_dual_coop<1, 1, 2>";
    } else {
      c.kernel = ca.sk ? k_dual_dwpw_coop<1, 4, 1> : k_dual_dwpw_coop<1, 1, 1>;
      c.label = ca.sk ? "conv_dual_coop<1, 4, 1>" : "conv_dual_coop<1, 1, 1>";
    }
  } else {
    if (b[0].in2 || b[0].res) return false;
    const PwConfig cb = pw_config(b, 1);
    if (cb.ksteps % 4 != 0 || (cb.sk && cb.ksteps % 16 != 0)) return false;  // (the dual instances are <1, 4, SK>)
    c.q.b = cb.p;
    blocks_b = cb.gx;
    c.lds = std::max(c.lds, cb.lds);
    if (cb.sk) {
      c.kernel = ca.sk ? k_dual_dwpw_pw<1, 4, 4> : k_dual_dwpw_pw<1, 1, 4>;
      c.label = ca.sk ? "conv_dual_pw<1, 4, 4>" : "conv_dual_pw<1, 1, 4>";
    } else {
      c.kernel = ca.sk ? k_dual_dwpw_pw<1, 4, 1> : k_dual_dwpw_pw<1, 1, 1>;
      c.label = ca.sk ? "conv_dual_pw<1, 4, 1>" : "conv_dual_pw<1, 1, 1>";
    }
  }
  c.blocks = blocks_a + blocks_b;
  return true;
}
}  // namespace

bool launch_conv_dual(const ConvArgs* a, int na, int a_stride, const ConvArgs* b, int b_stride, hipStream_t s) {
  DualChoice c;
  if (!choose_dual(a, na, a_stride, b, b_stride, &c)) return false;
  if (c.lds > 64 * 1024) allow_large_lds(reinterpret_cast<const void*>(c.kernel));
  ufd_launch(c.kernel, dim3(c.blocks), dim3(256), c.lds, s, c.q);
  return true;
}

// Profile label of the dual launch these arguments would get, or nullptr when launch_conv_dual would decline them.
const char* conv_dual_instance(const ConvArgs* a, int na, int a_stride, const ConvArgs* b, int b_stride) {
  DualChoice c;
  return choose_dual(a, na, a_stride, b, b_stride, &c) ? c.label : nullptr;
}

const char* conv_dwpw_instance(const ConvArgs* args, int n, int stride) {
  const ConvArgs& r = args[0];
  const long groups = (long)r.B * (r.oh * r.ow / 4);
  const long wave_tiles = (groups + kDwGroups - 1) / kDwGroups;
  const int ksteps = r.cin >> 1;
  int max_cts = 1;
  for (int i = 0; i < n; i++) max_cts = std::max(max_cts, (args[i].cout + 31) / 32);
  if (dwpw_uses_coop(args, n)) return stride == 1 ? (max_cts % 4 == 0 ? "<1, 4>" : "<1, 2>") : (max_cts % 4 == 0 ? "<2, 4>" : "<2, 2>");
  if (want_splitk(wave_tiles, max_cts * n, ksteps)) {
    if (want_sk8(true, n, wave_tiles * max_cts, ksteps, stride)) return "_sk8<1>";  // (dwpw_config's sk8)
    return stride == 1 ? "<1, 1, 2, 4>" : "<1, 2, 2, 4>";
  }
  const bool deep = ksteps % 4 == 0;
  return stride == 1 ? (deep ? "<1, 1, 2, 1>" : "<1, 1, 1, 1>") : (deep ? "<1, 2, 2, 1>" : "<1, 2, 1, 1>");
}

const char* conv_dwpw2_instance(const ConvArgs& first, const ConvArgs& second) {
  const int ct2 = (second.cout + 31) / 32;
  if (first.cin == 16) return ct2 == 1 ? "<16, 1, true>" : "<16, 2, true>";
  return ct2 == 1 ? "<32, 1, false>" : "<32, 2, false>";
}

bool dwpw2_supported(const ConvArgs& first, const ConvArgs& second) {
  return (first.cin == 16 || first.cin == 32) && first.cout == 32 && second.cin == 32 && second.cout <= 64 && second.cout % 32 == 0 && first.relu &&
         first.iw % 8 == 0 && first.iw == first.ow && first.ih == first.oh && second.ow % 4 == 0 && first.ow == 2 * second.ow &&
         first.oh == 2 * second.oh && first.res == nullptr && second.res == nullptr;
}

// Rows per band.  Measured on MI355X (rocprofv3, one batch in flight): a wave is bound by the SIMD's shared fp32
// matrix / vector issue whatever the band, so a taller band pays (fewer X1 rows computed) only while the launch still
// fills the GPU -- 2048 waves of this kernel are resident at once (two per SIMD).  m1 -> m2 of the 640 model at batch
// 32 (10240 waves at band 1): band 1 124 us, 2 116, 3 119, 5 104 (exactly one round of 2048 waves), 6 111, 8 137;
// m3 -> m4 (2560 waves at band 1): band 1 67 us, 2 70, 3 70.  Hence: the divisor of the output height that brings the
// launch closest to one full round, for launches of three rounds or more; band 1 otherwise.
static int dwpw2_band(const ConvArgs& second) {
  const long waves1 = (2L * second.B * (second.oh * second.ow / 4) + kDwGroups - 1) / kDwGroups;
  const double rounds = (double)waves1 / 2048.0;
  if (const char* e = experiment_env("UFD_BAND_SMALL"))  // (experiment knob)
    if (rounds < 3.0 && std::atoi(e) > 0 && second.oh % std::atoi(e) == 0) return std::atoi(e);
  if (rounds < 3.0) return 1;
  int best = 1;
  double best_d = 1e30;
  for (int r = 1; r <= 8; r++) {
    if (second.oh % r) continue;
    const double d = std::fabs(rounds / r - 1.0) + (rounds / r > 1.02 ? 0.5 : 0.0);  // (just over one round = two rounds)
    if (d < best_d) best_d = d, best = r;
  }
  return best;
}

void launch_conv_dwpw2_mfma(const ConvArgs& first, const ConvArgs& second, hipStream_t s) {
  ConvArgs3 p{};
  p.a[0] = first;
  p.a[1] = second;
  const int band = dwpw2_band(second);
  const long half_groups = 2L * second.B * ((second.oh / band) * second.ow / 4);  // one lane column per 2 output pixels of a band
  const long wave_tiles = (half_groups + kDwGroups - 1) / kDwGroups;
  p.a[1].tiles = (int)((wave_tiles + 3) / 4);
  p.a[1].cts = 1;
  p.a[1].band = band;
  const int ct2 = (second.cout + 31) / 32;
  // tables: first depthwise [3][cin][12] + its edge taps [4][cin][8], first pointwise [cin/2][64], second depthwise [2][32][16], second
  // pointwise [ct2][16][64], biases
  size_t lds = ((size_t)first.cin * (36 + 32) + (first.cin / 2) * 64 + 2 * 32 * 16 + (size_t)ct2 * 16 * 64 + 32 + 32 * ct2) * sizeof(float);
  const dim3 grid((unsigned)((p.a[1].tiles + 7) / 8 * 8));
  // the 16-channel instances keep a two-row ring per wave in LDS: 64 KB per block, two blocks per CU
  auto launch = [&](void (*kernel)(ConvArgs3), bool ring) {
    if (ring) lds += (size_t)4 * 2 * (first.cin / 2) * 64 * sizeof(float4);
    if (lds > 64 * 1024) allow_large_lds(reinterpret_cast<const void*>(kernel));
    ufd_launch(kernel, grid, dim3(256), lds, s, p);
  };
  if (first.cin == 16 && ct2 == 1) launch(k_dwpw2_mfma<16, 1, true>, true);
  else if (first.cin == 16) launch(k_dwpw2_mfma<16, 2, true>, true);
  else if (ct2 == 1) launch(k_dwpw2_mfma<32, 1, false>, false);
  else launch(k_dwpw2_mfma<32, 2, false>, false);
}

void launch_conv3x3_mfma(const ConvArgs* args, int n, hipStream_t s) {
  ConvArgs3 p{};
  for (int i = 0; i < n; i++) p.a[i] = args[i];
  const ConvArgs& a = args[0];
  const long total = (long)a.B * a.oh * a.ow;
  const unsigned ny = (unsigned)n;
  if (total >= 64L * 4 * 1024) {  // enough pixels to fill the chip with 4 groups per wave
    if (a.cin % 2 == 0)
      ufd_launch((k_conv3x3_mfma<4, 2>), dim3((unsigned)((total + 255) / 256), ny), dim3(256), 0, s, p);
    else
      ufd_launch((k_conv3x3_mfma<4, 1>), dim3((unsigned)((total + 255) / 256), ny), dim3(256), 0, s, p);
  } else {
    if (a.cin % 32 == 0 && a.cin >= 128 && total < 16384)  // few pixels, long channel chain: split-K
      ufd_launch((k_conv3x3_mfma<1, 8, 4>), dim3((unsigned)((total + 15) / 16), ny), dim3(256), 0, s, p);
    else if (a.cin % 8 == 0)
      ufd_launch((k_conv3x3_mfma<1, 8>), dim3((unsigned)((total + 63) / 64), ny), dim3(256), 0, s, p);
    else if (a.cin % 4 == 0)
      ufd_launch((k_conv3x3_mfma<1, 4>), dim3((unsigned)((total + 63) / 64), ny), dim3(256), 0, s, p);
    else
      ufd_launch((k_conv3x3_mfma<1, 1>), dim3((unsigned)((total + 63) / 64), ny), dim3(256), 0, s, p);
  }
}

void launch_conv_direct(const ConvArgs& a, hipStream_t s) {
  const unsigned gx = (a.oh * a.ow + 255) / 256;
  if (a.depthwise) {
    ufd_launch((k_conv_direct<1, true>), dim3(gx, a.cout, a.B), dim3(256), 0, s, a);
  } else if (a.cout >= 16) {
    ufd_launch((k_conv_direct<16, false>), dim3(gx, (a.cout + 15) / 16, a.B), dim3(256), 0, s, a);
  } else {
    ufd_launch((k_conv_direct<4, false>), dim3(gx, (a.cout + 3) / 4, a.B), dim3(256), 0, s, a);
  }
}

}  // namespace ufd
