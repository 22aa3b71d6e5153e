// post_kernels.hip -- tail of row A6 and rows A7-A10:
//   softmax over the 2 classes + SSD prior-box decode (embedded in the ONNX graph that
//     `self.model.run` executes, infer_server/src/nn.rs:181)
//   UltrafaceModel::postproc: keep conf > min_confidence (strict), stable ascending sort
//     (nn.rs:109-134), then non_maximum_suppression popping from the back (nn.rs:198-224) with
//     iou / bbox_area in the reference's exact f32 operation order (nn.rs:227-260, EPS nn.rs:18).
// Processing order = confidence descending, ties by HIGHER prior index first: encoded as one
// 64-bit key (orderable confidence bits << 32 | prior index + 1) sorted descending.
#include "kernels.hpp"
#include <cstdlib>

namespace ufd {
namespace {

constexpr float kEps = 1.0e-7f;           // nn.rs:18
constexpr int kSortLds = 2048;            // keys sorted inside LDS up to this many candidates (more: sorted in HBM)
constexpr int kSelLds = 512;              // selected boxes kept in LDS; the rest spill to HBM (only frames with more than
                                          // kMatMax candidates get there: the others have at most kNmsBlock or leave for
                                          // the matrix path).  LDS is what keeps other kernels off the frame's CU: 40 KB, not 68
constexpr int kBoxLds = 256;              // sorted candidate boxes gathered per round trip (power of two, <= block size)
constexpr int kNmsBlock = 256;            // candidates per greedy block (4 x 64-bit suppression masks per row)
// Frames with more candidates than one greedy block (and at most kMatMax) leave k_sort_nms after the
// sort: their pairwise suppression matrix is computed by the whole GPU (k_nms_matrix) and resolved
// by one wave per frame (k_nms_scan).  Inside k_sort_nms the quadratic work of such a frame runs on
// ONE compute unit and set the latency of the whole batch (1359 candidates: 380 us).
constexpr int kMatMin = 256;              // frames with more candidates than this take the matrix path
constexpr int kMatMax = 2048;             // = kSortLds: the sorted keys of such a frame are in LDS
constexpr int kMatWords = kMatMax / 64;   // 64-bit words per matrix row
constexpr uint32_t kHeavyFlag = 0x80000000u;  // ndet[frame] = kHeavyFlag | n between the kernels

__device__ __forceinline__ unsigned long long make_key(float conf, uint32_t k) {
  conf = conf + 0.0f;  // -0.0 -> +0.0 (partial_cmp treats them as equal)
  uint32_t u = __float_as_uint(conf);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> uint
  return ((unsigned long long)u << 32) | (unsigned long long)(k + 1);
}
__device__ __forceinline__ float key_conf(unsigned long long key) {
  uint32_t u = (uint32_t)(key >> 32);
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}

__global__ __launch_bounds__(256) void k_head_decode(HeadArgs h, const float* __restrict__ priors, int K,
                                                     float min_conf, float* __restrict__ scores,
                                                     float* __restrict__ boxes, unsigned long long* __restrict__ keys,
                                                     size_t key_stride, uint32_t* __restrict__ counts) {
  const int frame = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const int head = k >= h.base[3] ? 3 : (k >= h.base[2] ? 2 : (k >= h.base[1] ? 1 : 0));
  const int A = h.anchors[head], plane = h.plane[head];
  const int local = k - h.base[head];
  const int p = local / A, an = local - p * A;
  const float* cls = h.cls[head] + (size_t)frame * (A * 2) * plane;
  const float* reg = h.reg[head] + (size_t)frame * (A * 4) * plane;
  // NCHW [A*c][fh][fw] -> NHWC -> [fh*fw*A][c]
  const float s0 = cls[(size_t)(an * 2 + 0) * plane + p], s1 = cls[(size_t)(an * 2 + 1) * plane + p];
  const float mx = fmaxf(s0, s1);
  const float e0 = expf(s0 - mx), e1 = expf(s1 - mx);
  const float sum = e0 + e1;
  const float c0 = e0 / sum, c1 = e1 / sum;
  const float l0 = reg[(size_t)(an * 4 + 0) * plane + p], l1 = reg[(size_t)(an * 4 + 1) * plane + p];
  const float l2 = reg[(size_t)(an * 4 + 2) * plane + p], l3 = reg[(size_t)(an * 4 + 3) * plane + p];
  const float4 pr = reinterpret_cast<const float4*>(priors)[k];
  const float cx = __fadd_rn(__fmul_rn(__fmul_rn(l0, 0.1f), pr.z), pr.x);
  const float cy = __fadd_rn(__fmul_rn(__fmul_rn(l1, 0.1f), pr.w), pr.y);
  const float bw = expf(l2 * 0.2f) * pr.z, bh = expf(l3 * 0.2f) * pr.w;
  float4 bb;
  bb.x = cx - bw / 2.0f;
  bb.y = cy - bh / 2.0f;
  bb.z = cx + bw / 2.0f;
  bb.w = cy + bh / 2.0f;
  const size_t o = (size_t)frame * K + k;
  const bool cand = c1 > min_conf;  // strict; NaN fails
  // the raw [K, 6] outputs exist for the stage tap only (scores != null): the product keeps the boxes of the
  // candidates, which is all the sort / NMS kernels read
  if (scores) reinterpret_cast<float2*>(scores)[o] = make_float2(c0, c1);
  if (scores || cand) reinterpret_cast<float4*>(boxes)[o] = bb;
  if (cand) {
    const uint32_t pos = atomicAdd(&counts[frame], 1u);
    keys[(size_t)frame * key_stride + pos] = make_key(c1, (uint32_t)k);
  }
}

__global__ __launch_bounds__(256) void k_threshold(const float* __restrict__ scores, int K, float min_conf,
                                                   unsigned long long* __restrict__ keys, size_t key_stride,
                                                   uint32_t* __restrict__ counts) {
  const int frame = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const float c1 = scores[((size_t)frame * K + k) * 2 + 1];
  if (c1 > min_conf) {
    const uint32_t pos = atomicAdd(&counts[frame], 1u);
    keys[(size_t)frame * key_stride + pos] = make_key(c1, (uint32_t)k);
  }
}

// nn.rs:251-260
__device__ __forceinline__ float bbox_area(float x0, float y0, float x1, float y1) {
  const float width = __fsub_rn(y1, y0);
  const float height = __fsub_rn(x1, x0);
  if (width < 0.0f || height < 0.0f) return 0.0f;
  return __fmul_rn(width, height);
}

// descending bitonic sort of n2 (power of two) keys by one workgroup
// Bitonic sort, descending.  Pair t of a stage with stride s is (lo, lo + s), lo = 2t - (t & (s-1)):
// for s <= 64 the 64 pairs of a wave (t = 64w .. 64w+63, every pass of the t loop) stay inside one
// 128-key chunk that no other wave touches, so those stages need no workgroup barrier -- only the
// wave's own LDS/global accesses in order.  A 2048-key sort has 10 barrier stages instead of 66
// (a barrier of 16 waves costs ~0.5-1 us: the sort was most of the kernel).
__device__ void bitonic_desc(unsigned long long* keys, int n2, int tid, int nthreads, bool in_lds) {
  for (int size = 2; size <= n2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride >= 64 || !in_lds) __syncthreads();              // pairs cross the waves' chunks (64: the stage before did)
      else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // own stores before own loads
      for (int t = tid; t < (n2 >> 1); t += nthreads) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const unsigned long long a = keys[lo], b = keys[hi];
        if ((a < b) == desc) {
          keys[lo] = b;
          keys[hi] = a;
        }
      }
    }
  }
  __syncthreads();
}

// iou(candidate, selected) > max_iou, nn.rs:227-243 operation order
__device__ __forceinline__ bool iou_exceeds(const float4 c, float area_c, const float4 sb, float max_iou) {
  const float ox0 = fmaxf(c.x, sb.x), oy0 = fmaxf(c.y, sb.y), ox1 = fminf(c.z, sb.z), oy1 = fminf(c.w, sb.w);
  const float overlap = bbox_area(ox0, oy0, ox1, oy1);
  // Most pairs do not overlap: iou = 0 / (positive) = 0 exactly, so the IEEE division (the bulk of
  // this function) is skipped when no lane of the wave has an overlap.  (0 > max_iou keeps the
  // reference's answer for a negative threshold.)
  if (__ballot(overlap != 0.0f) == 0ull) return 0.0f > max_iou;
  const float denom = __fadd_rn(__fsub_rn(__fadd_rn(area_c, bbox_area(sb.x, sb.y, sb.z, sb.w)), overlap), kEps);
  return __fdiv_rn(overlap, denom) > max_iou;
}

// One 1024-thread workgroup per frame (16 waves; frames with thousands of candidates set the
// batch latency, so the quadratic phases are spread as wide as a workgroup goes): bitonic sort of the candidate keys, then greedy NMS in
// blocks of 64 candidates (sorted order).  Per block:
//   phase 1  every wave tests the block's 64 candidates (one per lane) against a stripe of the
//            boxes selected so far -> 64-bit "dead" mask (ballot + LDS atomicOr)
//   phase 2  64x64 pairwise matrix inside the block (row i = candidates j > i it would suppress)
//   phase 3  wave 0 resolves the block serially with bit operations: i survives iff not dead;
//            a survivor kills row i.  Exactly the reference's greedy order (nn.rs:198-224).
__global__ __launch_bounds__(1024) void k_sort_nms(unsigned long long* __restrict__ gkeys, size_t key_stride,
                                                  uint32_t* __restrict__ counts, const float* __restrict__ boxes,
                                                  int K, float max_iou, Det* __restrict__ dets, uint32_t det_stride,
                                                  uint32_t* __restrict__ ndet, float4* __restrict__ spill, int mat_min) {
  __shared__ unsigned long long s_keys[kSortLds];
  __shared__ float4 s_sel[kSelLds];
  __shared__ float4 s_cand[kNmsBlock];
  __shared__ float4 s_box[kBoxLds];
  __shared__ unsigned long long s_row[kNmsBlock][kNmsBlock / 64];  // row i: later candidates of the block that i suppresses
  __shared__ unsigned long long s_dead[kNmsBlock / 64], s_keep[kNmsBlock / 64];
  __shared__ int s_nsel;
  const int frame = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = blockDim.x, nwave = nthr >> 6;
  const int n = min((int)counts[frame], K);
  unsigned long long* fkeys = gkeys + (size_t)frame * key_stride;
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  unsigned long long* keys;
  if (n2 <= kSortLds) {
    for (int i = tid; i < n2; i += nthr) s_keys[i] = i < n ? fkeys[i] : 0ull;
    keys = s_keys;
  } else {
    for (int i = n + tid; i < n2; i += nthr) fkeys[i] = 0ull;  // key_stride >= next pow2 of K
    keys = fkeys;
  }
  if (tid == 0) s_nsel = 0;
  __syncthreads();
  // every thread has read the frame's candidate count: leave it at zero for the next batch's k_head_decode / k_threshold
  // (the counters are zero whenever no batch is between those kernels and this one: no memset launch per batch)
  if (tid == 0) counts[frame] = 0;
  if (n > 1) bitonic_desc(keys, n2, tid, nthr, keys == s_keys);
  const float4* fb = reinterpret_cast<const float4*>(boxes) + (size_t)frame * K;
  float4* fspill = spill + (size_t)frame * K;
  Det* fd = dets + (size_t)frame * det_stride;
  if (n > mat_min && n <= kMatMax) {
    // sorted keys and boxes for k_nms_matrix / k_nms_scan (the spill area is free: it only holds
    // selected boxes of the in-kernel path)
    for (int i = tid; i < n; i += nthr) {
      const unsigned long long key = keys[i];
      fkeys[i] = key;
      fspill[i] = fb[(int)(key & 0xffffffffull) - 1];
    }
    if (tid == 0) ndet[frame] = kHeavyFlag | (uint32_t)n;
    return;
  }

  constexpr int Q = kNmsBlock / 64;  // 64-candidate groups per block
  for (int b0 = 0; b0 < n; b0 += kNmsBlock) {
    const int m = min(kNmsBlock, n - b0);
    // candidate boxes in sorted order, kBoxLds at a time: one gather round trip per several blocks
    if ((b0 & (kBoxLds - 1)) == 0) {
      if (tid < kBoxLds && b0 + tid < n) s_box[tid] = fb[(int)(keys[b0 + tid] & 0xffffffffull) - 1];
      __syncthreads();
    }
    if (tid < kNmsBlock) {
      s_cand[tid] = tid < m ? s_box[(b0 & (kBoxLds - 1)) + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < Q; q++) s_row[tid][q] = 0ull;
    }
    if (tid < Q) s_dead[tid] = 0ull;
    __syncthreads();
    const int nsel = s_nsel;
    // the lane's candidates: lane, lane + 64, ...
    float4 c[Q];
    float area_c[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
      c[q] = s_cand[q * 64 + lane];
      area_c[q] = bbox_area(c[q].x, c[q].y, c[q].z, c[q].w);
    }
    // phase 1: against previously selected boxes, wave w takes s = w, w + nwave, ...
    unsigned long long dead[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) dead[q] = 0ull;
    for (int sidx = wave; sidx < nsel; sidx += nwave) {
      const float4 sb = sidx < kSelLds ? s_sel[sidx] : fspill[sidx];
#pragma unroll
      for (int q = 0; q < Q; q++) dead[q] |= __ballot(q * 64 + lane < m && iou_exceeds(c[q], area_c[q], sb, max_iou));
    }
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < Q; q++)
        if (dead[q]) atomicOr(&s_dead[q], dead[q]);
    }
    // phase 2: inside the block, wave w takes rows i = w, w + nwave, ...; lanes = the later columns
    for (int i = wave; i < m; i += nwave) {
      const float4 sb = s_cand[i];  // the earlier (higher-confidence) box plays "selected"
#pragma unroll
      for (int q = 0; q < Q; q++) {
        const int j = q * 64 + lane;
        const unsigned long long row = (q * 64 + 63 > i) ? __ballot(j > i && j < m && iou_exceeds(c[q], area_c[q], sb, max_iou)) : 0ull;
        if (lane == 0) s_row[i][q] = row;
      }
    }
    __syncthreads();
    // phase 3: greedy pass over the block in scalar registers (lane l of wave 0 holds rows l, l + 64, ...)
    if (wave == 0) {
      uint32_t rlo[Q][Q], rhi[Q][Q];
#pragma unroll
      for (int g = 0; g < Q; g++)
#pragma unroll
        for (int q = 0; q < Q; q++) {
          const unsigned long long r = s_row[g * 64 + lane][q];
          rlo[g][q] = (uint32_t)r, rhi[g][q] = (uint32_t)(r >> 32);
        }
      unsigned long long d[Q], keep[Q];
#pragma unroll
      for (int q = 0; q < Q; q++) d[q] = s_dead[q], keep[q] = 0ull;
#pragma unroll
      for (int g = 0; g < Q; g++) {
        const int cnt = min(64, m - g * 64);
        // rows that suppress nothing (most of them) need no register reads
        bool any = false;
#pragma unroll
        for (int q = g; q < Q; q++) any = any || (rlo[g][q] | rhi[g][q]) != 0u;
        const unsigned long long nz = __ballot(any);
        // everything alive in front of the next survivor with a non-empty row is kept in one step
        unsigned long long alive = cnt <= 0 ? 0ull : (~d[g] & (cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull)));  // (m may end before group g)
        while (alive) {
          const unsigned long long hot = alive & nz;
          if (!hot) {
            keep[g] |= alive;
            break;
          }
          const int i = __builtin_ctzll(hot);
          const unsigned long long upto = (2ull << i) - 1ull;  // bits 0..i
          keep[g] |= alive & upto;
#pragma unroll
          for (int q = g; q < Q; q++)  // a row only has bits of later candidates
            d[q] |= ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)rhi[g][q], i) << 32) |
                    (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)rlo[g][q], i);
          alive &= ~(upto | d[g]);
        }
      }
      if (lane < Q) s_keep[lane] = keep[lane == 0 ? 0 : (lane == 1 ? 1 : (lane == 2 ? 2 : 3))];
    }
    __syncthreads();
    // kept candidates -> selected list and output, in order
    if (tid < kNmsBlock) {
      const int g = tid >> 6;
      const unsigned long long kg = s_keep[g];
      if ((kg >> lane) & 1ull) {
        int pos = nsel + __popcll(kg & ((1ull << lane) - 1ull));
        for (int q = 0; q < g; q++) pos += __popcll(s_keep[q]);
        const float4 cc = s_cand[tid];
        if (pos < kSelLds)
          s_sel[pos] = cc;
        else
          fspill[pos] = cc;
        if ((uint32_t)pos < det_stride) {
          Det dd;
          dd.x_tl = cc.x, dd.y_tl = cc.y, dd.x_br = cc.z, dd.y_br = cc.w, dd.conf = key_conf(keys[b0 + tid]);
          fd[pos] = dd;
        }
      }
    }
    __threadfence_block();  // spilled boxes are read back by the other waves
    __syncthreads();
    if (tid == 0) {
      int add = 0;
      for (int q = 0; q < Q; q++) add += __popcll(s_keep[q]);
      s_nsel = nsel + add;
    }
    __syncthreads();
  }
  if (tid == 0) ndet[frame] = (uint32_t)s_nsel;
}

// Suppression matrix of a heavy frame: bit j of row i (i < j, sorted order) = "i, once selected,
// suppresses j".  Block (rb, cg) = 64 rows x 4 column words, one word per wave, lane = row; every
// test is iou_exceeds(candidate j, selected i), the in-kernel path's call.
constexpr int kMatColGroups = kMatWords / 4;
__global__ __launch_bounds__(256) void k_nms_matrix(const uint32_t* __restrict__ ndet, const float4* __restrict__ spill, int K,
                                                    float max_iou, unsigned long long* __restrict__ mat) {
  __shared__ float4 s_col[4][64];
  __shared__ float s_area[4][64];
  const int frame = blockIdx.z, rb = blockIdx.x, cg = blockIdx.y;
  const uint32_t flag = ndet[frame];
  if (!(flag & kHeavyFlag)) return;
  const int n = (int)(flag & ~kHeavyFlag);
  if (rb * 64 >= n || cg * 4 + 3 < rb || cg * 256 >= n) return;  // past the end / below the diagonal
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float4* fsp = spill + (size_t)frame * K;
  const int cb = cg * 4 + wave;  // this wave's column word
  {
    const int j = min(cb * 64 + lane, n - 1);
    const float4 b = fsp[j];
    s_col[wave][lane] = b;
    s_area[wave][lane] = bbox_area(b.x, b.y, b.z, b.w);
  }
  const int i = rb * 64 + lane;
  const float4 bi = fsp[min(i, n - 1)];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the wave reads only its own s_col / s_area row
  if (cb < rb || cb * 64 >= n) return;
  unsigned long long word = 0ull;
  const int jn = min(64, n - cb * 64);
  for (int b = 0; b < jn; b++) {
    const int j = cb * 64 + b;
    const bool hit = iou_exceeds(s_col[wave][b], s_area[wave][b], bi, max_iou);  // (wave-uniform j: the ballot inside sees all rows)
    word |= (hit && j > i) ? (1ull << b) : 0ull;
  }
  if (i < n) mat[((size_t)frame * kMatMax + i) * kMatWords + cb] = word;
}

// Greedy pass over the matrix, exactly the reference's order (nn.rs:198-224): candidate i is
// selected iff no selected earlier candidate has its bit set.  One wave per frame walks the 64-row
// blocks; the removed-mask is spread over the lanes (lanes l and l + 32 both hold word l), a block's
// rows arrive two per load instruction (lane = row parity x word) and are prefetched one block
// ahead in registers; no LDS, no barrier.
__global__ __launch_bounds__(64) void k_nms_scan(const unsigned long long* __restrict__ gkeys, size_t key_stride,
                                                 const float4* __restrict__ spill, int K, const unsigned long long* __restrict__ mat,
                                                 Det* __restrict__ dets, uint32_t det_stride, uint32_t* __restrict__ ndet) {
  static_assert(kMatWords == 32, "lane = (row parity, word) layout");
  const int frame = blockIdx.x;
  const uint32_t flag = ndet[frame];
  if (!(flag & kHeavyFlag)) return;
  const int n = (int)(flag & ~kHeavyFlag);
  const int lane = threadIdx.x, half = lane >> 5, word = lane & 31;
  const int nrb = (n + 63) >> 6;
  const unsigned long long* fmat = mat + (size_t)frame * kMatMax * kMatWords;
  const unsigned long long* fkeys = gkeys + (size_t)frame * key_stride;
  const float4* fsp = spill + (size_t)frame * K;
  Det* fd = dets + (size_t)frame * det_stride;
  unsigned long long rows[32], rows_next[32], diag, diag_next;
  // rows 2q + half of block rb, word `word`; the diagonal word of row `lane` (rows past n and words
  // below the diagonal were never written: loaded from clamped rows and masked where they are used)
  auto load_block = [&](int rb, unsigned long long (&r)[32], unsigned long long& d) {
    const int rbc = min(rb, nrb - 1);
#pragma unroll
    for (int q = 0; q < 32; q++) r[q] = fmat[(size_t)min(rbc * 64 + 2 * q + half, n - 1) * kMatWords + word];
    d = fmat[(size_t)min(rbc * 64 + lane, n - 1) * kMatWords + rbc];
  };
  load_block(0, rows, diag);
  unsigned long long removed = 0ull;
  int nsel = 0;
  for (int rb = 0; rb < nrb; rb++) {
    load_block(rb + 1, rows_next, diag_next);
    const int cnt = min(64, n - rb * 64);
    const unsigned long long valid = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
    const uint32_t rm_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)removed, rb);
    const uint32_t rm_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(removed >> 32), rb);
    unsigned long long alive = ~(((unsigned long long)rm_hi << 32) | rm_lo) & valid;
    const uint32_t dlo = (uint32_t)diag, dhi = (uint32_t)(diag >> 32);
    // Only a candidate whose row has bits inside the block can change `alive`: everything alive
    // in front of the next such candidate is kept in one step (most rows are empty: boxes rarely
    // overlap), so the serial chain has one iteration per overlapping survivor, not per candidate.
    const unsigned long long nz = __ballot(diag != 0ull);
    unsigned long long kept = 0ull;
    while (alive) {
      const unsigned long long hot = alive & nz;
      if (!hot) {
        kept |= alive;
        break;
      }
      const int b = __builtin_ctzll(hot);
      const unsigned long long upto = (2ull << b) - 1ull;  // bits 0..b
      kept |= alive & upto;
      const unsigned long long row = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)dhi, b) << 32) |
                                     (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)dlo, b);
      alive &= ~(row | upto);
    }
    // the kept rows' later words join the mask
    const unsigned long long kh = kept >> half;  // bit 2q = row 2q + half
    unsigned long long acc = 0ull;
#pragma unroll
    for (int q = 0; q < 32; q++) acc |= ((kh >> (2 * q)) & 1ull) ? rows[q] : 0ull;
    const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)acc, 32), ohi = (uint32_t)__shfl_xor((int)(uint32_t)(acc >> 32), 32);
    acc |= ((unsigned long long)ohi << 32) | olo;
    if (word > rb && word < nrb) removed |= acc;
    if ((kept >> lane) & 1ull) {
      const int pos = nsel + __popcll(kept & ((1ull << lane) - 1ull));
      if ((uint32_t)pos < det_stride) {
        const float4 cc = fsp[rb * 64 + lane];
        Det dd;
        dd.x_tl = cc.x, dd.y_tl = cc.y, dd.x_br = cc.z, dd.y_br = cc.w, dd.conf = key_conf(fkeys[rb * 64 + lane]);
        fd[pos] = dd;
      }
    }
    nsel += __popcll(kept);
#pragma unroll
    for (int q = 0; q < 32; q++) rows[q] = rows_next[q];
    diag = diag_next;
  }
  if (lane == 0) ndet[frame] = (uint32_t)nsel;
}

}  // namespace

void launch_head_decode(const HeadArgs& h, const float* d_priors, uint32_t B, float min_conf, float* d_scores,
                        float* d_boxes, unsigned long long* d_keys, size_t key_stride, uint32_t* d_counts,
                        hipStream_t s) {
  const int K = h.base[4];
  hipLaunchKernelGGL(k_head_decode, dim3((K + 255) / 256, B), dim3(256), 0, s, h, d_priors, K, min_conf, d_scores,
                     d_boxes, d_keys, key_stride, d_counts);
}

void launch_threshold(const float* d_scores, uint32_t K, uint32_t B, float min_conf, unsigned long long* d_keys,
                      size_t key_stride, uint32_t* d_counts, hipStream_t s) {
  hipLaunchKernelGGL(k_threshold, dim3((K + 255) / 256, B), dim3(256), 0, s, d_scores, (int)K, min_conf, d_keys,
                     key_stride, d_counts);
}

void launch_sort_nms(unsigned long long* d_keys, size_t key_stride, uint32_t* d_counts, const float* d_boxes,
                     uint32_t K, float max_iou, Det* d_dets, uint32_t det_stride, uint32_t* d_ndet, float4* d_sel_spill,
                     unsigned long long* d_mat, uint32_t B, hipStream_t s) {
  const int knob = kMatMin;
  const bool use_matrix = d_mat != nullptr && K > (uint32_t)knob;
  hipLaunchKernelGGL(k_sort_nms, dim3(B), dim3(1024), 0, s, d_keys, key_stride, d_counts, d_boxes, (int)K, max_iou,
                     d_dets, det_stride, d_ndet, d_sel_spill, use_matrix ? knob : kMatMax);
  if (!use_matrix) return;
  // (both return at once for frames the first kernel finished itself)
  hipLaunchKernelGGL(k_nms_matrix, dim3(kMatWords, kMatColGroups, B), dim3(256), 0, s, d_ndet, d_sel_spill, (int)K, max_iou, d_mat);
  hipLaunchKernelGGL(k_nms_scan, dim3(B), dim3(64), 0, s, d_keys, key_stride, d_sel_spill, (int)K, d_mat, d_dets, det_stride,
                     d_ndet);
}
size_t nms_matrix_bytes(uint32_t B) { return (size_t)B * kMatMax * kMatWords * sizeof(unsigned long long); }

}  // namespace ufd
