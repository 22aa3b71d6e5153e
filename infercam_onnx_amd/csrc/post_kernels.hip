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
constexpr int kMatHalf = 2048;            // the scan walks the matrix in squares of this many rows / columns (up to four on the diagonal)
constexpr int kMatMax = 8192;             // frames up to this many candidates take it (round 4; 4096 before: the reference has no cap,
                                          // nn.rs:198-224, and a low threshold on a crowd picture reaches thousands)
constexpr int kMatWords = kMatMax / 64;   // 64-bit words per matrix row
// rows of a frame's matrix: a frame has at most K candidates (UltraFace-320: 4420, 4.5 MB per frame instead of 8)
__host__ __device__ constexpr int mat_rows(int K) { return K < kMatMax ? K : kMatMax; }
constexpr int kHalfWords = kMatHalf / 64;
constexpr int kMatGridRows = 64;          // row blocks in k_nms_matrix's grid: a frame with more than 4096 candidates takes two passes
constexpr int kSortLdsHeavy = 4096;       // a frame with more than kSortLds candidates leaves for the matrix path after its
                                          // sort: the LDS of the greedy phase (selected boxes, block rows) holds its keys instead
                                          // (4096 at a time: a frame of 4097..8192 candidates sorts its two halves one after the other)
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

// Bitonic sort by one workgroup: every block of `top` keys (a power of two <= n2, itself a power of two) ends up sorted
// descending.  Pair t of a stage with stride s is (lo, lo + s), lo = 2t - (t & (s-1)).  In LDS a wave owns a CONTIGUOUS
// run of pairs, i.e. a chunk of 128 * iters keys: a stage with 2s <= chunk touches only the wave's own chunk and needs no
// workgroup barrier -- only the wave's own LDS accesses in order -- unless the stage before it crossed chunks.  A 2048-key
// sort has 10 barrier stages instead of 66, the 1024-key blocks of a 4096-key frame 5 (a barrier of 16 waves costs
// ~0.5-1 us: the sort was most of the kernel).
// (KeyPtr: the keys' address space is part of the type -- LDS or global.  Through a generic pointer every key access was a
// FLAT instruction, which reaches LDS the long way round: the sort's inner loop is two loads and two stores.)
typedef __attribute__((address_space(3))) unsigned long long* LdsKeys;
typedef __attribute__((address_space(1))) unsigned long long* GlobalKeys;
template <typename KeyPtr>
__device__ void bitonic_desc_in(KeyPtr keys, int n2, int top, int tid, int nthreads, bool in_lds) {
  const int pairs = n2 >> 1, lane = tid & 63, wave = tid >> 6;
  const int iters = max(1, pairs / nthreads);
  const int chunk = in_lds ? iters * 128 : 0;
  bool prev_cross = true;
  for (int size = 2; size <= top; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const bool cross = 2 * stride > chunk;
      if (cross || prev_cross) __syncthreads();
      else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // own stores before own loads
      prev_cross = cross;
      for (int p = 0; p < iters; p++) {
        const int t = in_lds ? (wave * iters + p) * 64 + lane : tid + p * nthreads;
        if (t >= pairs) break;
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = size == top || (lo & size) == 0;
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
__device__ void bitonic_desc(unsigned long long* keys, int n2, int top, int tid, int nthreads, bool in_lds) {
  if (in_lds) bitonic_desc_in((LdsKeys)keys, n2, top, tid, nthreads, true);
  else bitonic_desc_in((GlobalKeys)keys, n2, top, tid, nthreads, false);
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
                                                  uint32_t* __restrict__ ndet, float4* __restrict__ spill, int mat_min, int mat_max,
                                                  NmsHostOut ho) {
  // One 40 KB block of LDS, carved twice.  Greedy path (frames the kernel finishes itself): 2048 sort keys | selected
  // boxes | the block's candidates | gathered boxes | block rows.  A frame with 2049..4096 candidates only sorts here
  // (it leaves for the matrix path): its 4096 key slots lie over the first four pieces, which it never uses.
  constexpr int kRawWords = kSortLds + 2 * kSelLds + 2 * kNmsBlock + 2 * kBoxLds + kNmsBlock * (kNmsBlock / 64);
  static_assert(kRawWords >= kSortLdsHeavy, "the heavy sort's keys must fit the greedy phase's LDS");
  __shared__ __attribute__((aligned(16))) unsigned long long s_raw[kRawWords];
  unsigned long long* const s_keys = s_raw;
  float4* const s_sel = reinterpret_cast<float4*>(s_raw + kSortLds);
  float4* const s_cand = s_sel + kSelLds;
  float4* const s_box = s_cand + kNmsBlock;
  unsigned long long(*const s_row)[kNmsBlock / 64] = reinterpret_cast<unsigned long long(*)[kNmsBlock / 64]>(s_box + kBoxLds);
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
  const bool heavy = n > mat_min && n <= mat_max;  // finished by k_nms_matrix / k_nms_scan
  if (heavy && n2 > kSortLdsHeavy) {
    // 4097..8192 candidates: eight blocks of 1024 keys, sorted four at a time in LDS (the two halves of the key array, written
    // back in place), then merged by RANK like the four blocks of a 4096-key frame below: a key's place in the frame's order is
    // its place in its own block plus, per other block, the number of keys there that precede it -- counted by binary search
    // against the four blocks LDS holds (first 4..7, left there by the second sort, then 0..3 fetched back).  Thread t owns
    // position t of every block: its eight keys and partial ranks stay in registers, so once the second count is done nobody
    // reads the key array any more and the keys (and their boxes) are scattered to their final places in it.
    constexpr int CB = kSortLds / 2, NB = 2 * kSortLdsHeavy / CB;  // 1024, 8
    const LdsKeys lkeys = (LdsKeys)s_keys;
    if (tid == 0) s_nsel = 0;
    for (int h = 0; h < 2; h++) {
      __syncthreads();
      for (int i = tid; i < kSortLdsHeavy; i += nthr) s_keys[i] = h * kSortLdsHeavy + i < n ? fkeys[h * kSortLdsHeavy + i] : 0ull;
      __syncthreads();
      if (h == 0 && tid == 0) counts[frame] = 0;  // (every thread has read the count: see below)
      bitonic_desc(s_keys, kSortLdsHeavy, CB, tid, nthr, true);
      for (int i = tid; i < kSortLdsHeavy; i += nthr) fkeys[h * kSortLdsHeavy + i] = s_keys[i];
    }
    __threadfence_block();
    __syncthreads();
    unsigned long long mine[NB];
    int rank[NB];
    auto count_in_lds = [&](int first_block) {  // LDS holds blocks first_block .. first_block + 3
#pragma unroll
      for (int c = 0; c < NB; c++) {
        const unsigned long long key = mine[c];
        if (!key) continue;
        for (int o = 0; o < 4; o++) {
          if (first_block + o == c) continue;
          const LdsKeys blk = lkeys + o * CB;
          int lo = 0, hi = CB;  // first position whose key is smaller = number of keys greater
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (blk[mid] > key) lo = mid + 1;
            else hi = mid;
          }
          rank[c] += lo;
        }
      }
    };
#pragma unroll
    for (int c = 0; c < NB; c++) {
      mine[c] = c >= 4 ? (unsigned long long)lkeys[(c - 4) * CB + tid] : fkeys[c * CB + tid];
      rank[c] = tid;
    }
    count_in_lds(4);
    __syncthreads();
    for (int i = tid; i < kSortLdsHeavy; i += nthr) s_keys[i] = fkeys[i];
    __syncthreads();
    count_in_lds(0);
    __syncthreads();  // (every read of the key array is done: scatter in place)
    const float4* fb8 = reinterpret_cast<const float4*>(boxes) + (size_t)frame * K;
    float4* fspill8 = spill + (size_t)frame * K;
#pragma unroll
    for (int c = 0; c < NB; c++) {
      if (!mine[c]) continue;
      fkeys[rank[c]] = mine[c];
      fspill8[rank[c]] = fb8[(int)(mine[c] & 0xffffffffull) - 1];
    }
    if (tid == 0) ndet[frame] = kHeavyFlag | (uint32_t)n;
    return;
  }
  if (n2 <= kSortLds || (heavy && n2 <= kSortLdsHeavy)) {
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
  // (a frame of 2049..4096 candidates: its four 1024-key blocks only -- they are merged by rank on the way out)
  const bool merge4 = heavy && n2 > kSortLds;
  if (n > 1) bitonic_desc(keys, n2, merge4 ? kSortLds / 2 : n2, tid, nthr, keys == s_keys);
  const float4* fb = reinterpret_cast<const float4*>(boxes) + (size_t)frame * K;
  float4* fspill = spill + (size_t)frame * K;
  Det* fd = dets + (size_t)frame * det_stride;
  if (heavy) {
    // sorted keys and boxes for k_nms_matrix / k_nms_scan (the spill area is free: it only holds
    // selected boxes of the in-kernel path)
    if (merge4) {
      // Four sorted 1024-key blocks -> the frame's order: a key's rank is its place in its own block plus, per other
      // block, the number of keys there that precede it (keys are distinct: confidence bits | prior index), found by
      // binary search -- 30 LDS reads per key instead of 23 more sort stages with 11 barriers.  Padding keys (0) sort
      // last in their block and precede nothing.
      constexpr int CB = kSortLds / 2;  // 1024
      const LdsKeys lkeys = (LdsKeys)s_keys;  // (a heavy frame's keys are in LDS: the test above)
      for (int e = tid; e < n2; e += nthr) {
        const unsigned long long key = lkeys[e];
        if (!key) continue;
        const int c = e / CB;
        int rank = e - c * CB;
        for (int o = 0; o < n2 / CB; o++) {
          if (o == c) continue;
          const LdsKeys blk = lkeys + o * CB;
          int lo = 0, hi = CB;  // first position whose key is smaller = number of keys greater
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (blk[mid] > key) lo = mid + 1;
            else hi = mid;
          }
          rank += lo;
        }
        fkeys[rank] = key;
        fspill[rank] = fb[(int)(key & 0xffffffffull) - 1];
      }
    } else {
      for (int i = tid; i < n; i += nthr) {
        const unsigned long long key = keys[i];
        fkeys[i] = key;
        fspill[i] = fb[(int)(key & 0xffffffffull) - 1];
      }
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
  if (ho.h_ndet) {  // (launch-uniform) the frame's results straight into the slot's pinned host arrays: what k_results_out did
    // (s_nsel and the detections in fd are complete: the barrier at the end of the last block, or the one behind the sort)
    const uint32_t nd = (uint32_t)s_nsel;
    if (tid == 0) {
      ho.h_ndet[frame] = nd;
      if (ho.d_status) ho.h_status[frame] = ho.d_status[frame];
    }
    constexpr uint32_t kW = (uint32_t)(sizeof(Det) / 4);
    const uint32_t words = min(nd, ho.max_rows) * kW;
    const float* src = reinterpret_cast<const float*>(fd);
    float* dst = ho.h_dets + (size_t)frame * ho.max_rows * kW;
    for (uint32_t i = tid; i < words; i += (uint32_t)nthr) dst[i] = src[i];
  }
}

// Suppression matrix of a heavy frame: bit j of row i (i < j, sorted order) = "i, once selected,
// suppresses j".  Block (rb, cg) = 64 rows x 4 column words, one word per wave, lane = row; every
// test is iou_exceeds(candidate j, selected i), the in-kernel path's call.
// A frame with more than kMatHalf candidates has many times the blocks: the grid covers the row blocks of 4096 candidates
// and the column groups of one 2048-column square only (most batches have no such frame, and every block of the grid is
// launched for every frame: 4.6 us for the blocks of a batch that leave at once), and a block of a larger frame walks the
// column squares right of its row's own and, past 4096 candidates, a second row block.
constexpr int kMatColGroups = kHalfWords / 4;
__global__ __launch_bounds__(256) void k_nms_matrix(const uint32_t* __restrict__ ndet, const float4* __restrict__ spill, int K,
                                                    float max_iou, unsigned long long* __restrict__ mat) {
  __shared__ float4 s_col[4][64];
  __shared__ float s_area[4][64];
  const int frame = blockIdx.z;
  const uint32_t flag = ndet[frame];
  if (!(flag & kHeavyFlag)) return;
  const int n = (int)(flag & ~kHeavyFlag);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float4* fsp = spill + (size_t)frame * K;
  const int quads = (n + kMatHalf - 1) / kMatHalf;  // squares along the diagonal: 1..4
  // (the grid has kMatGridRows row blocks: a frame of more than 4096 candidates takes a second pass over the rows below)
  for (int rb = (int)blockIdx.x; rb * 64 < n; rb += kMatGridRows)
  for (int cq = rb / kHalfWords; cq < quads; cq++) {  // (column squares left of the row's own are below the diagonal)
      const int cg = (int)blockIdx.y + kMatColGroups * cq;
      if (cg * 4 + 3 < rb || cg * 256 >= n) continue;  // below the diagonal / past the end (block-uniform)
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
      if (cb >= rb && cb * 64 < n) {
        unsigned long long word = 0ull;
        const int jn = min(64, n - cb * 64);
        for (int b = 0; b < jn; b++) {
          const int j = cb * 64 + b;
          const bool hit = iou_exceeds(s_col[wave][b], s_area[wave][b], bi, max_iou);  // (wave-uniform j: the ballot inside sees all rows)
          word |= (hit && j > i) ? (1ull << b) : 0ull;
        }
        if (i < n) mat[((size_t)frame * mat_rows(K) + i) * kMatWords + cb] = word;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // own reads of s_col before the next column half's stores
  }
}

// Greedy pass over the matrix, exactly the reference's order (nn.rs:198-224): candidate i is
// selected iff no selected earlier candidate has its bit set.  One wave walks the 64-row blocks of a
// HALF of the matrix (rows row0 .. row0 + 2047, words word0 .. word0 + 31: the square on the diagonal);
// the removed-mask is spread over the lanes (lanes l and l + 32 both hold word word0 + l), a block's
// rows arrive two per load instruction (lane = row parity x word) and are prefetched one block
// ahead in registers; no LDS, no barrier.  Returns the number of selected candidates so far.
__device__ __forceinline__ int nms_scan_half(const unsigned long long* __restrict__ fmat, int n, int row0, int word0,
                                             unsigned long long removed, const unsigned long long* __restrict__ fkeys,
                                             const float4* __restrict__ fsp, Det* __restrict__ fd, uint32_t det_stride, int nsel,
                                             unsigned long long* s_kept /* kept mask per row block, or null */,
                                             int* s_progress /* row blocks published in s_kept */) {
  static_assert(kHalfWords == 32, "lane = (row parity, word) layout");
  const int lane = threadIdx.x & 63, half = lane >> 5, word = lane & 31;
  const int nl = min(n - row0, kMatHalf);  // rows of this half
  const int nrb = (nl + 63) >> 6;
  unsigned long long rows[32], rows_next[32], diag, diag_next;
  // rows 2q + half of block rb, word `word`; the diagonal word of row `lane` (rows past n and words
  // below the diagonal were never written: loaded from clamped rows and masked where they are used)
  auto load_block = [&](int rb, unsigned long long (&r)[32], unsigned long long& d) {
    const int rbc = min(rb, nrb - 1);
#pragma unroll
    for (int q = 0; q < 32; q++) r[q] = fmat[(size_t)(row0 + min(rbc * 64 + 2 * q + half, nl - 1)) * kMatWords + word0 + word];
    d = fmat[(size_t)(row0 + min(rbc * 64 + lane, nl - 1)) * kMatWords + word0 + rbc];
  };
  load_block(0, rows, diag);
  for (int rb = 0; rb < nrb; rb++) {
    load_block(rb + 1, rows_next, diag_next);
    const int cnt = min(64, nl - rb * 64);
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
    if (s_kept && lane == 0) {  // the folding waves pick the block up while this wave goes on
      s_kept[rb] = kept;
      __hip_atomic_store(s_progress, rb + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
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
        const float4 cc = fsp[row0 + rb * 64 + lane];
        Det dd;
        dd.x_tl = cc.x, dd.y_tl = cc.y, dd.x_br = cc.z, dd.y_br = cc.w, dd.conf = key_conf(fkeys[row0 + rb * 64 + lane]);
        fd[pos] = dd;
      }
    }
    nsel += __popcll(kept);
#pragma unroll
    for (int q = 0; q < 32; q++) rows[q] = rows_next[q];
    diag = diag_next;
  }
  return nsel;
}

// One workgroup per heavy frame.  Up to kMatHalf candidates: wave 0 scans the one square, the other waves leave at once.
// More (two to four squares on the diagonal): wave 0 scans square q and publishes, block by block, which rows it kept; waves
// 1..3 follow it and OR the kept rows' words of the column squares to the RIGHT of q (the off-diagonal squares: bulk loads,
// no serial chain) into the masks the later squares start from; then wave 0 scans square q + 1.
__global__ __launch_bounds__(256) void k_nms_scan(const unsigned long long* __restrict__ gkeys, size_t key_stride,
                                                 const float4* __restrict__ spill, int K, const unsigned long long* __restrict__ mat,
                                                 Det* __restrict__ dets, uint32_t det_stride, uint32_t* __restrict__ ndet) {
  constexpr int QMAX = kMatMax / kMatHalf;  // 4
  __shared__ unsigned long long s_kept[kHalfWords];
  __shared__ unsigned long long s_hi[3][QMAX - 1][kHalfWords];  // [follower wave][later square][word]
  __shared__ unsigned long long s_removed[QMAX][kHalfWords];    // mask every square starts from
  __shared__ int s_progress;
  const int frame = blockIdx.x;
  const uint32_t flag = ndet[frame];
  if (!(flag & kHeavyFlag)) return;
  const int n = (int)(flag & ~kHeavyFlag);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, word = lane & 31;
  const int Q = (n + kMatHalf - 1) / kMatHalf;
  if (Q == 1 && wave > 0) return;  // (no barrier on this path)
  const unsigned long long* fmat = mat + (size_t)frame * mat_rows(K) * kMatWords;
  const unsigned long long* fkeys = gkeys + (size_t)frame * key_stride;
  const float4* fsp = spill + (size_t)frame * K;
  Det* fd = dets + (size_t)frame * det_stride;
  if (Q == 1) {
    const int nsel = nms_scan_half(fmat, n, 0, 0, 0ull, fkeys, fsp, fd, det_stride, 0, nullptr, nullptr);
    if (lane == 0) ndet[frame] = (uint32_t)nsel;
    return;
  }
  for (int i = threadIdx.x; i < QMAX * kHalfWords; i += 256) s_removed[i / kHalfWords][i % kHalfWords] = 0ull;
  int nsel = 0;
  for (int q = 0; q < Q; q++) {
    if (threadIdx.x == 0) s_progress = 0;
    __syncthreads();
    const bool more = q + 1 < Q;  // squares to the right of this one
    if (wave == 0) {
      nsel = nms_scan_half(fmat, n, q * kMatHalf, q * kHalfWords, s_removed[q][word], fkeys, fsp, fd, det_stride, nsel,
                           more ? s_kept : nullptr, &s_progress);
    } else if (more) {
      // wave w: row blocks w - 1, w + 2, ... of square q (all 32: it is not the last); lane = (row parity, word of a later square)
      unsigned long long acc[QMAX - 1];
#pragma unroll
      for (int p = 0; p < QMAX - 1; p++) acc[p] = 0ull;
      for (int rb = wave - 1; rb < kHalfWords; rb += 3) {
        bool waited = false;
#pragma unroll
        for (int p = 0; p < QMAX - 1; p++) {
          const int sq = q + 1 + p;  // the later square
          if (sq >= Q) break;
          const int nwp = (min(n - sq * kMatHalf, kMatHalf) + 63) >> 6;  // words that square uses
          // all 32 row pairs of the block in flight at once (unconditional loads, selected afterwards: one memory round
          // trip per block and square); they do not depend on what wave 0 keeps, so they are issued before the wait
          unsigned long long v[32];
#pragma unroll
          for (int r = 0; r < 32; r++)
            v[r] = fmat[(size_t)(q * kMatHalf + rb * 64 + 2 * r + half) * kMatWords + sq * kHalfWords + min(word, nwp - 1)];
          if (!waited) {
            while (__hip_atomic_load(&s_progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= rb) __builtin_amdgcn_s_sleep(8);
            waited = true;
          }
          const unsigned long long kh = s_kept[rb] >> half;
#pragma unroll
          for (int r = 0; r < 32; r++) acc[p] |= ((kh >> (2 * r)) & 1ull) ? v[r] : 0ull;
        }
      }
#pragma unroll
      for (int p = 0; p < QMAX - 1; p++) {
        const int sq = q + 1 + p;
        if (sq >= Q) break;
        const int nwp = (min(n - sq * kMatHalf, kMatHalf) + 63) >> 6;
        unsigned long long a = acc[p];
        const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)a, 32), ohi = (uint32_t)__shfl_xor((int)(uint32_t)(a >> 32), 32);
        a |= ((unsigned long long)ohi << 32) | olo;
        if (half == 0) s_hi[wave - 1][p][word] = word < nwp ? a : 0ull;
      }
    }
    __syncthreads();
    if (more) {
      for (int i = threadIdx.x; i < (Q - 1 - q) * kHalfWords; i += 256) {
        const int p = i / kHalfWords, w = i % kHalfWords;
        s_removed[q + 1 + p][w] |= s_hi[0][p][w] | s_hi[1][p][w] | s_hi[2][p][w];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) ndet[frame] = (uint32_t)nsel;
}

}  // namespace

void launch_head_decode(const HeadArgs& h, const float* d_priors, uint32_t B, float min_conf, float* d_scores,
                        float* d_boxes, unsigned long long* d_keys, size_t key_stride, uint32_t* d_counts,
                        hipStream_t s) {
  const int K = h.base[4];
  ufd_launch(k_head_decode, dim3((K + 255) / 256, B), dim3(256), 0, s, h, d_priors, K, min_conf, d_scores,
                     d_boxes, d_keys, key_stride, d_counts);
}

void launch_threshold(const float* d_scores, uint32_t K, uint32_t B, float min_conf, unsigned long long* d_keys,
                      size_t key_stride, uint32_t* d_counts, hipStream_t s) {
  ufd_launch(k_threshold, dim3((K + 255) / 256, B), dim3(256), 0, s, d_scores, (int)K, min_conf, d_keys,
                     key_stride, d_counts);
}

void launch_sort_nms(unsigned long long* d_keys, size_t key_stride, uint32_t* d_counts, const float* d_boxes,
                     uint32_t K, float max_iou, Det* d_dets, uint32_t det_stride, uint32_t* d_ndet, float4* d_sel_spill,
                     unsigned long long* d_mat, uint32_t B, hipStream_t s, const NmsHostOut& host_out) {
  const int knob = kMatMin;
  const bool use_matrix = d_mat != nullptr && K > (uint32_t)knob;
  // (without the matrix scratch no frame is "heavy": mat_min = mat_max = 0 keeps everything in the first kernel)
  ufd_launch(k_sort_nms, dim3(B), dim3(1024), 0, s, d_keys, key_stride, d_counts, d_boxes, (int)K, max_iou,
                     d_dets, det_stride, d_ndet, d_sel_spill, use_matrix ? knob : 0, use_matrix ? kMatMax : 0,
                     use_matrix ? NmsHostOut() : host_out);  // (a frame the matrix path finishes has no results yet)
  if (!use_matrix) return;
  // (both return at once for frames the first kernel finished itself)
  ufd_launch(k_nms_matrix, dim3(kMatGridRows, kMatColGroups, B), dim3(256), 0, s, d_ndet, d_sel_spill, (int)K, max_iou, d_mat);
  ufd_launch(k_nms_scan, dim3(B), dim3(256), 0, s, d_keys, key_stride, d_sel_spill, (int)K, d_mat, d_dets, det_stride,
                     d_ndet);
}
size_t nms_matrix_bytes(uint32_t B, uint32_t K) { return (size_t)B * mat_rows((int)K) * kMatWords * sizeof(unsigned long long); }

}  // namespace ufd
