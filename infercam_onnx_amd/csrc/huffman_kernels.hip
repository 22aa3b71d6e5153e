// huffman_kernels.hip -- entropy-decoding half of row A1 on the device (the part of
// turbojpeg::decompress_image, infer_server/src/inferer.rs:35, that libjpeg-turbo does in
// jdhuff.c) for baseline single-scan streams, with or without restart markers.  Serial
// bit-twiddling per lane, latency-bound integer work in short launches that run beside the other
// context's convolution kernels.  The coefficient slab is zeroed first; lanes store only non-zero
// coefficients.
#include "kernels.hpp"

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <type_traits>

namespace ufd {
namespace {

// ---------------------------------------------------------------------------------------------
// Self-synchronising parallel decoder for streams WITHOUT restart markers (the common camera
// case).  A Huffman bit stream has no random access, but a decoder started at a wrong bit offset
// or in a wrong state falls into step with the true symbol sequence after a while.  The stream of
// every frame is cut into subsequences of >= 64 bytes (>= 32 in a batch of a few frames); the decoder state at a subsequence boundary
// is (bit position of the first symbol that starts inside, block inside the MCU, zigzag index).
// Per subsequence the kernels keep a small cache of (entry state -> exit state, MCUs completed)
// pairs, filled by speculation, and then look the true chain up in it:
//   k_huff_unstuff  one workgroup per frame: copies the entropy-coded segment to a scratch stream
//                   with the 0xFF00 byte stuffing removed (block scan over 16-byte pieces), so that
//                   bit offsets are plain arithmetic;
//   k_huff_seed     one lane per (subsequence, hypothesis c): decodes the subsequence, storing
//                   nothing, from "block c of the MCU starts exactly at the boundary".  Trying every
//                   c matters: a decoder with the wrong block index uses the chroma tables on luma
//                   blocks and keeps falling out of step (measured: 12 rounds to settle a 640x480
//                   frame with one hypothesis, 3 with one per block of the MCU);
//   k_huff_extend   (twice) every exit state cached for subsequence i-1 that subsequence i has not
//                   seen as an entry yet is decoded and appended: hypotheses that have fallen into
//                   step merge with the true chain here;
//   k_huff_resolve  one workgroup per frame: the cached pairs of subsequence i form a function
//                   f_i: slot -> slot of i+1 (16 nibbles in a uint64); a block scan under function
//                   composition yields the slot of the true chain in every subsequence at once.  A
//                   miss (the true state was never speculated) is decoded on the spot and the scan
//                   repeated, so the result is exact for any input; then a scan of the MCU counts
//                   gives every subsequence its first MCU;
//   k_huff_write    one lane per subsequence decodes it a last time from its true entry state and
//                   stores the coefficients, DC as differences;
//   k_dc_prefix     DC differences -> DC values, on a compact side array (one int16 per block) that k_idct reads.
// Integer, latency-bound work spread over the whole chip in short launches.
constexpr int kSyncThreads = 1024;
constexpr int kSyncMinBytes = 64;  // subsequence length written for a frame without a scan (the host picks the real one, >= 32 or >= 64 bytes by batch size: model.cpp, sub_floor)

struct SyncState {
  uint32_t p;   // bit position of the next symbol in the compacted stream
  uint32_t cz;  // block inside the MCU | zigzag position << 8
};

struct SyncStepTabs {       // state-only passes (SyncLutImage::step_dc, step_ac): bits consumed (code + magnitude) | zigzag advance << 5
  uint16_t dc[2][1024];
  uint32_t ac[2][1024];     // low half: the symbol; high half: the symbol and the one behind it, or 0
};
struct SyncSymTabs {        // write pass (SyncLutImage::fast): (code length << 8) | symbol; 0 = code longer than 10 bits
  uint16_t tab[4][1024];
};
template <bool kWrite>
struct SyncTablesT {
  // (head and table: copied from the prebuilt per-table-set image, load_sync_tables)
  HuffSlow slow[4];
  typename std::conditional<kWrite, SyncSymTabs, SyncStepTabs>::type t;
  uint32_t blk[12];        // per block of the MCU: comp | bx << 8 | by << 12 | dc << 16 | ac << 20
  uint32_t dc_bits, ac_bits;  // bit c = table slot (0/1) of block c of the MCU
  // coefficient offset of block c of MCU (mx, my): blk_base[c] + mx * blk_dx[c] + my * blk_dy[c]
  uint32_t blk_base[12], blk_dx[12], blk_dy[12];
  int mcux;
};
using SyncTables = SyncTablesT<false>;
using WriteTables = SyncTablesT<true>;

// Bit window over the unstuffed stream: n valid bits at the top of acc, one aligned word in flight.
struct BitWindow {
  const uint32_t* words;
  uint64_t acc;
  uint32_t nextw, wi;
  int n;
  __device__ __forceinline__ void init(const uint32_t* w, uint32_t pos) {
    words = w;
    wi = pos >> 5;
    const int sh = (int)(pos & 31);
    acc = (((uint64_t)__builtin_bswap32(words[wi]) << 32) | __builtin_bswap32(words[wi + 1])) << sh;
    n = 64 - sh;
    nextw = words[wi + 2];  // kept raw: the byte swap happens at use, so the load stays in flight
    wi += 3;
  }
  __device__ __forceinline__ void refill() {
    if (n < 32) {
      acc |= (uint64_t)__builtin_bswap32(nextw) << (32 - n);
      n += 32;
      nextw = words[wi++];
    }
  }
  __device__ __forceinline__ uint32_t top() const { return (uint32_t)(acc >> 32); }
  __device__ __forceinline__ void skip(int k) {
    acc <<= k;
    n -= k;
  }
};

// Code word longer than the 10-bit lookup: (length << 8) | symbol, or 0 if the bits are no code
// word.  Some lane of a wave is here in most iterations when hypotheses decode out of step, so the
// search over the lengths is not a serial loop of dependent LDS reads: maxcode[l] (exclusive bound
// of the code space used by lengths <= l, left-justified) grows with l, hence
// length = 11 + #{l in 11..16 : code >= maxcode[l]}, with all bounds and offsets read at once.
__device__ __forceinline__ int slow_symbol(const HuffSlow& t, uint32_t top) {
  const int code = (int)(top >> 16);
  int mc[6], dl[6];
#pragma unroll
  for (int q = 0; q < 6; q++) mc[q] = t.maxcode[11 + q], dl[q] = t.delta[11 + q];
  int l = 11, d = dl[0];
#pragma unroll
  for (int q = 0; q < 5; q++)
    if (code >= mc[q]) l = 12 + q, d = dl[q + 1];
  if (code >= mc[5]) return 0;
  return (l << 8) | t.sym[((code >> (16 - l)) + d) & 0xFF];
}

// Decodes every symbol that starts in [st.p, limit), tracking the state only, and leaves the exit
// state in st; returns the number of MCUs those symbols complete.  One symbol per iteration for
// every lane; a code word is <= 16 bits and its magnitude <= 15 bits, so everything happens in the
// top 32 bits of the window.
// CHECKPOINTS (cp != null): the subsequence starts at bit sub_start and is cut into kWriteParts parts of part_bits; the
// state at the first symbol that starts at or behind part boundary k (exactly what an entry state is for a subsequence)
// and the MCUs completed in front of it go to cp[k-1] / cpn[k-1].  k_huff_write then decodes every part of the true
// chain with a lane of its own: its serial symbol chain is a quarter as long.  Boundaries the walk never reaches
// (limit inside the subsequence, entry behind a boundary) get the state the walk is in when it passes / ends: the
// part's lane then starts at or behind its own limit and does nothing.
constexpr int kWriteParts = 4;
__device__ __forceinline__ int sync_span(const uint32_t* words, SyncState& st, uint32_t limit, const SyncTables& T, int bpm,
                                         uint32_t sub_start = 0, uint32_t part_bits = 0, uint2* cp = nullptr, int* cpn = nullptr) {
  // The walk is one dependent chain per lane: table entry -> bits consumed -> shifted window -> next table entry.  The
  // choice of the NEXT symbol's table (zigzag advance -> block end? -> next block -> its DC / AC table -> address) is kept
  // off that chain: the lookup of the next symbol is issued for BOTH candidates as soon as the window has moved -- the AC
  // table of the current block and the DC table of the next block, neither of which depends on the symbol being decoded
  // -- and the block-end test, resolved meanwhile, picks one.  (Measured with s_memtime: ~545 shader cycles per symbol
  // for this 45-instruction loop with the SIMD almost to itself, 3 % less than with the table choice on the chain: the
  // four compare -> exec-mask -> branch round trips per iteration weigh more than the data dependences.)
  int nmcu = 0;
  uint32_t pos = st.p;
  int c = (int)(st.cz & 0xFF), z = (int)(st.cz >> 8);
  BitWindow bw;
  bw.init(words, pos);
  const uint32_t dc_bits = T.dc_bits, ac_bits = T.ac_bits;
  int cur_t = z == 0 ? (int)((dc_bits >> c) & 1) : 2 + (int)((ac_bits >> c) & 1);
  int k = 1;                                                    // next boundary
  uint32_t next_cp = cp ? sub_start + part_bits : 0xFFFFFFFFu;  // (no checkpoints: never reached)
  // Two symbols per look-up (round 6): an AC entry's high half covers the symbol AND the one behind it.  It is taken when the
  // first of the two cannot end the block (zigzag index below 48: a run is at most 16) and both START in front of the
  // walk's limit and of the next checkpoint boundary -- so every state the walk is asked for (exit state, checkpoints, MCU
  // counts) is the one single steps give; pos + bits of both <= lim2 is sufficient for that.
  uint32_t lim2 = min(limit, next_cp);
  // tables of the candidates: AC table of block c, DC table of the block behind it (and that block's index)
  int cn = c + 1 == bpm ? 0 : c + 1;
  const uint32_t* tab_ac = T.t.ac[(int)((ac_bits >> c) & 1)];
  const uint16_t* tab_dcn = T.t.dc[(int)((dc_bits >> cn) & 1)];
  uint32_t top = bw.top();  // (init leaves >= 33 valid bits)
  uint32_t e = cur_t < 2 ? (uint32_t)T.t.dc[cur_t][top >> 22] : (T.t.ac[cur_t - 2][top >> 22] & 0xFFFFu);
  while (pos < limit) {
    if (__builtin_expect(pos >= next_cp, 0)) {  // the symbol about to be decoded is the first one at / behind boundary k
      cp[k - 1] = make_uint2(pos, (uint32_t)c | ((uint32_t)z << 8));
      cpn[k - 1] = nmcu;
      k++;
      next_cp = k < kWriteParts ? next_cp + part_bits : 0xFFFFFFFFu;
      lim2 = min(limit, next_cp);
      continue;  // (the entry may lie behind several boundaries)
    }
    if (__builtin_expect(e == 0, 0)) {
      const int ls = slow_symbol(T.slow[cur_t], top);
      // no code word (only reachable out of step or on corrupt data): consume one bit as EOB
      e = ls ? (uint32_t)sync_step(ls >> 8, ls & 0xFF, cur_t < 2) : (uint32_t)sync_step(1, 0, cur_t < 2);
    }
    const int adv = (int)(e & 31);
    bw.skip(adv);
    pos += (uint32_t)adv;
    bw.refill();
    top = bw.top();
    // both candidates for the next symbol's entry, in flight while the state catches up
    const uint32_t e_ac = tab_ac[top >> 22], e_dc = tab_dcn[top >> 22];
    z += (int)(e >> 5);
    const bool be = z >= 64;  // block finished (some lane of the wave is here in almost every iteration: selects, no branch)
    nmcu += (be && cn == 0) ? 1 : 0;
    const uint32_t e_two = e_ac >> 16;
    const bool two = e_two != 0 && z < 48 && pos + (e_two & 31) <= lim2;
    e = be ? e_dc : (two ? e_two : (e_ac & 0xFFFFu));
    cur_t = be ? (int)((dc_bits >> cn) & 1) : 2 + (int)((ac_bits >> c) & 1);
    c = be ? cn : c;
    z = be ? 0 : z;
    // the candidates of the symbol after that (they change at a block end only)
    cn = c + 1 == bpm ? 0 : c + 1;
    tab_ac = T.t.ac[(int)((ac_bits >> c) & 1)];
    tab_dcn = T.t.dc[(int)((dc_bits >> cn) & 1)];
  }
  if (cp)
    for (; k < kWriteParts; k++) {  // boundaries behind the end of the walk
      cp[k - 1] = make_uint2(pos, (uint32_t)c | ((uint32_t)z << 8));
      cpn[k - 1] = nmcu;
    }
  st.p = pos;
  st.cz = (uint32_t)c | ((uint32_t)z << 8);
  return nmcu;
}

// Same walk from the true entry state, storing the coefficients: AC in ZIGZAG order (the IDCT
// kernel undoes it), DC differences into the compact side array (k_dc_prefix sums them).  Zero coefficients are not stored
// (the slab is pre-zeroed).
__device__ __forceinline__ void write_span(const uint32_t* words, SyncState st, uint32_t limit, const WriteTables& T, int bpm,
                                           int16_t* fcoef, int16_t* fdc, int mcu, int total_mcus, bool* bad) {
  uint32_t pos = st.p;
  int c = (int)(st.cz & 0xFF), z = (int)(st.cz >> 8);
  if (mcu >= total_mcus) return;  // trailing pad bits only
  BitWindow bw;
  bw.init(words, pos);
  const uint32_t dc_bits = T.dc_bits, ac_bits = T.ac_bits;
  int cur_t = z == 0 ? (int)((dc_bits >> c) & 1) : 2 + (int)((ac_bits >> c) & 1);
  int my = mcu / T.mcux, mx = mcu - my * T.mcux;
  int16_t* blk = fcoef + T.blk_base[c] + mx * T.blk_dx[c] + (size_t)my * T.blk_dy[c];
  while (pos < limit) {
    bw.refill();
    const uint32_t top = bw.top();
    int e = T.t.tab[cur_t][top >> 22];
    if (__builtin_expect(e == 0, 0)) {
      e = slow_symbol(T.slow[cur_t], top);
      if (!e) e = 1 << 8, *bad = true;
    }
    const int len = e >> 8, sz = e & 15, run = (e >> 4) & 15;
    const uint32_t m = top << len;
    const int raw = (int)((m >> 1) >> (31 - sz));
    const int val = raw < ((1 << sz) >> 1) ? raw - (1 << sz) + 1 : raw;
    const int adv = len + sz;
    bw.skip(adv);
    pos += (uint32_t)adv;
    if (z == 0) {
      fdc[(blk - fcoef) >> 6] = (int16_t)val;  // DC difference of the block, in the compact side array (k_dc_prefix sums it)
      if ((e & 0xFF) > 15) *bad = true;
      z = 1;
    } else if (sz) {
      const int k = z + run;
      if (k > 63) *bad = true;
      else blk[k] = (int16_t)val;
      z = k + 1;
    } else {
      z = run == 15 ? z + 16 : 64;  // ZRL or EOB
    }
    cur_t = 2 + (int)((ac_bits >> c) & 1);
    if (z >= 64) {  // block finished
      z = 0;
      if (++c == bpm) {
        c = 0;
        if (++mx == T.mcux) mx = 0, my++;
        if (++mcu >= total_mcus) break;
      }
      cur_t = (int)((dc_bits >> c) & 1);
      blk = fcoef + T.blk_base[c] + mx * T.blk_dx[c] + (size_t)my * T.blk_dy[c];
    }
  }
}

// exclusive scan of one int per thread over the 1024-thread block; returns the block total in *total
// inclusive scan over the wave on the vector ALU: four row shifts (a DPP row is 16 lanes; lanes without a source add 0), then
// lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3 -- six instructions with the shift folded into the add
// (as __shfl_up steps: six ds_bpermute round trips through the LDS pipe plus a compare and a select each)
__device__ __forceinline__ int wave_inscan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112 /*row_shr:2*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114 /*row_shr:4*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118 /*row_shr:8*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142 /*row_bcast:15*/, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143 /*row_bcast:31*/, 0xc, 0xf, false);
  return v;
}
__device__ __forceinline__ int block_exscan(int v, int* s_wave /*16*/, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int inc = wave_inscan(v);
  __syncthreads();  // s_wave reuse
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kSyncThreads / 64; w++) {
    const int x = s_wave[w];
    if (w < wave) base += x;
    tot += x;
  }
  *total = tot;
  return base + inc - v;
}

// three scans at once (k_dc_prefix: one per component): one pair of barriers instead of three
__device__ __forceinline__ void block_exscan3(const int (&v)[3], int (*s_wave)[kSyncThreads / 64] /*[3][16]*/, int (&pre)[3], int (&total)[3]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc[3];
#pragma unroll
  for (int c = 0; c < 3; c++) inc[c] = wave_inscan(v[c]);
  __syncthreads();  // s_wave reuse
  if (lane == 63) s_wave[0][wave] = inc[0], s_wave[1][wave] = inc[1], s_wave[2][wave] = inc[2];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 3; c++) {
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kSyncThreads / 64; w++) {
      const int x = s_wave[c][w];
      if (w < wave) base += x;
      tot += x;
    }
    total[c] = tot;
    pre[c] = base + inc[c] - v[c];
  }
}

constexpr int kSyncLaneThreads = 256;  // seed / extend / write: one lane per (subsequence, hypothesis)
constexpr int kHypSlots = 16;          // cached (entry -> exit) pairs per subsequence; nibble 15 = "not cached"
constexpr int kHypAppendMax = 14;      // the speculation rounds fill slots 0..13, k_huff_resolve may use 14

static_assert(sizeof(HuffSlow) % 16 == 0 && offsetof(SyncLutImage, step_dc) == 4 * sizeof(HuffSlow) &&
                  offsetof(SyncLutImage, step_ac) == offsetof(SyncLutImage, step_dc) + sizeof(SyncLutImage::step_dc) &&
                  offsetof(SyncLutImage, fast) == offsetof(SyncLutImage, step_ac) + sizeof(SyncLutImage::step_ac) &&
                  offsetof(SyncTables, t) == offsetof(SyncLutImage, step_dc) && offsetof(SyncTables, blk) == offsetof(SyncLutImage, fast) &&
                  offsetof(WriteTables, t) == offsetof(SyncLutImage, step_dc) &&
                  offsetof(WriteTables, blk) == offsetof(WriteTables, t) + sizeof(SyncLutImage::fast) && sizeof(SyncStepTabs) % 16 == 0,
              "image layout");

// Fills the block's tables: the head of the table-set image and its step (or symbol) tables with
// wide loads that are all in flight at once (a load -> store loop pays one memory round trip per
// iteration), the per-frame layout from the scan and frame descriptors.
template <bool kWrite>
__device__ __forceinline__ void load_sync_tables(SyncTablesT<kWrite>& T, const HuffScan& sc, const SyncLutImage* __restrict__ images,
                                                 const JpegFrameDesc* d, int tid, int nthreads) {
  constexpr int kHead = (int)(offsetof(SyncLutImage, step_dc) / 16);
  constexpr int kSkip = kWrite ? (int)(sizeof(SyncStepTabs) / 16) : 0;  // the write pass takes `fast` in place of the step tables
  constexpr int kVec = kHead + (int)((kWrite ? sizeof(SyncLutImage::fast) : sizeof(SyncStepTabs)) / 16);
  const uint4* src = reinterpret_cast<const uint4*>(images + sc.lut_base / 4);
  uint4* dst = reinterpret_cast<uint4*>(&T);
  if (nthreads == kSyncLaneThreads) {
    constexpr int kPer = (kVec + kSyncLaneThreads - 1) / kSyncLaneThreads;
    // (unconditional loads from clamped indices, predicated stores: with the load under the predicate as well the array
    // went to SCRATCH memory and every piece became load -> wait -> scratch store ... scratch load -> wait -> LDS store)
    uint4 r[kPer];
#pragma unroll
    for (int q = 0; q < kPer; q++) {
      const int i = min(tid + q * kSyncLaneThreads, kVec - 1);
      r[q] = src[i < kHead ? i : i + kSkip];
    }
#pragma unroll
    for (int q = 0; q < kPer; q++)
      if (tid + q * kSyncLaneThreads < kVec) dst[tid + q * kSyncLaneThreads] = r[q];
  } else {
    for (int i = tid; i < kVec; i += nthreads) dst[i] = src[i < kHead ? i : i + kSkip];
  }
  if (tid < 12) {
    T.blk[tid] = sc.blk_comp[tid] | (sc.blk_bx[tid] << 8) | (sc.blk_by[tid] << 12) | (sc.blk_dc[tid] << 16) |
                 (sc.blk_ac[tid] << 20);
    if (d) {
      const int comp = sc.blk_comp[tid];
      T.blk_base[tid] = d->coef_off[comp] + ((uint32_t)sc.blk_by[tid] * d->wblk[comp] + sc.blk_bx[tid]) * 64;
      T.blk_dx[tid] = (uint32_t)d->h[comp] * 64;
      T.blk_dy[tid] = (uint32_t)d->v[comp] * d->wblk[comp] * 64;
    }
  }
  if (tid == 32) {
    uint32_t db = 0, ab = 0;
    for (int c = 0; c < 12; c++) db |= (uint32_t)(sc.blk_dc[c] & 1) << c, ab |= (uint32_t)(sc.blk_ac[c] & 1) << c;
    T.dc_bits = db, T.ac_bits = ab;
    T.mcux = d ? d->mcux : 1;
  }
}

// Unstuffs every segment of the frame into its subsequence slots (segment q starts at slot
// ivs[q].first_sub) and fills the per-slot tables.  Two passes over the raw range with a block
// scan each: pass 1 counts the stuffed zeros in front of every segment begin / end, pass 2 stores
// the kept bytes at  slot offset + (position - begin) - (stuffed zeros since begin).
// Rows blockIdx.y > 0 of the grid clear the frame's coefficient slab (the decoder stores non-zero coefficients only) and
// its DC side array beside the unstuffing: independent work that used to be a launch of its own in front of the chain.
struct ZeroArgs {
  int16_t* coef;
  size_t coef_stride;
  uint32_t vec_per_frame;     // 16-byte stores per frame
  int16_t* dc;
  size_t dc_stride;
  uint32_t dc_vec_per_frame;
  uint32_t* status;           // per-frame decode status, cleared by the frame's unstuff block
};
__global__ __launch_bounds__(kSyncThreads) void k_huff_unstuff(const uint8_t* __restrict__ blob,
                                                               const HuffScan* __restrict__ scans,
                                                               const HuffInterval* __restrict__ ivs, SyncBuffers sb, ZeroArgs za) {
  __shared__ int s_wave[kSyncThreads / 64];
  __shared__ uint32_t s_beg[kSyncMaxSeg], s_end[kSyncMaxSeg], s_zb[kSyncMaxSeg], s_ze[kSyncMaxSeg];
  const int frame = blockIdx.x, tid = threadIdx.x;
  if (blockIdx.y > 0) {  // (whole block)
    const uint32_t rows = gridDim.y - 1, row = blockIdx.y - 1;
    uint4* dst = reinterpret_cast<uint4*>(za.coef + (size_t)frame * za.coef_stride);
    for (uint32_t v = row * kSyncThreads + tid; v < za.vec_per_frame; v += rows * kSyncThreads) dst[v] = make_uint4(0, 0, 0, 0);
    if (za.dc && (za.dc_stride & 7) == 0) {
      uint4* d = reinterpret_cast<uint4*>(za.dc + (size_t)frame * za.dc_stride);
      for (uint32_t v = row * kSyncThreads + tid; v < za.dc_vec_per_frame; v += rows * kSyncThreads) d[v] = make_uint4(0, 0, 0, 0);
    } else if (za.dc) {  // (frame bases not 16-byte aligned: tiny max_src sizes only)
      int16_t* d = za.dc + (size_t)frame * za.dc_stride;
      for (uint32_t v = row * kSyncThreads + tid; v < za.dc_vec_per_frame * 8; v += rows * kSyncThreads)
        if (v < za.dc_stride) d[v] = 0;
    }
    return;
  }
  if (tid == 0 && za.status) za.status[frame] = 0;
  const HuffScan& sc = scans[frame];
  SyncFrame* fr = sb.frames + frame;
  const int nseg = (int)sc.nseg;
  if (nseg == 0) {  // frame failed on the host side (uniform per block)
    if (tid == 0) fr->total_bits = 0, fr->nsub = 0, fr->sub_bits = kSyncMinBytes * 8;
    return;
  }
  const HuffInterval* seg = ivs + sc.seg_base;
  for (int q = tid; q < nseg; q += kSyncThreads) s_beg[q] = seg[q].begin, s_end[q] = seg[q].end, s_zb[q] = 0, s_ze[q] = 0;
  __syncthreads();
  const uint8_t* src = blob + sc.blob_off;  // 16-byte aligned
  uint8_t* dst = sb.stream + (size_t)frame * sb.stream_stride;
  const uint32_t lo = s_beg[0], hi = s_end[nseg - 1];
  const uint32_t sub_bytes = sc.sub_bytes;
  // first segment whose end lies behind `pos` (binary search in LDS)
  auto seg_behind = [&](uint32_t pos) {
    int a = 0, b = nseg;
    while (a < b) {
      const int mid = (a + b) >> 1;
      if (s_end[mid] > pos) b = mid; else a = mid + 1;
    }
    return a;
  };
  // (a single segment needs no boundary counts: its begin is the start of the range)
  for (int pass = nseg > 1 ? 0 : 1; pass < 2; pass++) {
    uint32_t zbase = 0;  // stuffed zeros in front of the tile
    for (uint32_t tile = lo & ~15u; tile <= hi; tile += kSyncThreads * 16) {
      const uint32_t off = tile + (uint32_t)tid * 16;
      uint4 v = make_uint4(0, 0, 0, 0);
      unsigned prev = 0;
      if (off < hi) {
        v = *reinterpret_cast<const uint4*>(src + off);
        if (off > lo) prev = src[off - 1];
      }
      const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
      uint32_t zmask = 0;  // stuffed zero bytes of the piece (0xFF 0x00 only occurs inside segments)
      unsigned pb = prev;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const unsigned bv = (w4[j >> 2] >> (8 * (j & 3))) & 0xFF;
        const uint32_t pos = off + j;
        if (pos >= lo && pos < hi && bv == 0 && pb == 0xFF) zmask |= 1u << j;
        pb = bv;
      }
      int total;
      const uint32_t myz = zbase + (uint32_t)block_exscan(__popc(zmask), s_wave, &total);
      if (off <= hi) {
        int q = seg_behind(off > 0 ? off - 1 : 0);  // segments ending at `off` still need their end count
        if (pass == 0) {
          for (; q < nseg && s_beg[q] < off + 16; q++) {
            const uint32_t bq = s_beg[q], eq = s_end[q];
            if (bq >= off) s_zb[q] = myz + __popc(zmask & ((1u << (bq - off)) - 1));
            if (eq >= off && eq < off + 16) s_ze[q] = myz + __popc(zmask & ((1u << (eq - off)) - 1));
          }
        } else if (q < nseg && s_beg[q] <= off && off + 16 <= s_end[q]) {
          // the whole piece lies inside one segment: running output pointer
          uint8_t* o = dst + seg[q].first_sub * sub_bytes + (off - s_beg[q]) - (myz - s_zb[q]);
          if (zmask == 0) {
            // no stuffed byte in the piece (15 pieces of 16): ONE 16-byte store to the -- arbitrarily aligned -- output
            // position (gfx950 executes unaligned global stores: tools/ubench/unaligned_store.hip) instead of 16 byte stores
            struct __attribute__((packed, aligned(1))) Piece { uint32_t w[4]; };
            const Piece pc = {{v.x, v.y, v.z, v.w}};
            *reinterpret_cast<Piece*>(o) = pc;
          } else {
#pragma unroll
            for (int j = 0; j < 16; j++)
              if (!(zmask & (1u << j))) *o++ = (uint8_t)((w4[j >> 2] >> (8 * (j & 3))) & 0xFF);
          }
        } else {
#pragma unroll 1
          for (int j = 0; j < 16; j++) {  // a piece with a segment boundary (or the edge of the range) in it
            const uint32_t pos = off + j;
            while (q < nseg && pos >= s_end[q]) q++;
            if (q >= nseg) break;
            if (pos < s_beg[q] || (zmask & (1u << j))) continue;
            const uint32_t o = seg[q].first_sub * sub_bytes + (pos - s_beg[q]) - (myz + __popc(zmask & ((1u << j) - 1)) - s_zb[q]);
            dst[o] = (uint8_t)((w4[j >> 2] >> (8 * (j & 3))) & 0xFF);
          }
        }
      }
      zbase += (uint32_t)total;
    }
    if (nseg == 1 && tid == 0) s_ze[0] = zbase;  // (s_zb[0] = 0)
    __syncthreads();
  }
  // per segment: zero slack behind the data; per slot (all threads): limit, flags, segment
  uint32_t* lim = sb.lim + (size_t)frame * kSyncMaxSub;
  uint16_t* sseg = sb.seg + (size_t)frame * kSyncMaxSub;
  for (int q = tid; q < nseg; q += kSyncThreads) {
    const uint32_t len = (s_end[q] - s_beg[q]) - (s_ze[q] - s_zb[q]);
    const uint32_t first = seg[q].first_sub;
    uint8_t* z = dst + (size_t)first * sub_bytes + len;
    for (int k = 0; k < 32; k++) z[k] = 0;  // the bit window reads a few words past the last symbol
    s_zb[q] = first;                         // (the counts are consumed: reuse the arrays)
    s_ze[q] = (first * sub_bytes + len) * 8;  // end of the segment's data in bits
  }
  __syncthreads();
  for (uint32_t i = tid; i < sc.nsub; i += kSyncThreads) {
    int a = 0, b = nseg;  // last segment whose first slot is <= i
    while (b - a > 1) {
      const int mid = (a + b) >> 1;
      if (s_zb[mid] <= i) a = mid; else b = mid;
    }
    const uint32_t first = s_zb[a], end_bits = s_ze[a];
    // bit 31: the entry state of the slot needs no search -- first slot of a segment (exact), or a
    // slot behind the segment's data (nothing to decode: stuffing and the slack shrink the data)
    const bool settled = i == first || i * sub_bytes * 8 >= end_bits;
    lim[i] = min((i + 1) * sub_bytes * 8, max(end_bits, i * sub_bytes * 8)) | (settled ? 0x80000000u : 0u);
    sseg[i] = (uint16_t)a;
  }
  if (tid == 0) fr->total_bits = sc.nsub * sub_bytes * 8, fr->sub_bits = sub_bytes * 8, fr->nsub = sc.nsub;
}

// The lanes of a block decode neighbouring subsequences: their part of the stream is staged in LDS
// with coalesced loads first.  (Per-lane global loads of one word at a time made every refill of the
// bit window a ~1-2 us round trip: 16 of them in a row dominated the pass.)
constexpr uint32_t kStageWords = 2048 + 16;  // 8 KB: a block of k_huff_seed spans 43 subsequences (2.7 KB at 64 bytes each); longer spans read the stream from global memory.  With the 18.6 KB of tables five blocks fit a CU -- at 32 KB of staging three did, k_huff_extend ran in two rounds and the blocks kept the convolutions of the other contexts off the CU's LDS (43.2 k -> 44.7 k frames/s)

// Stages the words of subsequences [i_lo, i_hi] plus the look-ahead tail; returns the pointer p with
// p[w] = word w of the stream (nullptr: the range does not fit, read global memory instead).
__device__ __forceinline__ const uint32_t* stage_stream(uint32_t* s_stream, const uint32_t* words, const SyncFrame& fr,
                                                        uint32_t i_lo, uint32_t i_hi, int tid, int nthreads) {
  const uint32_t sub_words = fr.sub_bits >> 5;
  const uint32_t w0 = i_lo * sub_words;
  const uint32_t w1 = min((i_hi + 1) * sub_words + 8, fr.total_bits / 32 + 8);
  if (w1 <= w0 || w1 - w0 > kStageWords) return nullptr;
  for (uint32_t j = tid; j < w1 - w0; j += nthreads) s_stream[j] = words[w0 + j];
  return s_stream - w0;
}

__global__ __launch_bounds__(kSyncLaneThreads) void k_huff_seed(const HuffScan* __restrict__ scans,
                                                                const SyncLutImage* __restrict__ luts, SyncBuffers sb,
                                                                uint8_t* __restrict__ cnt_out) {
  __shared__ SyncTables T;
  __shared__ uint32_t s_stream[kStageWords];
  const int frame = blockIdx.y, tid = threadIdx.x;
  const SyncFrame fr = sb.frames[frame];
  const HuffScan& sc = scans[frame];
  const int bpm = (int)sc.blocks_per_mcu;
  const uint32_t lane = blockIdx.x * kSyncLaneThreads + tid;
  if ((blockIdx.x * kSyncLaneThreads) / bpm >= fr.nsub) return;  // whole block past the end (uniform)
  load_sync_tables(T, sc, luts, nullptr, tid, kSyncLaneThreads);
  const uint32_t* words = reinterpret_cast<const uint32_t*>(sb.stream + (size_t)frame * sb.stream_stride);
  const uint32_t* staged = stage_stream(s_stream, words, fr, (blockIdx.x * kSyncLaneThreads) / bpm,
                                        min((blockIdx.x * kSyncLaneThreads + kSyncLaneThreads - 1) / bpm, fr.nsub - 1), tid,
                                        kSyncLaneThreads);
  __syncthreads();
  const uint32_t i = lane / bpm;
  const int g = (int)(lane - i * bpm);
  if (i >= fr.nsub) return;
  const uint32_t lim = sb.lim[(size_t)frame * kSyncMaxSub + i];
  const bool seg_start = (lim >> 31) != 0;  // first slot of a segment: the entry state is exact
  if (seg_start && g != 0) return;
  SyncState st;
  st.p = i * fr.sub_bits, st.cz = (uint32_t)g;
  const SyncState entry = st;
  const uint32_t limit = lim & 0x7FFFFFFFu;
  const size_t slot = ((size_t)frame * kSyncMaxSub + i) * kHypSlots + g;
  const uint32_t part_bits = fr.sub_bits / kWriteParts;
  uint2* cp = sb.cp + slot * (kWriteParts - 1);
  int* cpn = sb.cpn + slot * (kWriteParts - 1);
  const int nm = staged ? sync_span(staged, st, limit, T, bpm, i * fr.sub_bits, part_bits, cp, cpn)
                        : sync_span(words, st, limit, T, bpm, i * fr.sub_bits, part_bits, cp, cpn);
  sb.ent[slot] = make_uint2(entry.p, entry.cz);
  sb.ext[slot] = make_uint2(st.p, st.cz);
  sb.nm[slot] = nm;
  if (g == 0) cnt_out[(size_t)frame * kSyncMaxSub + i] = (uint8_t)(seg_start ? 1 : min(bpm, kHypAppendMax));
}

// 16 lanes per subsequence i >= 1: lane k owns the exit state cached in slot k of subsequence i-1.
// Two phases.  (A) every lane decides whether its candidate is new to subsequence i and which slot it gets (shuffles
// inside its 16-lane group).  Only one or two of a group's sixteen lanes come out with work, and a wave pays for the
// symbol loop in full however few of its lanes are active (4400 waves per batch with a handful of live lanes each when the
// walk was done in place).  So (B) the block's new candidates are COMPACTED into a list in LDS and walked by the block's
// first waves with every lane busy; the other waves leave: 46 -> 41 us per launch, one table fill per 64 subsequences.
constexpr int kExtendThreads = 1024;  // 64 subsequences per block: one table fill, ~100-200 walks
__global__ __launch_bounds__(kExtendThreads) void k_huff_extend(const HuffScan* __restrict__ scans,
                                                                const SyncLutImage* __restrict__ luts, SyncBuffers sb,
                                                                const uint8_t* __restrict__ cnt_in,
                                                                uint8_t* __restrict__ cnt_out) {
  __shared__ SyncTables T;
  __shared__ uint32_t s_stream[kStageWords];
  __shared__ uint2 s_cand[kExtendThreads];     // compacted work list: candidate state ...
  __shared__ uint32_t s_where[kExtendThreads]; // ... subsequence << 4 | slot
  __shared__ int s_wbase[kExtendThreads / 64 + 1];
  const int frame = blockIdx.y, tid = threadIdx.x;
  const SyncFrame fr = sb.frames[frame];
  const uint32_t lane = blockIdx.x * kExtendThreads + tid;
  if ((blockIdx.x * kExtendThreads) / kHypSlots >= fr.nsub) return;  // uniform
  const HuffScan& sc = scans[frame];
  load_sync_tables(T, sc, luts, nullptr, tid, kExtendThreads);
  const uint32_t* words = reinterpret_cast<const uint32_t*>(sb.stream + (size_t)frame * sb.stream_stride);
  const uint32_t* staged = stage_stream(s_stream, words, fr, (blockIdx.x * kExtendThreads) / kHypSlots,
                                        min((blockIdx.x * kExtendThreads + kExtendThreads - 1) / kHypSlots, fr.nsub - 1),
                                        tid, kExtendThreads);
  // ---- phase A
  const uint32_t i = lane / kHypSlots;
  const int k = (int)(lane % kHypSlots);
  const bool in_range = i < fr.nsub;
  const size_t fbase = (size_t)frame * kSyncMaxSub;
  // All five loads of the lane are issued together, from clamped (always allocated) slots, and masked afterwards: as
  // `x = cond ? load : c` chains they were four dependent memory round trips in a row (limit -> counts -> candidate -> entry).
  const uint32_t ic = min(i, fr.nsub - 1), ip = ic > 0 ? ic - 1 : 0;  // (fr.nsub >= 1: the block passed the test above)
  const uint32_t lim_raw = sb.lim[fbase + ic];
  const int nc_raw = cnt_in[fbase + ic], np_raw = cnt_in[fbase + ip];
  const uint2 cand_raw = sb.ext[(fbase + ip) * kHypSlots + k];
  const uint2 mine_raw = sb.ent[(fbase + ic) * kHypSlots + k];
  const uint32_t lim = in_range ? lim_raw : 0x80000000u;
  const bool seg_start = (lim >> 31) != 0;  // exact entry state: nothing to extend
  const int n_cur = in_range ? nc_raw : 0;
  const int n_prev = (in_range && !seg_start) ? np_raw : 0;
  // lane k holds candidate k (exit state cached in slot k of subsequence i-1) and entry k of
  // subsequence i; the 16 lanes of the group compare through shuffles
  const uint2 cand = k < n_prev ? cand_raw : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
  const uint2 mine = k < n_cur ? mine_raw : make_uint2(0xFFFFFFFEu, 0xFFFFFFFEu);
  bool fresh = k < n_prev;
  const int gbase = (tid & 63) & ~(kHypSlots - 1);
#pragma unroll
  for (int q = 0; q < kHypSlots; q++) {
    const uint32_t ex = __shfl(mine.x, gbase + q, 64), ey = __shfl(mine.y, gbase + q, 64);
    const uint32_t ox = __shfl(cand.x, gbase + q, 64), oy = __shfl(cand.y, gbase + q, 64);
    if (ex == cand.x && ey == cand.y) fresh = false;            // already cached (unused slots hold a sentinel)
    if (q < k && ox == cand.x && oy == cand.y) fresh = false;   // duplicate candidate: the lowest lane keeps it
  }
  const unsigned long long ball = __ballot(fresh);
  const uint32_t gmask = (uint32_t)(ball >> gbase) & 0xFFFFu;
  const int slot = n_cur + __popc(gmask & ((1u << k) - 1));
  if (k == 0 && in_range) cnt_out[fbase + i] = (uint8_t)min(n_cur + __popc(gmask), kHypAppendMax);
  const bool work = fresh && slot < kHypAppendMax;
  // compaction: position of this lane's item = items of the waves in front + items of the lower lanes of its wave
  const unsigned long long wball = __ballot(work);
  const int wv = tid >> 6, ln = tid & 63;
  if (ln == 0) s_wbase[wv] = __popcll(wball);
  __syncthreads();  // (also: tables and staged stream complete)
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kExtendThreads / 64; w++) {
    const int x = s_wbase[w];
    base += w < wv ? x : 0;
    total += x;
  }
  if (work) {
    const int at = base + __popcll(wball & ((1ull << ln) - 1ull));
    s_cand[at] = cand;
    s_where[at] = (i << 4) | (uint32_t)slot;
  }
  __syncthreads();
  // ---- phase B: one walk per lane, the lanes of the first waves all busy
  if (tid >= total) return;
  const uint2 c2 = s_cand[tid];
  const uint32_t wi = s_where[tid] >> 4;
  const int wslot = (int)(s_where[tid] & 15u);
  SyncState st;
  st.p = c2.x, st.cz = c2.y;
  const uint32_t limit = sb.lim[fbase + wi] & 0x7FFFFFFFu;
  const size_t o = (fbase + wi) * kHypSlots + wslot;
  const uint32_t part_bits = fr.sub_bits / kWriteParts;
  uint2* cp = sb.cp + o * (kWriteParts - 1);
  int* cpn = sb.cpn + o * (kWriteParts - 1);
  const int nm = staged ? sync_span(staged, st, limit, T, (int)sc.blocks_per_mcu, wi * fr.sub_bits, part_bits, cp, cpn)
                        : sync_span(words, st, limit, T, (int)sc.blocks_per_mcu, wi * fr.sub_bits, part_bits, cp, cpn);
  sb.ent[o] = c2;
  sb.ext[o] = make_uint2(st.p, st.cz);
  sb.nm[o] = nm;
}

// After the speculation rounds: map[i] = f_i, the function "slot of subsequence i -> slot of
// subsequence i+1 whose entry state equals that slot's exit state" as 16 nibbles (15 = not cached).
__global__ __launch_bounds__(kSyncLaneThreads) void k_huff_link(SyncBuffers sb, const uint8_t* __restrict__ cnt) {
  const int frame = blockIdx.y, tid = threadIdx.x;
  const SyncFrame fr = sb.frames[frame];
  const uint32_t lane = blockIdx.x * kSyncLaneThreads + tid;
  if ((blockIdx.x * kSyncLaneThreads) / kHypSlots >= fr.nsub) return;  // uniform
  const uint32_t i = lane / kHypSlots;
  const int k = (int)(lane % kHypSlots);
  const size_t fbase = (size_t)frame * kSyncMaxSub;
  const bool has_next = i + 1 < fr.nsub;
  const int n0 = has_next ? cnt[fbase + i] : 0, n1 = has_next ? cnt[fbase + i + 1] : 0;
  uint2 x = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu), e = make_uint2(0xFFFFFFFEu, 0xFFFFFFFEu);
  if (k < n0) x = sb.ext[(fbase + i) * kHypSlots + k];
  if (k < n1) e = sb.ent[(fbase + i + 1) * kHypSlots + k];
  const int gbase = (tid & 63) & ~(kHypSlots - 1);
  unsigned hit = 15;
#pragma unroll
  for (int q = 0; q < kHypSlots; q++) {
    const uint32_t ex = __shfl(e.x, gbase + q, 64), ey = __shfl(e.y, gbase + q, 64);
    if (ex == x.x && ey == x.y) hit = (unsigned)q;
  }
  // OR the 16 nibbles of the group together
  uint32_t lo = k < 8 ? hit << (4 * k) : 0, hi = k >= 8 ? hit << (4 * (k - 8)) : 0;
#pragma unroll
  for (int o = 1; o < kHypSlots; o <<= 1) {
    lo |= __shfl_xor(lo, o, 64);
    hi |= __shfl_xor(hi, o, 64);
  }
  if (k == 0 && i < fr.nsub) {
    unsigned long long f = has_next ? ((unsigned long long)hi << 32) | lo : 0xFEDCBA9876543210ull;
    if (has_next && (sb.lim[fbase + i + 1] >> 31)) f = 0;  // the next slot starts a segment: its slot 0 is exact
    sb.map[fbase + i] = f;
  }
}

// f: slot of subsequence i -> slot of subsequence i+1, 16 nibbles; (g o f)(k) = g(f(k))
__device__ __forceinline__ unsigned long long compose_slots(unsigned long long f, unsigned long long g) {
  unsigned long long r = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) r |= ((g >> (4 * ((f >> (4 * k)) & 15))) & 15ull) << (4 * k);
  return r;
}

constexpr unsigned long long kSlotIdentity = 0xFEDCBA9876543210ull;
constexpr unsigned long long kSlotAllMiss = 0xFFFFFFFFFFFFFFFFull;

__device__ __forceinline__ unsigned long long build_slot_map(const SyncBuffers& sb, size_t fbase, uint32_t i,
                                                              const uint8_t* cnt) {
  // exit states of subsequence i looked up among the entry states of subsequence i+1
  if (sb.lim[fbase + i + 1] >> 31) return 0;  // the next slot starts a segment: its slot 0 is exact
  unsigned long long f = kSlotAllMiss;
  const int n0 = cnt[fbase + i], n1 = cnt[fbase + i + 1];
  for (int k = 0; k < n0; k++) {
    const uint2 x = sb.ext[(fbase + i) * kHypSlots + k];
    int hit = 15;
    for (int j = 0; j < n1; j++) {
      const uint2 e = sb.ent[(fbase + i + 1) * kHypSlots + j];
      if (e.x == x.x && e.y == x.y) hit = j;
    }
    f = (f & ~(15ull << (4 * k))) | ((unsigned long long)hit << (4 * k));
  }
  return f;
}

__global__ __launch_bounds__(kSyncThreads) void k_huff_resolve(const HuffScan* __restrict__ scans,
                                                               const HuffInterval* __restrict__ ivs,
                                                               const SyncLutImage* __restrict__ luts, SyncBuffers sb,
                                                               uint8_t* __restrict__ cnt, uint32_t* __restrict__ status) {
  constexpr int kPer = kSyncMaxSub / kSyncThreads;  // subsequences per thread
  __shared__ SyncTables T;
  __shared__ unsigned long long s_f[kSyncThreads];
  __shared__ int s_wave[kSyncThreads / 64];
  __shared__ int s_miss;
  const int frame = blockIdx.x, tid = threadIdx.x;
  const SyncFrame fr = sb.frames[frame];
  const HuffScan& sc = scans[frame];
  if (sc.nseg == 0) return;
  const HuffInterval* seg = ivs + sc.seg_base;
  const int nsub = (int)fr.nsub;
  load_sync_tables(T, sc, luts, nullptr, tid, kSyncThreads);
  const size_t fbase = (size_t)frame * kSyncMaxSub;
  const uint32_t* words = reinterpret_cast<const uint32_t*>(sb.stream + (size_t)frame * sb.stream_stride);
  // slot maps of this thread's kPer consecutive subsequences: f[q] maps i = tid * kPer + q to i + 1
  unsigned long long f[kPer];
#pragma unroll
  for (int q = 0; q < kPer; q++) {
    const int i = tid * kPer + q;
    f[q] = i < nsub ? sb.map[fbase + i] : kSlotIdentity;
  }
  int slot[kPer];
  for (;;) {
    // inclusive scan under composition: pre = f[tid*kPer - 1] o ... o f[0]; the true chain starts in slot 0
    unsigned long long mine = f[0];
#pragma unroll
    for (int q = 1; q < kPer; q++) mine = compose_slots(mine, f[q]);
    unsigned long long inc = mine;
    __syncthreads();
    s_f[tid] = inc;
    __syncthreads();
    for (int o = 1; o < kSyncThreads; o <<= 1) {
      unsigned long long left = kSlotIdentity;
      if (tid >= o) left = s_f[tid - o];
      __syncthreads();
      if (tid >= o) inc = compose_slots(left, inc), s_f[tid] = inc;
      __syncthreads();
    }
    const unsigned long long pre = tid ? s_f[tid - 1] : kSlotIdentity;
    int t = (int)(pre & 15);  // slot of the true chain in subsequence tid * kPer
    int first_miss = 0x7FFFFFFF;
#pragma unroll
    for (int q = 0; q < kPer; q++) {
      const int i = tid * kPer + q;
      slot[q] = t;
      if (i < nsub && t == 15) first_miss = min(first_miss, i);
      t = t == 15 ? 15 : (int)((f[q] >> (4 * t)) & 15);
    }
    if (tid == 0) s_miss = 0x7FFFFFFF;
    __syncthreads();
    if (first_miss != 0x7FFFFFFF) atomicMin(&s_miss, first_miss);
    __syncthreads();
    const int m = s_miss;
    if (m == 0x7FFFFFFF) break;
    // the true entry state of subsequence m (>= 1) was never speculated: decode it now, exactly
    const int owner = (m - 1) / kPer;  // thread that knows the true slot of subsequence m - 1
    if (tid == owner) {
      const int tq = slot[(m - 1) % kPer];
      const uint2 x = sb.ext[(fbase + m - 1) * kHypSlots + tq];
      SyncState st;
      st.p = x.x, st.cz = x.y;
      const int n = cnt[fbase + m];
      const int dst = n < kHypSlots - 1 ? n : kHypSlots - 2;  // append, or recycle the last usable slot
      const size_t o = (fbase + m) * kHypSlots + dst;
      const int nm = sync_span(words, st, sb.lim[fbase + m] & 0x7FFFFFFFu, T, (int)sc.blocks_per_mcu, (uint32_t)m * fr.sub_bits,
                               fr.sub_bits / kWriteParts, sb.cp + o * (kWriteParts - 1), sb.cpn + o * (kWriteParts - 1));
      sb.ent[o] = x;
      sb.ext[o] = make_uint2(st.p, st.cz);
      sb.nm[o] = nm;
      if (n < kHypSlots - 1) cnt[fbase + m] = (uint8_t)(n + 1);
      __threadfence_block();
    }
    __syncthreads();  // (workgroup-scope fence: the new pair is visible to the block)
#pragma unroll
    for (int q = 0; q < kPer; q++) {
      const int i = tid * kPer + q;
      if ((i == m - 1 || i == m) && i + 1 < nsub) f[q] = build_slot_map(sb, fbase, (uint32_t)i, cnt);
    }
  }
  // true entry state and first MCU of every subsequence
  int my_nm[kPer], sum = 0;
#pragma unroll
  for (int q = 0; q < kPer; q++) {
    const int i = tid * kPer + q;
    my_nm[q] = 0;
    if (i < nsub) {
      const size_t o = (fbase + i) * kHypSlots + slot[q];
      sb.start[fbase + i] = sb.ent[o];
      sb.tslot[fbase + i] = (uint8_t)slot[q];  // k_huff_write finds the checkpoints of the true chain there
      my_nm[q] = sb.nm[o];
    }
    sum += my_nm[q];
  }
  // MCUs completed in front of every slot (scan over the frame), then relative to its segment
  int total;
  int first = block_exscan(sum, s_wave, &total);
  __shared__ int s_ex[kSyncMaxSub + 1];
#pragma unroll
  for (int q = 0; q < kPer; q++) {
    const int i = tid * kPer + q;
    if (i < nsub) s_ex[i] = first;
    first += my_nm[q];
  }
  if (tid == 0) s_ex[nsub] = total;
  __syncthreads();
  bool bad = false;
#pragma unroll
  for (int q = 0; q < kPer; q++) {
    const int i = tid * kPer + q;
    if (i >= nsub) continue;
    const HuffInterval& sg = seg[sb.seg[fbase + i]];
    sb.mcu0[fbase + i] = (int)sg.mcu0 + s_ex[i] - s_ex[sg.first_sub];
    // last slot of the segment: the segment must have produced exactly its MCUs
    const bool last = i + 1 == nsub || (sb.lim[fbase + i + 1] >> 31);
    if (last && s_ex[i + 1] - s_ex[sg.first_sub] < (int)sg.nmcu) bad = true;  // the data ended before the last MCU
  }
  if (bad) atomicOr(&status[frame], 1u);
}

__global__ __launch_bounds__(kSyncLaneThreads) void k_huff_write(const HuffScan* __restrict__ scans,
                                                                 const HuffInterval* __restrict__ ivs,
                                                                 const SyncLutImage* __restrict__ luts,
                                                                 const JpegFrameDesc* __restrict__ descs, SyncBuffers sb,
                                                                 int16_t* __restrict__ coef, size_t coef_stride,
                                                                 uint32_t* __restrict__ status) {
  // kWriteParts lanes per subsequence: lane (i, q) decodes the symbols that start in part q, from the subsequence's true
  // entry state (q = 0) or from the checkpoint the state-only walk of the true chain left at boundary q
  constexpr uint32_t kSubsPerBlock = kSyncLaneThreads / kWriteParts;
  __shared__ WriteTables T;
  __shared__ uint32_t s_stream[kStageWords];
  const int frame = blockIdx.y, tid = threadIdx.x;
  const SyncFrame fr = sb.frames[frame];
  if (blockIdx.x * kSubsPerBlock >= fr.nsub) return;  // uniform
  const HuffScan& sc = scans[frame];
  load_sync_tables(T, sc, luts, descs + frame, tid, kSyncLaneThreads);
  const uint32_t* words = reinterpret_cast<const uint32_t*>(sb.stream + (size_t)frame * sb.stream_stride);
  const uint32_t* staged = stage_stream(s_stream, words, fr, blockIdx.x * kSubsPerBlock,
                                        min(blockIdx.x * kSubsPerBlock + kSubsPerBlock - 1, fr.nsub - 1), tid, kSyncLaneThreads);
  __syncthreads();
  const uint32_t i = blockIdx.x * kSubsPerBlock + (uint32_t)tid / kWriteParts;
  const uint32_t q = (uint32_t)tid % kWriteParts;
  if (i >= fr.nsub) return;
  const size_t fbase = (size_t)frame * kSyncMaxSub;
  // (the independent loads first, all in flight together; then the two that need the true slot / the segment)
  const uint32_t tslot = sb.tslot[fbase + i];
  const uint2 e0 = sb.start[fbase + i];
  const int mcu0 = sb.mcu0[fbase + i];
  const uint32_t lim_raw = sb.lim[fbase + i];
  const uint32_t segi = sb.seg[fbase + i];
  const size_t cpo = ((fbase + i) * kHypSlots + tslot) * (kWriteParts - 1) + (q > 0 ? q - 1 : 0);
  const uint2 ecp = sb.cp[cpo];  // (q = 0 reads checkpoint 0 and drops it: an unconditional load)
  const int ncp = sb.cpn[cpo];
  const HuffInterval& sg = ivs[sc.seg_base + segi];
  const int mcu_end = (int)(sg.mcu0 + sg.nmcu);  // (a segment never writes into the next one's MCUs)
  const uint2 e = q == 0 ? e0 : ecp;
  const int mcu_first = mcu0 + (q == 0 ? 0 : ncp);
  SyncState st;
  st.p = e.x, st.cz = e.y;
  bool bad = false;
  const uint32_t part_bits = fr.sub_bits / kWriteParts;
  uint32_t limit = lim_raw & 0x7FFFFFFFu;
  if (q + 1 < kWriteParts) limit = min(limit, i * fr.sub_bits + (q + 1) * part_bits);
  int16_t* fdc = sb.dc + (size_t)frame * sb.dc_stride;
  if (staged)
    write_span(staged, st, limit, T, (int)sc.blocks_per_mcu, coef + (size_t)frame * coef_stride, fdc, mcu_first, mcu_end, &bad);
  else
    write_span(words, st, limit, T, (int)sc.blocks_per_mcu, coef + (size_t)frame * coef_stride, fdc, mcu_first, mcu_end, &bad);
  if (bad) atomicOr(&status[frame], 1u);
}

// DC differences -> DC values: per component a running sum over the blocks in scan order, restarted
// at every restart interval (jdhuff.c last_dc_val / process_restart).  One workgroup per frame,
// one MCU per thread and pass, on the compact side array of DC terms (one int16 per block, in the slab's block order:
// 14 KB per 640x480 frame instead of a 128-byte-strided walk over the 0.9 MB slab); k_idct takes the DC from there.
__global__ __launch_bounds__(kSyncThreads) void k_dc_prefix(const HuffScan* __restrict__ scans,
                                                            const JpegFrameDesc* __restrict__ descs,
                                                            int16_t* __restrict__ dc, size_t dc_stride) {
  __shared__ int s_wave[3][kSyncThreads / 64];
  __shared__ int s_pre[3][kSyncThreads];  // exclusive prefix of this pass (without the carry)
  const int frame = blockIdx.x, tid = threadIdx.x;
  if (scans[frame].nseg == 0) return;
  const JpegFrameDesc& d = descs[frame];
  int16_t* fdc = dc + (size_t)frame * dc_stride;
  const int total = d.mcux * d.mcuy;
  const int ri = d.restart_interval > 0 ? d.restart_interval : total;
  int carry[3] = {0, 0, 0};      // sum over all MCUs in front of this pass
  int seg_carry[3] = {0, 0, 0};  // ... in front of the restart interval that is open at the start of the pass
  // Sampling factors <= 2 (4:4:4, 4:2:2, 4:2:0, 4:4:0 -- all the GPU path's upsamplers cover): the up to 3 x 2 x 2 blocks of
  // the thread's MCU are requested TOGETHER, from clamped addresses with a predicate per block, and kept in registers for the
  // second phase.  (As loops over run-time h x v every block was a load, a wait and a branch of its own, twice: a dozen
  // dependent memory round trips per pass, 13.6 us for a kernel that moves 14 KB per frame.)
  int hh[3], vv[3], wb[3], co[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const bool have = c < d.ncomp;
    hh[c] = have ? d.h[c] : 0, vv[c] = have ? d.v[c] : 0, wb[c] = d.wblk[c], co[c] = (int)(d.coef_off[c] >> 6);
  }
  const bool small = max(max(hh[0], hh[1]), hh[2]) <= 2 && max(max(vv[0], vv[1]), vv[2]) <= 2;
  for (int base = 0; base < total; base += kSyncThreads) {
    const int mcu = base + tid;
    const bool valid = mcu < total;
    const int my = mcu / d.mcux, mx = mcu - my * d.mcux;
    int sums[3] = {0, 0, 0};
    int dcv[3][2][2];
    if (small) {
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int by = 0; by < 2; by++)
#pragma unroll
          for (int bx = 0; bx < 2; bx++) {
            const bool have = valid && by < vv[c] && bx < hh[c];
            const int idx = have ? co[c] + (my * vv[c] + by) * wb[c] + mx * hh[c] + bx : 0;
            const int v = fdc[idx];
            dcv[c][by][bx] = have ? v : 0;
          }
#pragma unroll
      for (int c = 0; c < 3; c++) sums[c] = dcv[c][0][0] + dcv[c][0][1] + dcv[c][1][0] + dcv[c][1][1];
    } else if (valid) {
      for (int c = 0; c < d.ncomp; c++)
        for (int by = 0; by < d.v[c]; by++)
          for (int bx = 0; bx < d.h[c]; bx++)
            sums[c] += fdc[(d.coef_off[c] >> 6) + (size_t)(my * d.v[c] + by) * d.wblk[c] + mx * d.h[c] + bx];
    }
    int pre[3], tot[3];
    block_exscan3(sums, s_wave, pre, tot);
#pragma unroll
    for (int c = 0; c < 3; c++) s_pre[c][tid] = pre[c];
    __syncthreads();
    // predictor = (sum in front of this MCU) - (sum in front of its restart interval)
    const int seg_start = (mcu / ri) * ri;
    int pred[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int before_seg = seg_start >= base ? carry[c] + s_pre[c][seg_start - base] : seg_carry[c];
      pred[c] = carry[c] + pre[c] - before_seg;
    }
    if (small) {
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int by = 0; by < 2; by++)
#pragma unroll
          for (int bx = 0; bx < 2; bx++) {
            pred[c] += dcv[c][by][bx];  // (0 for a block that does not exist)
            if (valid && by < vv[c] && bx < hh[c]) fdc[co[c] + (my * vv[c] + by) * wb[c] + mx * hh[c] + bx] = (int16_t)pred[c];
          }
    } else if (valid) {
      for (int c = 0; c < d.ncomp; c++)
        for (int by = 0; by < d.v[c]; by++)
          for (int bx = 0; bx < d.h[c]; bx++) {
            int16_t* p = fdc + (d.coef_off[c] >> 6) + (size_t)(my * d.v[c] + by) * d.wblk[c] + mx * d.h[c] + bx;
            pred[c] += *p;
            *p = (int16_t)pred[c];
          }
    }
    // the interval open at the start of the next pass
    const int nbase = base + kSyncThreads;
    const int nstart = (nbase / ri) * ri;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      if (nstart >= base && nstart < nbase) seg_carry[c] = carry[c] + s_pre[c][nstart - base];
      else if (nstart >= nbase) seg_carry[c] = carry[c] + tot[c];
      carry[c] += tot[c];
    }
    __syncthreads();  // s_pre is reused by the next pass
  }
}

}  // namespace

size_t sync_buffers_bytes(uint32_t max_frames, size_t stream_stride, size_t dc_stride, SyncBuffers* layout) {
  // carve one allocation: returns the size; with layout != nullptr fills offsets relative to layout->stream
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t subs = (size_t)max_frames * kSyncMaxSub;
  const size_t o_stream = take(stream_stride * max_frames);
  const size_t o_frames = take(sizeof(SyncFrame) * max_frames);
  const size_t o_ent = take(subs * kHypSlots * sizeof(uint2));
  const size_t o_ext = take(subs * kHypSlots * sizeof(uint2));
  const size_t o_nm = take(subs * kHypSlots * sizeof(int));
  const size_t o_cnt = take(subs * 2);
  const size_t o_start = take(subs * sizeof(uint2));
  const size_t o_mcu0 = take(subs * sizeof(int));
  const size_t o_map = take(subs * sizeof(unsigned long long));
  const size_t o_lim = take(subs * sizeof(uint32_t));
  const size_t o_seg = take(subs * sizeof(uint16_t));
  const size_t o_dc = take(dc_stride * sizeof(int16_t) * max_frames);
  const size_t o_cp = take(subs * kHypSlots * (kWriteParts - 1) * sizeof(uint2));
  const size_t o_cpn = take(subs * kHypSlots * (kWriteParts - 1) * sizeof(int));
  const size_t o_tslot = take(subs);
  if (layout) {
    uint8_t* base = layout->stream;
    layout->stream = base + o_stream;
    layout->stream_stride = stream_stride;
    layout->frames = reinterpret_cast<SyncFrame*>(base + o_frames);
    layout->ent = reinterpret_cast<uint2*>(base + o_ent);
    layout->ext = reinterpret_cast<uint2*>(base + o_ext);
    layout->nm = reinterpret_cast<int*>(base + o_nm);
    layout->cnt = base + o_cnt;
    layout->start = reinterpret_cast<uint2*>(base + o_start);
    layout->mcu0 = reinterpret_cast<int*>(base + o_mcu0);
    layout->map = reinterpret_cast<unsigned long long*>(base + o_map);
    layout->lim = reinterpret_cast<uint32_t*>(base + o_lim);
    layout->seg = reinterpret_cast<uint16_t*>(base + o_seg);
    layout->dc = reinterpret_cast<int16_t*>(base + o_dc);
    layout->dc_stride = dc_stride;
    layout->cp = reinterpret_cast<uint2*>(base + o_cp);
    layout->cpn = reinterpret_cast<int*>(base + o_cpn);
    layout->tslot = base + o_tslot;
    layout->max_frames = max_frames;
  }
  return off;
}

void launch_huffman_sync(const uint8_t* d_blob, const HuffScan* d_scans, const HuffInterval* d_ivs, uint32_t frames,
                         uint32_t max_nsub, uint32_t max_blocks_per_mcu, const SyncLutImage* d_luts,
                         const JpegFrameDesc* d_descs, int16_t* d_coef, size_t coef_stride, const SyncBuffers& sb,
                         uint32_t* d_status, hipStream_t s, const HuffStageHook* hook, size_t zero_int16) {
  if (!frames) return;
  // the slab (first zero_int16 coefficients of every frame), the DC side array and the decode status are cleared by the
  // first launch of the chain, beside the unstuffing
  ZeroArgs za{};
  za.coef = d_coef, za.coef_stride = coef_stride, za.vec_per_frame = (uint32_t)((zero_int16 + 7) / 8);
  za.dc = sb.dc, za.dc_stride = sb.dc_stride;
  za.dc_vec_per_frame = (uint32_t)std::min<size_t>((zero_int16 / 64 + 7) / 8, (sb.dc_stride + 7) / 8);
  za.status = d_status;
  const unsigned zero_rows = zero_int16 ? std::min<unsigned>((za.vec_per_frame + 4095) / 4096, 64u) : 0u;  // >= 4 stores per thread
  const uint32_t nsub = std::max(max_nsub, 1u);  // slots of the longest frame
  uint8_t* cnt_a = sb.cnt;
  uint8_t* cnt_b = sb.cnt + (size_t)sb.max_frames * kSyncMaxSub;
  const dim3 lanes(kSyncLaneThreads);
  // (hook: per-kernel profiling scopes of the caller; begin = true before, false after the launch)
  auto stage = [&](const char* name, auto&& launch) {
    if (hook && *hook) (*hook)(name, true);
    launch();
    if (hook && *hook) (*hook)(name, false);
  };
  stage("huff_unstuff", [&] {
    ufd_launch(k_huff_unstuff, dim3(frames, 1 + zero_rows), dim3(kSyncThreads), 0, s, d_blob, d_scans, d_ivs, sb, za);
  });
  stage("huff_seed", [&] {
    ufd_launch(k_huff_seed, dim3((nsub * max_blocks_per_mcu + kSyncLaneThreads - 1) / kSyncLaneThreads, frames), lanes, 0, s,
                       d_scans, d_luts, sb, cnt_a);
  });
  const dim3 gext((nsub * kHypSlots + kSyncLaneThreads - 1) / kSyncLaneThreads, frames);  // k_huff_link: 16 subsequences per block
  const dim3 gext2((nsub * kHypSlots + kExtendThreads - 1) / kExtendThreads, frames);     // k_huff_extend: 64
  const dim3 ext_threads(kExtendThreads);
  // speculation rounds (UFD_EXTEND_ROUNDS while measuring; 2 is what ships: docs/EXPERIMENTS.md rounds 4 and 5)
  static const int rounds = experiment_env("UFD_EXTEND_ROUNDS") ? std::max(0, std::min(4, std::atoi(experiment_env("UFD_EXTEND_ROUNDS")))) : 2;
  uint8_t *cnt_in = cnt_a, *cnt_out = cnt_b;
  for (int r = 0; r < rounds; r++) {
    stage("huff_extend", [&] { ufd_launch(k_huff_extend, gext2, ext_threads, 0, s, d_scans, d_luts, sb, (const uint8_t*)cnt_in, cnt_out); });
    std::swap(cnt_in, cnt_out);
  }
  stage("huff_link", [&] { ufd_launch(k_huff_link, gext, lanes, 0, s, sb, (const uint8_t*)cnt_in); });
  stage("huff_resolve", [&] {
    ufd_launch(k_huff_resolve, dim3(frames), dim3(kSyncThreads), 0, s, d_scans, d_ivs, d_luts, sb, cnt_in, d_status);
  });
  stage("huff_write", [&] {
    ufd_launch(k_huff_write, dim3((nsub * kWriteParts + kSyncLaneThreads - 1) / kSyncLaneThreads, frames), lanes, 0, s, d_scans, d_ivs,
                       d_luts, d_descs, sb, d_coef, coef_stride, d_status);
  });
  stage("dc_prefix", [&] { ufd_launch(k_dc_prefix, dim3(frames), dim3(kSyncThreads), 0, s, d_scans, d_descs, sb.dc, sb.dc_stride); });
}

}  // namespace ufd
