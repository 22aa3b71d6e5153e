// huffman_kernels.hip -- entropy-decoding half of row A1 on the device (the part of
// turbojpeg::decompress_image, infer_server/src/inferer.rs:35, that libjpeg-turbo does in
// jdhuff.c), for baseline streams with restart intervals: every restart interval is an
// independent Huffman stream with reset DC predictors, so one lane decodes one interval
// (camera MJPG with one interval per MCU row: 30 intervals per 640x480 frame, 960 per 32-frame
// batch).  Serial bit-twiddling per lane, irregular byte gathers: latency-bound integer work
// that runs beside the other context's convolution kernels.  The coefficient slab is zeroed by a
// memset node before the launch; lanes store only non-zero coefficients (natural order).
#include "kernels.hpp"

#include <algorithm>
#include <cstddef>

namespace ufd {
namespace {

__constant__ uint8_t c_zigzag[80] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33,
                                     40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
                                     29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                                     47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

// Bit reader over one interval.  Bytes come from 8-byte aligned chunks held in registers with two
// chunks prefetched ahead, so the lane issues one global load per 8 input bytes and never waits
// on a byte-by-byte dependent chain; 0xFF00 unstuffing happens in registers.
struct BitReader {
  const uint64_t* q;      // next aligned chunk to prefetch
  uint64_t cur, n1, n2;   // chunk being consumed + two prefetched
  int ci;                 // next byte inside cur (0..8)
  int remaining;          // bytes of the interval not yet consumed
  uint64_t acc;
  int n;                  // valid bits at the top of acc
  int pad;                // zero bits appended past the end of the interval

  __device__ __forceinline__ void init(const uint8_t* begin, const uint8_t* end) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(begin);
    q = reinterpret_cast<const uint64_t*>(a & ~(uintptr_t)7);
    ci = (int)(a & 7);
    remaining = (int)(end - begin);
    cur = q[0], n1 = q[1], n2 = q[2];  // the blob has >= 32 bytes of slack behind every frame
    q += 3;
    acc = 0, n = 0, pad = 0;
  }
  __device__ __forceinline__ unsigned next_byte() {
    if (ci == 8) {
      cur = n1, n1 = n2, n2 = *q++;
      ci = 0;
    }
    const unsigned b = (unsigned)(cur >> (8 * ci)) & 0xFFu;
    ci++;
    remaining--;
    return b;
  }
  __device__ __forceinline__ void fill() {
    while (n <= 56) {
      unsigned c = 0;
      if (remaining > 0) {
        c = next_byte();
        if (c == 0xFF) {
          if (remaining > 0 && next_byte() == 0) {
            // stuffed zero consumed
          } else {
            remaining = 0;  // fill byte or marker: the interval's data is over
            c = 0;
            pad += 8;
          }
        }
      } else {
        pad += 8;
      }
      acc |= (uint64_t)c << (56 - n);
      n += 8;
    }
  }
  __device__ __forceinline__ uint32_t peek(int k) const { return (uint32_t)(acc >> (64 - k)); }
  __device__ __forceinline__ void skip(int k) {
    acc <<= k;
    n -= k;
  }
  __device__ __forceinline__ int get(int k) {  // 1..16 bits
    if (n < k) fill();
    const int v = (int)peek(k);
    skip(k);
    return v;
  }
  __device__ __forceinline__ bool overrun() const { return pad > 0 && n < pad; }
};

__device__ __forceinline__ int decode_symbol(BitReader& br, const HuffLut* __restrict__ t) {
  const int e = t->fast[br.peek(10)];
  if (e) {
    br.skip(e >> 8);
    return e & 0xFF;
  }
  const int code = (int)br.peek(16);
  int l = 11;
  while (l <= 16 && code >= t->maxcode[l]) l++;
  if (l > 16) return -1;
  br.skip(l);
  return t->sym[((code >> (16 - l)) + t->delta[l]) & 0xFF];
}

__device__ __forceinline__ int extend(int v, int s) { return v < (1 << (s - 1)) ? v + (int)(0xFFFFFFFFu << s) + 1 : v; }

__global__ __launch_bounds__(64) void k_huffman_rst(const uint8_t* __restrict__ blob, const HuffScan* __restrict__ scans,
                                                    const HuffInterval* __restrict__ ivs, uint32_t n_iv,
                                                    const HuffLut* __restrict__ luts,
                                                    const JpegFrameDesc* __restrict__ descs, int16_t* __restrict__ coef,
                                                    size_t coef_stride, uint32_t* __restrict__ status) {
  // the wave's table set (frames of one camera stream share it) lives in LDS: every symbol is a
  // dependent table lookup, LDS latency instead of a global round trip
  // (the host only takes this path when every frame of the batch uses the same table set and
  // MCU layout, so both are wave-uniform: plain LDS addressing, no generic pointers)
  __shared__ HuffLut s_lut[4];
  __shared__ uint32_t s_blk[12];  // per block of the MCU: comp | bx << 8 | by << 12 | dc << 16 | ac << 20
  const uint32_t t0 = blockIdx.x * 64;
  {
    const HuffScan& s0 = scans[ivs[min(t0, n_iv - 1)].frame];
    const uint32_t* src = reinterpret_cast<const uint32_t*>(luts + s0.lut_base);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_lut);
    for (int i = threadIdx.x; i < (int)(sizeof(HuffLut) * 4 / 4); i += 64) dst[i] = src[i];
    if (threadIdx.x < 12)
      s_blk[threadIdx.x] = s0.blk_comp[threadIdx.x] | (s0.blk_bx[threadIdx.x] << 8) | (s0.blk_by[threadIdx.x] << 12) |
                           (s0.blk_dc[threadIdx.x] << 16) | (s0.blk_ac[threadIdx.x] << 20);
  }
  __syncthreads();
  const uint32_t t = t0 + threadIdx.x;
  if (t >= n_iv) return;
  const HuffInterval iv = ivs[t];
  const HuffScan& sc = scans[iv.frame];
  const JpegFrameDesc& d = descs[iv.frame];
  const uint8_t* base = blob + sc.blob_off;
  BitReader br;
  br.init(base + iv.begin, base + iv.end);
  int16_t* fcoef = coef + (size_t)iv.frame * coef_stride;
  const int bpm = (int)sc.blocks_per_mcu;
  const int mcux = d.mcux;
  // Flat state machine, ONE symbol per loop iteration for every lane: lanes of a wave sit in
  // different blocks / zigzag positions, and nested per-block loops would make the wave run
  // sum-over-blocks(max-over-lanes(symbols)) iterations instead of max-over-lanes(total symbols).
  int m = 0, j = 0, k = 0;  // MCU inside the interval, block inside the MCU, zigzag position (0 = DC next)
  int pred0 = 0, pred1 = 0, pred2 = 0;
  bool bad = false;
  int16_t* blk = nullptr;
  int comp = 0;
  uint32_t binfo = 0;
  bool need_block = true;
  while (m < (int)iv.nmcu) {
    if (need_block) {
      const int mcu = (int)iv.mcu0 + m;
      const int my = mcu / mcux, mx = mcu - my * mcux;
      binfo = s_blk[j];
      comp = binfo & 0xFF;
      const int row = my * d.v[comp] + ((binfo >> 12) & 15), col = mx * d.h[comp] + ((binfo >> 8) & 15);
      blk = fcoef + d.coef_off[comp] + ((size_t)row * d.wblk[comp] + col) * 64;
      need_block = false;
    }
    if (br.n < 32) br.fill();  // >= 32 bits: a 16-bit code plus its 16 magnitude bits
    const HuffLut* t = (k == 0) ? &s_lut[(binfo >> 16) & 1] : &s_lut[2 + ((binfo >> 20) & 1)];
    const int rs = decode_symbol(br, t);
    if (rs < 0) {
      bad = true;
      break;
    }
    if (k == 0) {
      if (rs > 15) {
        bad = true;
        break;
      }
      int pred = comp == 0 ? pred0 : (comp == 1 ? pred1 : pred2);
      if (rs) pred += extend((int)br.peek(rs), rs), br.skip(rs);
      if (comp == 0) pred0 = pred; else if (comp == 1) pred1 = pred; else pred2 = pred;
      if (pred) blk[0] = (int16_t)pred;
      k = 1;
    } else {
      const int r = rs >> 4, sz = rs & 15;
      if (sz == 0) {
        k = (r == 15) ? k + 16 : 64;  // ZRL or EOB
      } else {
        k += r;
        if (k > 63) {
          bad = true;
          break;
        }
        blk[c_zigzag[k]] = (int16_t)extend((int)br.peek(sz), sz);
        br.skip(sz);
        k++;
      }
    }
    if (k >= 64) {  // block finished
      k = 0;
      need_block = true;
      if (++j == bpm) {
        j = 0;
        m++;
        if (br.overrun()) {
          bad = true;
          break;
        }
      }
    }
  }
  if (br.overrun()) bad = true;
  if (bad) atomicOr(&status[iv.frame], 1u);
}


// ---------------------------------------------------------------------------------------------
// Self-synchronising parallel decoder for streams WITHOUT restart markers (the common camera
// case).  A Huffman bit stream has no random access, but a decoder started at a wrong bit offset
// or in a wrong state falls into step with the true symbol sequence after a while.  The stream of
// every frame is cut into subsequences of >= 64 bytes; the decoder state at a subsequence boundary
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
//   k_dc_prefix     DC differences -> DC values.
// Integer, latency-bound work spread over the whole chip in short launches.
constexpr int kSyncThreads = 1024;
constexpr int kSyncMinBytes = 64;  // shortest subsequence

struct SyncState {
  uint32_t p;   // bit position of the next symbol in the compacted stream
  uint32_t cz;  // block inside the MCU | zigzag position << 8
};

struct SyncTables {
  // (the first two members mirror SyncLutImage: copied from the prebuilt per-table-set image)
  HuffLut lut[4];
  uint16_t step[4][1024];  // state-only form of lut[].fast: bits consumed (code + magnitude) | zigzag advance << 5
  uint32_t blk[12];        // per block of the MCU: comp | bx << 8 | by << 12 | dc << 16 | ac << 20
  uint32_t dc_bits, ac_bits;  // bit c = table slot (0/1) of block c of the MCU
  // coefficient offset of block c of MCU (mx, my): blk_base[c] + mx * blk_dx[c] + my * blk_dy[c]
  uint32_t blk_base[12], blk_dx[12], blk_dy[12];
  int mcux;
};

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
__device__ __forceinline__ int slow_symbol(const HuffLut& t, uint32_t top) {
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
__device__ __forceinline__ int sync_span(const uint32_t* words, SyncState& st, uint32_t limit, const SyncTables& T, int bpm) {
  int nmcu = 0;
  uint32_t pos = st.p;
  int c = (int)(st.cz & 0xFF), z = (int)(st.cz >> 8);
  BitWindow bw;
  bw.init(words, pos);
  const uint32_t dc_bits = T.dc_bits, ac_bits = T.ac_bits;
  int cur_t = z == 0 ? (int)((dc_bits >> c) & 1) : 2 + (int)((ac_bits >> c) & 1);
  while (pos < limit) {
    bw.refill();
    const uint32_t top = bw.top();
    uint32_t e = T.step[cur_t][top >> 22];
    if (__builtin_expect(e == 0, 0)) {
      const int ls = slow_symbol(T.lut[cur_t], top);
      // no code word (only reachable out of step or on corrupt data): consume one bit as EOB
      e = ls ? (uint32_t)sync_step(ls >> 8, ls & 0xFF, cur_t < 2) : (uint32_t)sync_step(1, 0, cur_t < 2);
    }
    const int adv = (int)(e & 31);
    bw.skip(adv);
    pos += (uint32_t)adv;
    z += (int)(e >> 5);
    // block finished?  (branch-free: some lane of the wave is at a block end in almost every iteration)
    const bool be = z >= 64;
    int cn = c + 1;
    cn = cn == bpm ? 0 : cn;
    nmcu += (be && cn == 0) ? 1 : 0;
    c = be ? cn : c;
    z = be ? 0 : z;
    cur_t = be ? (int)((dc_bits >> c) & 1) : 2 + (int)((ac_bits >> c) & 1);
  }
  st.p = pos;
  st.cz = (uint32_t)c | ((uint32_t)z << 8);
  return nmcu;
}

// Same walk from the true entry state, storing the coefficients: AC in ZIGZAG order (the IDCT
// kernel undoes it), DC as differences (k_dc_prefix sums them).  Zero coefficients are not stored
// (the slab is pre-zeroed).
__device__ __forceinline__ void write_span(const uint32_t* words, SyncState st, uint32_t limit, const SyncTables& T, int bpm,
                                           int16_t* fcoef, int mcu, int total_mcus, bool* bad) {
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
    int e = T.lut[cur_t].fast[top >> 22];
    if (__builtin_expect(e == 0, 0)) {
      e = slow_symbol(T.lut[cur_t], top);
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
      if (val) blk[0] = (int16_t)val;
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
__device__ __forceinline__ int block_exscan(int v, int* s_wave /*16*/, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
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

constexpr int kSyncLaneThreads = 256;  // seed / extend / write: one lane per (subsequence, hypothesis)
constexpr int kHypSlots = 16;          // cached (entry -> exit) pairs per subsequence; nibble 15 = "not cached"
constexpr int kHypAppendMax = 14;      // the speculation rounds fill slots 0..13, k_huff_resolve may use 14

__device__ __forceinline__ uint32_t sync_sub_bytes(uint32_t raw_bytes) {
  const uint32_t b = (raw_bytes + kSyncMaxSub - 1) / kSyncMaxSub;
  return max((b + 3u) & ~3u, (uint32_t)kSyncMinBytes);
}

static_assert(sizeof(SyncLutImage) % 16 == 0 && offsetof(SyncTables, blk) == sizeof(SyncLutImage), "image layout");

// Fills the block's tables: the table-set image with wide loads that are all in flight at once
// (a load -> store loop pays one memory round trip per iteration), the per-frame layout from the
// scan and frame descriptors.
__device__ __forceinline__ void load_sync_tables(SyncTables& T, const HuffScan& sc, const SyncLutImage* __restrict__ images,
                                                 const JpegFrameDesc* d, int tid, int nthreads) {
  constexpr int kVec = (int)(sizeof(SyncLutImage) / 16);
  const uint4* src = reinterpret_cast<const uint4*>(images + sc.lut_base / 4);
  uint4* dst = reinterpret_cast<uint4*>(&T);
  if (nthreads == kSyncLaneThreads) {
    constexpr int kPer = (kVec + kSyncLaneThreads - 1) / kSyncLaneThreads;
    uint4 r[kPer];
#pragma unroll
    for (int q = 0; q < kPer; q++)
      if (tid + q * kSyncLaneThreads < kVec) r[q] = src[tid + q * kSyncLaneThreads];
#pragma unroll
    for (int q = 0; q < kPer; q++)
      if (tid + q * kSyncLaneThreads < kVec) dst[tid + q * kSyncLaneThreads] = r[q];
  } else {
    for (int i = tid; i < kVec; i += nthreads) dst[i] = src[i];
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

__global__ __launch_bounds__(kSyncThreads) void k_huff_unstuff(const uint8_t* __restrict__ blob,
                                                               const HuffScan* __restrict__ scans,
                                                               const HuffInterval* __restrict__ ivs, SyncBuffers sb) {
  __shared__ int s_wave[kSyncThreads / 64];
  const int frame = blockIdx.x, tid = threadIdx.x;
  const HuffInterval iv = ivs[frame];
  SyncFrame* fr = sb.frames + frame;
  if (iv.nmcu == 0) {  // frame failed on the host side (uniform per block)
    if (tid == 0) fr->total_bits = 0, fr->nsub = 0, fr->sub_bits = kSyncMinBytes * 8;
    return;
  }
  const uint8_t* src = blob + scans[frame].blob_off;  // 16-byte aligned
  uint8_t* dst = sb.stream + (size_t)frame * sb.stream_stride;
  uint32_t out_base = 0;
  for (uint32_t tile = iv.begin & ~15u; tile < iv.end; tile += kSyncThreads * 16) {
    const uint32_t off = tile + (uint32_t)tid * 16;
    uint4 v = make_uint4(0, 0, 0, 0);
    unsigned prev = 0;
    if (off < iv.end) {
      v = *reinterpret_cast<const uint4*>(src + off);
      if (off > iv.begin) prev = src[off - 1];
    }
    const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
    uint32_t keep = 0;
    unsigned pb = prev;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const unsigned b = (w4[j >> 2] >> (8 * (j & 3))) & 0xFF;
      const uint32_t pos = off + j;
      if (pos == iv.begin) pb = 0;
      if (pos >= iv.begin && pos < iv.end && !(b == 0 && pb == 0xFF)) keep |= 1u << j;
      pb = b;
    }
    const int cnt = __popc(keep);
    int total;
    const int my_off = block_exscan(cnt, s_wave, &total);
    uint8_t* o = dst + out_base + my_off;
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (keep & (1u << j)) *o++ = (uint8_t)((w4[j >> 2] >> (8 * (j & 3))) & 0xFF);
    out_base += (uint32_t)total;
  }
  if (tid < 48) dst[out_base + tid] = 0;  // the bit window reads up to 3 words past the last symbol
  if (tid == 0) {
    const uint32_t sub_bytes = sync_sub_bytes(iv.end - iv.begin);
    fr->total_bits = out_base * 8;
    fr->sub_bits = sub_bytes * 8;
    fr->nsub = (out_base + sub_bytes - 1) / sub_bytes;
  }
}

// The lanes of a block decode neighbouring subsequences: their part of the stream is staged in LDS
// with coalesced loads first.  (Per-lane global loads of one word at a time made every refill of the
// bit window a ~1-2 us round trip: 16 of them in a row dominated the pass.)
constexpr uint32_t kStageWords = 8192 + 16;

// Stages the words of subsequences [i_lo, i_hi] plus the look-ahead tail; returns the pointer p with
// p[w] = word w of the stream (nullptr: the range does not fit, read global memory instead).
__device__ __forceinline__ const uint32_t* stage_stream(uint32_t* s_stream, const uint32_t* words, const SyncFrame& fr,
                                                        uint32_t i_lo, uint32_t i_hi, int tid, int nthreads) {
  const uint32_t sub_words = fr.sub_bits >> 5;
  const uint32_t w0 = i_lo * sub_words;
  const uint32_t w1 = min((i_hi + 1) * sub_words + 8, (fr.total_bits / 8 + 48) / 4);
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
  if (i >= fr.nsub || (i == 0 && g != 0)) return;
  SyncState st;
  st.p = i * fr.sub_bits, st.cz = (uint32_t)g;
  const SyncState entry = st;
  const uint32_t limit = min((i + 1) * fr.sub_bits, fr.total_bits);
  const int nm = staged ? sync_span(staged, st, limit, T, bpm) : sync_span(words, st, limit, T, bpm);
  const size_t slot = ((size_t)frame * kSyncMaxSub + i) * kHypSlots + g;
  sb.ent[slot] = make_uint2(entry.p, entry.cz);
  sb.ext[slot] = make_uint2(st.p, st.cz);
  sb.nm[slot] = nm;
  if (g == 0) cnt_out[(size_t)frame * kSyncMaxSub + i] = (uint8_t)(i == 0 ? 1 : min(bpm, kHypAppendMax));
}

// 16 lanes per subsequence i >= 1: lane k owns the exit state cached in slot k of subsequence i-1.
__global__ __launch_bounds__(kSyncLaneThreads) void k_huff_extend(const HuffScan* __restrict__ scans,
                                                                  const SyncLutImage* __restrict__ luts, SyncBuffers sb,
                                                                  const uint8_t* __restrict__ cnt_in,
                                                                  uint8_t* __restrict__ cnt_out) {
  __shared__ SyncTables T;
  __shared__ uint32_t s_stream[kStageWords];
  const int frame = blockIdx.y, tid = threadIdx.x;
  const SyncFrame fr = sb.frames[frame];
  const uint32_t lane = blockIdx.x * kSyncLaneThreads + tid;
  if ((blockIdx.x * kSyncLaneThreads) / kHypSlots >= fr.nsub) return;  // uniform
  const HuffScan& sc = scans[frame];
  load_sync_tables(T, sc, luts, nullptr, tid, kSyncLaneThreads);
  const uint32_t* words = reinterpret_cast<const uint32_t*>(sb.stream + (size_t)frame * sb.stream_stride);
  const uint32_t* staged = stage_stream(s_stream, words, fr, (blockIdx.x * kSyncLaneThreads) / kHypSlots,
                                        min((blockIdx.x * kSyncLaneThreads + kSyncLaneThreads - 1) / kHypSlots, fr.nsub - 1),
                                        tid, kSyncLaneThreads);
  __syncthreads();
  const uint32_t i = lane / kHypSlots;
  const int k = (int)(lane % kHypSlots);
  const bool in_range = i < fr.nsub;
  const size_t fbase = (size_t)frame * kSyncMaxSub;
  const int n_cur = in_range ? cnt_in[fbase + i] : 0;
  const int n_prev = (in_range && i > 0) ? cnt_in[fbase + i - 1] : 0;
  // lane k holds candidate k (exit state cached in slot k of subsequence i-1) and entry k of
  // subsequence i; the 16 lanes of the group compare through shuffles
  uint2 cand = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu), mine = make_uint2(0xFFFFFFFEu, 0xFFFFFFFEu);
  if (k < n_prev) cand = sb.ext[(fbase + i - 1) * kHypSlots + k];
  if (k < n_cur) mine = sb.ent[(fbase + i) * kHypSlots + k];
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
  if (!fresh || slot >= kHypAppendMax) return;
  SyncState st;
  st.p = cand.x, st.cz = cand.y;
  const uint32_t limit = min((i + 1) * fr.sub_bits, fr.total_bits);
  const int nm = staged ? sync_span(staged, st, limit, T, (int)sc.blocks_per_mcu)
                        : sync_span(words, st, limit, T, (int)sc.blocks_per_mcu);
  const size_t o = (fbase + i) * kHypSlots + slot;
  sb.ent[o] = cand;
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
  if (k == 0 && i < fr.nsub) sb.map[fbase + i] = has_next ? ((unsigned long long)hi << 32) | lo : 0xFEDCBA9876543210ull;
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
  const HuffInterval iv = ivs[frame];
  if (iv.nmcu == 0) return;
  const int nsub = (int)fr.nsub;
  if (nsub == 0) {
    if (tid == 0) atomicOr(&status[frame], 1u);
    return;
  }
  const HuffScan& sc = scans[frame];
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
      const int nm = sync_span(words, st, min((uint32_t)(m + 1) * fr.sub_bits, fr.total_bits), T, (int)sc.blocks_per_mcu);
      const int n = cnt[fbase + m];
      const int dst = n < kHypSlots - 1 ? n : kHypSlots - 2;  // append, or recycle the last usable slot
      const size_t o = (fbase + m) * kHypSlots + dst;
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
      my_nm[q] = sb.nm[o];
    }
    sum += my_nm[q];
  }
  int total;
  int first = block_exscan(sum, s_wave, &total);
#pragma unroll
  for (int q = 0; q < kPer; q++) {
    const int i = tid * kPer + q;
    if (i < nsub) sb.mcu0[fbase + i] = first;
    first += my_nm[q];
  }
  if (tid == 0 && total < (int)iv.nmcu) atomicOr(&status[frame], 1u);  // the data ended before the last MCU
}

__global__ __launch_bounds__(kSyncLaneThreads) void k_huff_write(const HuffScan* __restrict__ scans,
                                                                 const HuffInterval* __restrict__ ivs,
                                                                 const SyncLutImage* __restrict__ luts,
                                                                 const JpegFrameDesc* __restrict__ descs, SyncBuffers sb,
                                                                 int16_t* __restrict__ coef, size_t coef_stride,
                                                                 uint32_t* __restrict__ status) {
  __shared__ SyncTables T;
  __shared__ uint32_t s_stream[kStageWords];
  const int frame = blockIdx.y, tid = threadIdx.x;
  const SyncFrame fr = sb.frames[frame];
  if (blockIdx.x * kSyncLaneThreads >= fr.nsub) return;  // uniform
  const HuffScan& sc = scans[frame];
  load_sync_tables(T, sc, luts, descs + frame, tid, kSyncLaneThreads);
  const uint32_t* words = reinterpret_cast<const uint32_t*>(sb.stream + (size_t)frame * sb.stream_stride);
  const uint32_t* staged = stage_stream(s_stream, words, fr, blockIdx.x * kSyncLaneThreads,
                                        min(blockIdx.x * kSyncLaneThreads + kSyncLaneThreads - 1, fr.nsub - 1), tid,
                                        kSyncLaneThreads);
  __syncthreads();
  const uint32_t i = blockIdx.x * kSyncLaneThreads + tid;
  if (i >= fr.nsub) return;
  const size_t fbase = (size_t)frame * kSyncMaxSub;
  const uint2 e = sb.start[fbase + i];
  SyncState st;
  st.p = e.x, st.cz = e.y;
  bool bad = false;
  const uint32_t limit = min((i + 1) * fr.sub_bits, fr.total_bits);
  if (staged)
    write_span(staged, st, limit, T, (int)sc.blocks_per_mcu, coef + (size_t)frame * coef_stride, sb.mcu0[fbase + i],
               (int)ivs[frame].nmcu, &bad);
  else
    write_span(words, st, limit, T, (int)sc.blocks_per_mcu, coef + (size_t)frame * coef_stride, sb.mcu0[fbase + i],
               (int)ivs[frame].nmcu, &bad);
  if (bad) atomicOr(&status[frame], 1u);
}

// DC differences -> DC values: per component a running sum over the blocks in scan order
// (jdhuff.c last_dc_val).  One workgroup per frame, one MCU per thread and pass.
__global__ __launch_bounds__(kSyncThreads) void k_dc_prefix(const HuffInterval* __restrict__ ivs,
                                                            const JpegFrameDesc* __restrict__ descs,
                                                            int16_t* __restrict__ coef, size_t coef_stride) {
  __shared__ int s_wave[kSyncThreads / 64];
  const int frame = blockIdx.x, tid = threadIdx.x;
  if (ivs[frame].nmcu == 0) return;
  const JpegFrameDesc& d = descs[frame];
  int16_t* fcoef = coef + (size_t)frame * coef_stride;
  const int total = d.mcux * d.mcuy;
  int carry[3] = {0, 0, 0};
  for (int base = 0; base < total; base += kSyncThreads) {
    const int mcu = base + tid;
    const bool valid = mcu < total;
    const int my = mcu / d.mcux, mx = mcu - my * d.mcux;
    int sums[3] = {0, 0, 0};
    if (valid)
      for (int c = 0; c < d.ncomp; c++)
        for (int by = 0; by < d.v[c]; by++)
          for (int bx = 0; bx < d.h[c]; bx++)
            sums[c] += fcoef[d.coef_off[c] + ((size_t)(my * d.v[c] + by) * d.wblk[c] + mx * d.h[c] + bx) * 64];
    int pred[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      int tot;
      pred[c] = carry[c] + block_exscan(sums[c], s_wave, &tot);
      carry[c] += tot;
    }
    if (valid)
      for (int c = 0; c < d.ncomp; c++)
        for (int by = 0; by < d.v[c]; by++)
          for (int bx = 0; bx < d.h[c]; bx++) {
            int16_t* p = fcoef + d.coef_off[c] + ((size_t)(my * d.v[c] + by) * d.wblk[c] + mx * d.h[c] + bx) * 64;
            pred[c] += *p;
            *p = (int16_t)pred[c];
          }
  }
}

}  // namespace

void launch_huffman_rst(const uint8_t* d_blob, const HuffScan* d_scans, const HuffInterval* d_ivs, uint32_t n_iv,
                        const HuffLut* d_luts, const JpegFrameDesc* d_descs, int16_t* d_coef, size_t coef_stride,
                        uint32_t* d_status, hipStream_t s) {
  if (!n_iv) return;
  hipLaunchKernelGGL(k_huffman_rst, dim3((n_iv + 63) / 64), dim3(64), 0, s, d_blob, d_scans, d_ivs, n_iv, d_luts, d_descs,
                     d_coef, coef_stride, d_status);
}

namespace {

__global__ __launch_bounds__(256) void k_zero_coef(int16_t* __restrict__ coef, size_t coef_stride, uint32_t vec_per_frame) {
  // 16-byte stores; coef_stride is a multiple of 8 int16
  uint4* dst = reinterpret_cast<uint4*>(coef + (size_t)blockIdx.y * coef_stride);
  for (uint32_t v = blockIdx.x * 256 + threadIdx.x; v < vec_per_frame; v += gridDim.x * 256) dst[v] = make_uint4(0, 0, 0, 0);
}

}  // namespace

void launch_zero_coef(int16_t* d_coef, size_t coef_stride, size_t used_int16, uint32_t frames, hipStream_t s) {
  if (!frames || !used_int16) return;
  const uint32_t vecs = (uint32_t)((used_int16 + 7) / 8);
  const unsigned gx = std::min<unsigned>((vecs + 1023) / 1024, 256u);  // >= 4 stores per thread
  hipLaunchKernelGGL(k_zero_coef, dim3(std::max(gx, 1u), frames), dim3(256), 0, s, d_coef, coef_stride, vecs);
}

size_t sync_buffers_bytes(uint32_t max_frames, size_t stream_stride, SyncBuffers* layout) {
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
    layout->max_frames = max_frames;
  }
  return off;
}

void launch_huffman_sync(const uint8_t* d_blob, const HuffScan* d_scans, const HuffInterval* d_ivs, uint32_t frames,
                         uint32_t max_raw_bytes, uint32_t max_blocks_per_mcu, const SyncLutImage* d_luts,
                         const JpegFrameDesc* d_descs, int16_t* d_coef, size_t coef_stride, const SyncBuffers& sb,
                         uint32_t* d_status, hipStream_t s, const HuffStageHook* hook) {
  if (!frames) return;
  // upper bound of the subsequence count of any frame (the kernels use the unstuffed length)
  uint32_t sub_bytes = (max_raw_bytes + kSyncMaxSub - 1) / kSyncMaxSub;
  sub_bytes = std::max((sub_bytes + 3u) & ~3u, (uint32_t)kSyncMinBytes);
  const uint32_t nsub = sub_bytes == (uint32_t)kSyncMinBytes ? std::max(1u, (max_raw_bytes + sub_bytes - 1) / sub_bytes)
                                                             : (uint32_t)kSyncMaxSub;
  uint8_t* cnt_a = sb.cnt;
  uint8_t* cnt_b = sb.cnt + (size_t)sb.max_frames * kSyncMaxSub;
  const dim3 lanes(kSyncLaneThreads);
  // (hook: per-kernel profiling scopes of the caller; begin = true before, false after the launch)
  auto stage = [&](const char* name, auto&& launch) {
    if (hook && *hook) (*hook)(name, true);
    launch();
    if (hook && *hook) (*hook)(name, false);
  };
  stage("huff_unstuff", [&] { hipLaunchKernelGGL(k_huff_unstuff, dim3(frames), dim3(kSyncThreads), 0, s, d_blob, d_scans, d_ivs, sb); });
  stage("huff_seed", [&] {
    hipLaunchKernelGGL(k_huff_seed, dim3((nsub * max_blocks_per_mcu + kSyncLaneThreads - 1) / kSyncLaneThreads, frames), lanes, 0, s,
                       d_scans, d_luts, sb, cnt_a);
  });
  const dim3 gext((nsub * kHypSlots + kSyncLaneThreads - 1) / kSyncLaneThreads, frames);
  stage("huff_extend", [&] { hipLaunchKernelGGL(k_huff_extend, gext, lanes, 0, s, d_scans, d_luts, sb, (const uint8_t*)cnt_a, cnt_b); });
  stage("huff_extend", [&] { hipLaunchKernelGGL(k_huff_extend, gext, lanes, 0, s, d_scans, d_luts, sb, (const uint8_t*)cnt_b, cnt_a); });
  stage("huff_link", [&] { hipLaunchKernelGGL(k_huff_link, gext, lanes, 0, s, sb, (const uint8_t*)cnt_a); });
  stage("huff_resolve", [&] {
    hipLaunchKernelGGL(k_huff_resolve, dim3(frames), dim3(kSyncThreads), 0, s, d_scans, d_ivs, d_luts, sb, cnt_a, d_status);
  });
  stage("huff_write", [&] {
    hipLaunchKernelGGL(k_huff_write, dim3((nsub + kSyncLaneThreads - 1) / kSyncLaneThreads, frames), lanes, 0, s, d_scans, d_ivs,
                       d_luts, d_descs, sb, d_coef, coef_stride, d_status);
  });
  stage("dc_prefix", [&] { hipLaunchKernelGGL(k_dc_prefix, dim3(frames), dim3(kSyncThreads), 0, s, d_ivs, d_descs, d_coef, coef_stride); });
}

}  // namespace ufd
