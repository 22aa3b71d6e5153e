// huffman_kernels.hip -- entropy-decoding half of row A1 on the device (the part of
// turbojpeg::decompress_image, infer_server/src/inferer.rs:35, that libjpeg-turbo does in
// jdhuff.c), for baseline streams with restart intervals: every restart interval is an
// independent Huffman stream with reset DC predictors, so one lane decodes one interval
// (camera MJPG with one interval per MCU row: 30 intervals per 640x480 frame, 960 per 32-frame
// batch).  Serial bit-twiddling per lane, irregular byte gathers: latency-bound integer work
// that runs beside the other context's convolution kernels.  The coefficient slab is zeroed by a
// memset node before the launch; lanes store only non-zero coefficients (natural order).
#include "kernels.hpp"

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

}  // namespace

void launch_huffman_rst(const uint8_t* d_blob, const HuffScan* d_scans, const HuffInterval* d_ivs, uint32_t n_iv,
                        const HuffLut* d_luts, const JpegFrameDesc* d_descs, int16_t* d_coef, size_t coef_stride,
                        uint32_t* d_status, hipStream_t s) {
  if (!n_iv) return;
  hipLaunchKernelGGL(k_huffman_rst, dim3((n_iv + 63) / 64), dim3(64), 0, s, d_blob, d_scans, d_ivs, n_iv, d_luts, d_descs,
                     d_coef, coef_stride, d_status);
}

}  // namespace ufd
