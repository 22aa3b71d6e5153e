// jpeg_host.hpp -- host half of the JPEG stage (row A1: turbojpeg::decompress_image,
// infer_server/src/inferer.rs:35): marker parsing and Huffman entropy decoding into quantised
// DCT coefficient blocks.  Everything after the entropy decoder (dequantisation, ISLOW IDCT,
// fancy chroma upsampling, YCbCr->RGB) runs on the GPU (jpeg_kernels.hip).
#pragma once
#include <cstddef>
#include <cstdint>

namespace ufd {

constexpr int kMaxComps = 3;

enum JpegStatus { kJpegOk = 0, kJpegCorrupt = -2, kJpegUnsupported = -3, kJpegNotEligible = -100 };

// Colour handling after upsampling (libjpeg default_decompress_parms for 1/3 components).
enum JpegColor : int32_t { kColorGray = 0, kColorYCbCr = 1, kColorRGB = 2 };

// Geometry of one frame; mirrored 1:1 into device memory for the kernels.
struct JpegFrameDesc {
  int32_t width, height, ncomp, color;
  int32_t hmax, vmax;
  int32_t mcux, mcuy;
  int32_t h[kMaxComps], v[kMaxComps];
  int32_t wblk[kMaxComps], hblk[kMaxComps];  // allocated blocks (MCU padded)
  int32_t dw[kMaxComps], dh[kMaxComps];      // libjpeg downsampled_width / downsampled_height
  uint32_t coef_off[kMaxComps];              // int16 offset of component c's blocks in the frame's coefficient slab
  uint32_t plane_off[kMaxComps];             // byte offset of component c's sample plane
  uint32_t coef_total;                       // int16 count for the frame
  uint32_t plane_total;                      // bytes
  uint32_t total_blocks;
  uint16_t qt[kMaxComps][64];                // natural order, per component
  int32_t progressive, restart_interval;
};

// Parses markers up to the first SOS.  Returns kJpegOk and fills `d` (without qt if DQT follows SOF: fixed later).
int jpeg_parse_header(const uint8_t* data, size_t len, JpegFrameDesc* d);

// Full entropy decode: fills d (geometry + quant tables) and coef[d->coef_total]
// (int16, natural order, [comp][block_row][block_col][64]).  `coef_cap` in int16 units.
int jpeg_decode_coefficients(const uint8_t* data, size_t len, JpegFrameDesc* d, int16_t* coef, size_t coef_cap);

// ---- device entropy decoding of restart-interval streams (huffman_kernels.hip) ----
// Lookup form of one Huffman table, identical on host and device.
struct HuffLut {
  uint16_t fast[1024];  // peek(10) -> (code length << 8) | symbol; 0 = code longer than 10 bits
  int32_t maxcode[18];  // exclusive upper bound of codes of length l, left-justified to 16 bits
  int32_t delta[17];    // symbol index = (code >> (16 - l)) + delta[l]
  uint8_t sym[256];
};

// Scan layout of one frame for the device decoder.
struct HuffScan {
  uint32_t blob_off;        // offset of the frame's JPEG bytes inside the batch blob
  uint32_t blocks_per_mcu;  // <= 10
  uint32_t lut_base;        // first of the 4 HuffLut of this frame: dc slot 0/1, ac slot 0/1
  // self-synchronising decoder: the frame's intervals ("segments": one without restart markers)
  uint32_t seg_base, nseg;  // range in the batch's interval array (nseg = 0: frame skipped)
  uint32_t sub_bytes;       // subsequence length for this frame (multiple of 4; >= 64, >= 32 in a batch of a few frames: model.cpp, sub_floor)
  uint32_t nsub;            // subsequence slots of the frame (sum over its segments)
  uint32_t pad;
  uint8_t blk_comp[12], blk_bx[12], blk_by[12], blk_dc[12], blk_ac[12];  // per block of the MCU
};

// One restart interval = one independent unit of entropy decoding.
struct HuffInterval {
  uint32_t frame;
  uint32_t begin, end;  // byte range inside the frame's JPEG (end = position of the terminating marker)
  uint32_t mcu0, nmcu;  // first MCU and MCU count
  uint32_t first_sub;   // self-synchronising decoder: first subsequence slot of the segment
};

struct GpuScanPlan {
  HuffScan scan;
  HuffLut luts[4];   // (only with build_luts)
  // What determines `luts`: the payload of every DHT segment in front of the scan, in stream order, then the scan's
  // component count and table selectors.  A camera stream repeats the same bytes in every frame, so the caller keeps
  // the table sets it has seen keyed by them and asks for the tables to be built only for a key it has not seen.
  static constexpr uint32_t kMaxKeyBytes = 1200;  // (the four Annex-K tables: 416 + 7 bytes)
  uint64_t key_hash;         // FNV-1a of key_bytes; 0 = no key (DHT payload longer than kMaxKeyBytes)
  uint32_t key_len;
  uint8_t key_bytes[kMaxKeyBytes];
  static constexpr int kMaxIntervals = 1024;
  uint32_t n_intervals;
  HuffInterval iv[kMaxIntervals];
};

// Header + marker scan only (no entropy decoding): fills the frame geometry and the interval
// list for the device decoder.  kJpegNotEligible when the stream is not a single interleaved
// baseline scan with restart intervals (the caller then decodes on the host).
// build_luts = false: the DHT segments are checked (structure, code space) and hashed into the plan's key but no lookup
// table is built -- about a quarter of the work; the result code is the same as with build_luts = true.
int jpeg_plan_gpu_scan(const uint8_t* data, size_t len, JpegFrameDesc* d, GpuScanPlan* plan, bool build_luts = true);

}  // namespace ufd
