// jpeg_host.hpp -- host half of the JPEG stage (row A1: turbojpeg::decompress_image,
// infer_server/src/inferer.rs:35): marker parsing and Huffman entropy decoding into quantised
// DCT coefficient blocks.  Everything after the entropy decoder (dequantisation, ISLOW IDCT,
// fancy chroma upsampling, YCbCr->RGB) runs on the GPU (jpeg_kernels.hip).
#pragma once
#include <cstddef>
#include <cstdint>

namespace ufd {

constexpr int kMaxComps = 3;

enum JpegStatus { kJpegOk = 0, kJpegCorrupt = -2, kJpegUnsupported = -3 };

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

}  // namespace ufd
