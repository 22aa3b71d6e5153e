// kernels.hpp -- launchers of the gfx950 HIP kernels of the hot path (one per stage row of
// SURVEY.md section 8a).  Host code (model.cpp) only sees these plain C++ signatures.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <functional>

#include "experiments.hpp"
#include "jpeg_host.hpp"

namespace ufd {

// Every kernel launch of the library goes through ufd_launch: it notes the launch's shape (function, workgroups, threads,
// dynamic LDS) in a thread-local that the profiling scope around it reads -- ufd_profile_shapes then reports, per profiled
// label, how the launch sits on the GPU (registers and resident workgroups per CU from the runtime's own occupancy query).
struct LaunchShape {
  const void* fn = nullptr;
  uint32_t blocks = 0, threads = 0, lds = 0;
  uint32_t launches = 0;  // launches since the enclosing profiling scope opened (the shape kept is the first one's)
};
extern thread_local LaunchShape tl_launch_shape;
template <typename... KArgs, typename... Args>
inline void ufd_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args&&... args) {
  LaunchShape& sh = tl_launch_shape;
  if (sh.launches == 0) {  // the FIRST launch of a profiling scope is the one its label names (k_sort_nms, not the two behind it)
    sh.fn = reinterpret_cast<const void*>(kernel);
    sh.blocks = grid.x * grid.y * grid.z, sh.threads = block.x * block.y * block.z, sh.lds = (uint32_t)lds;
  }
  sh.launches++;
  hipLaunchKernelGGL(kernel, grid, block, lds, stream, static_cast<KArgs>(args)...);
}

// ---------------- A1: JPEG reconstruction (jpeg_kernels.hip) ----------------
// Dequantise + ISLOW IDCT of every 8x8 block of `count` frames into u8 sample planes.
// d_dc (device entropy decoder): DC term of block g of frame f at d_dc[f * dc_stride + g], overriding the slab's; or null.
void launch_idct(const JpegFrameDesc* d_descs, const int16_t* d_coef, size_t coef_stride, uint8_t* d_planes,
                 size_t plane_stride, uint32_t max_blocks, uint32_t count, bool zigzag, hipStream_t s,
                 const int16_t* d_dc = nullptr, size_t dc_stride = 0);
// Fancy upsampling + colour conversion -> interleaved RGB8 (pitch 3*width, frame stride rgb_stride).
void launch_upsample_rgb(const JpegFrameDesc* d_descs, const uint8_t* d_planes, size_t plane_stride, uint8_t* d_rgb,
                         size_t rgb_stride, uint32_t max_w, uint32_t max_h, uint32_t count, hipStream_t s);
// 4:2:0 YCbCr frames whose width is a multiple of 8 (every non-skipped frame of the batch): same pixels, 8 per thread.
void launch_upsample_rgb_420(const JpegFrameDesc* d_descs, const uint8_t* d_planes, size_t plane_stride, uint8_t* d_rgb,
                             size_t rgb_stride, uint32_t max_w, uint32_t max_h, uint32_t count, hipStream_t s);
// Same, fused with the A4 normalisation for frames that already have the model size:
// -> f32 [count][3][H][W].  norm_lut: [3][256].
void launch_upsample_norm(const JpegFrameDesc* d_descs, const uint8_t* d_planes, size_t plane_stride,
                          const float* d_norm_lut, float* d_out, uint32_t W, uint32_t H, uint32_t count, hipStream_t s);

// Entropy decoding on the device: self-synchronising subsequence decoding with speculation over
// the block index (huffman_kernels.hip).  scans[f].seg_base/nseg select the frame's segments in ivs
// (one per restart interval; a stream without restart markers is one segment); nseg = 0 skips the
// frame.  SyncBuffers is device scratch of one context, carved from one allocation.
// Per table set: the four lookup tables plus their state-only form (bits consumed by code word and
// magnitude | zigzag advance << 5; 0 = code word longer than the lookup).  Built on the host when a
// new table set is first seen (build_sync_lut_image), copied to LDS by every decoding block.
// Table set of a frame as the entropy kernels copy it to LDS: the long-code search data of the four
// tables, then their state-only step tables (seed / extend / resolve) and their symbol tables (write)
// -- a kernel takes the head and ONE of the two, 9.6 KB instead of 17.6 (LDS is what keeps the
// convolutions of the other contexts off a CU while these latency-bound kernels sit on it).
struct HuffSlow {
  int32_t maxcode[18];  // as HuffLut
  int32_t delta[17];
  uint8_t sym[256];
  int32_t pad;
};
struct SyncLutImage {
  HuffSlow slow[4];
  uint16_t step_dc[2][1024];  // state-only passes, DC tables: sync_step of the symbol behind this 10-bit window (0: code longer than 10 bits)
  // ... AC tables: low half = the same for the AC symbol; high half (round 6) = the entry of this symbol AND THE NEXT ONE
  // taken together -- bits consumed by both, zigzag advance of both -- when the first does not end the block and the
  // second's CODE lies inside the window behind the first's code and magnitude bits (a state-only walk never looks at
  // magnitude bits, so the second's may lie outside); 0: no such pair.  1.49 symbols per dependent look-up on the bench's
  // frames instead of 1 (tools/huff_pair_sim.py).
  uint32_t step_ac[2][1024];
  uint16_t fast[4][1024];
};
// State-only step of a symbol.  A DC symbol advances the zigzag index 0 -> 1, an AC coefficient by
// run + 1, ZRL by 16 and EOB to 64, so the decoder state update is "z += dz" for all of them.
__host__ __device__ inline uint16_t sync_step(int len, int sym, bool is_dc) {
  const int sz = sym & 15, run = sym >> 4;
  const int dz = is_dc ? 1 : (sz ? run + 1 : (run == 15 ? 16 : 64));
  return (uint16_t)((len + sz) | (dz << 5));
}
inline void build_sync_lut_image(const HuffLut* luts /*[4]: dc0 dc1 ac0 ac1*/, SyncLutImage* out) {
  for (int t = 0; t < 4; t++) {
    for (int i = 0; i < 18; i++) out->slow[t].maxcode[i] = luts[t].maxcode[i];
    for (int i = 0; i < 17; i++) out->slow[t].delta[i] = luts[t].delta[i];
    for (int i = 0; i < 256; i++) out->slow[t].sym[i] = luts[t].sym[i];
    out->slow[t].pad = 0;
    for (int i = 0; i < 1024; i++) {
      const int e = luts[t].fast[i];
      out->fast[t][i] = (uint16_t)e;
      const uint16_t one = e ? sync_step(e >> 8, e & 0xFF, t < 2) : (uint16_t)0;
      if (t < 2) {
        out->step_dc[t][i] = one;
        continue;
      }
      uint32_t two = 0;
      const int l1 = (one & 31), dz1 = one >> 5;  // code + magnitude bits, zigzag advance of the first symbol
      if (one && dz1 < 64 && l1 < 10) {           // (an EOB ends the block: the symbol behind it is a DC symbol of another table)
        const int e2 = luts[t].fast[(i << l1) & 1023];
        if (e2 && (e2 >> 8) <= 10 - l1) {         // the second CODE is fully inside the window's known bits
          const uint16_t s2 = sync_step(e2 >> 8, e2 & 0xFF, false);
          two = (uint32_t)(l1 + (s2 & 31)) | ((uint32_t)(dz1 + (s2 >> 5)) << 5);
        }
      }
      out->step_ac[t - 2][i] = (uint32_t)one | (two << 16);
    }
  }
}
constexpr int kSyncMaxSub = 4096;  // subsequence slots per frame (longer streams get longer subsequences)
constexpr int kSyncMaxSeg = 1024;  // segments (restart intervals) per frame
struct SyncFrame {
  uint32_t total_bits, nsub, sub_bits, pad;
};
struct SyncBuffers {
  uint8_t* stream = nullptr;  // unstuffed entropy-coded segment per frame
  size_t stream_stride = 0;
  SyncFrame* frames = nullptr;
  uint2* ent = nullptr;    // [frame][subsequence][slot] cached entry states (bit position, block | zigzag << 8)
  uint2* ext = nullptr;    // ... their exit states
  int* nm = nullptr;       // ... MCUs completed in between
  uint8_t* cnt = nullptr;  // [2][frame][subsequence] cached pairs (double-buffered across rounds)
  uint2* start = nullptr;  // [frame][subsequence] true entry state
  int* mcu0 = nullptr;     // [frame][subsequence] first MCU
  uint32_t* lim = nullptr; // [frame][subsequence] end of the slot's data in bits (min(slot end, segment end)) | first-of-segment << 31
  uint16_t* seg = nullptr; // [frame][subsequence] segment of the slot
  unsigned long long* map = nullptr;  // [frame][subsequence] slot -> slot of the next subsequence, 16 nibbles
  int16_t* dc = nullptr;   // [frame][block] DC term of every block (differences from k_huff_write, values after k_dc_prefix)
  size_t dc_stride = 0;
  uint2* cp = nullptr;     // [frame][subsequence][slot][3] state at the part boundaries inside the subsequence (checkpoints of the walk)
  int* cpn = nullptr;      // ... MCUs completed in front of each
  uint8_t* tslot = nullptr;  // [frame][subsequence] slot of the true chain (k_huff_resolve)
  uint32_t max_frames = 0;
};
// Called before (true) and after (false) each kernel of the pipeline with its short name (profiling).
using HuffStageHook = std::function<void(const char* kernel, bool begin)>;
// Size of the allocation for `max_frames` frames; with `layout` (layout->stream = base pointer) fills it in.
size_t sync_buffers_bytes(uint32_t max_frames, size_t stream_stride, size_t dc_stride /* blocks per frame */, SyncBuffers* layout);
void launch_huffman_sync(const uint8_t* d_blob, const HuffScan* d_scans, const HuffInterval* d_ivs, uint32_t frames,
                         uint32_t max_nsub, uint32_t max_blocks_per_mcu, const SyncLutImage* d_luts,
                         const JpegFrameDesc* d_descs, int16_t* d_coef, size_t coef_stride, const SyncBuffers& sb,
                         uint32_t* d_status, hipStream_t s, const HuffStageHook* hook = nullptr, size_t zero_int16 = 0);
// (zero_int16 > 0: the first launch also clears that many coefficients of every frame's slab, the DC side array and
// d_status -- what launch_zero_coef does as a launch of its own)

// 4:2:0 YCbCr specialisation (every frame of the batch: 3 components, 2x2 luma sampling, fancy
// upsampling applicable, W % 8 == 0): same results, 8 pixels per thread with wide loads.
void launch_upsample_norm_420(const JpegFrameDesc* d_descs, const uint8_t* d_planes, size_t plane_stride,
                              const float* d_norm_lut, float* d_out, uint32_t W, uint32_t H, uint32_t count,
                              hipStream_t s);

// ---------------- A2-A4: Triangle resize + normalise (preproc_kernels.hip) ----------------
struct ResizeTaps {          // device pointers, one table per axis
  const int32_t* left;       // [D]
  const int32_t* cnt;        // [D]
  const float* w;            // [D][stride]
  int32_t stride;
};
// src: [count] frames of sh x sw RGB8 (row pitch, frame stride src_stride) -> f32 [count][3][dh][dw].
void launch_resize_norm(const uint8_t* d_src, uint32_t sw, uint32_t sh, uint32_t pitch, size_t src_stride,
                        ResizeTaps vert, ResizeTaps horz, const float* d_norm_lut, float* d_out, uint32_t dw,
                        uint32_t dh, uint32_t count, hipStream_t s);
// same-size frames: normalise only.
void launch_norm_only(const uint8_t* d_src, uint32_t w, uint32_t h, uint32_t pitch, size_t src_stride,
                      const float* d_norm_lut, float* d_out, uint32_t count, hipStream_t s);

// ---------------- A6: convolutions (conv_kernels.hip) ----------------
struct ConvArgs {
  const float* in;    // [B][in_ctotal][ih][iw]
  const float* w;     // layer weights, kernel-specific packing
  const float* bias;  // [cout]
  float* out;         // [B][out_ctotal][oh][ow], written at channel offset out_coff
  const float* res;   // optional residual [B][cout][oh][ow]: out = relu(conv + res)
  const float* w2;    // fused dw->pw kernel: depthwise weights packed [cin][12] (taps, bias, pad)
  const float* in2;   // pointwise kernel: second input tensor for k-steps >= ksplit (two 1x1 convs summed as one), or null
  const float* bias2; // unused by the fused kernel (bias is inside w2)
  int32_t B, cin, cout, ih, iw, oh, ow;
  int32_t k, stride, pad, dil, depthwise, relu;
  int32_t in_ctotal, out_ctotal, out_coff;
  int32_t in2_ctotal, ksplit;  // with in2: its channel count, first k-step (2 channels each) read from it
  int32_t dbg;         // ablation switches (UFD_CONV_DBG), timing experiments only
  int32_t tiles, cts;  // MFMA kernels: pixel tiles (blocks) and 32-cout tiles, set by the launcher
  int32_t band;        // chained dw->pw kernel: output rows per wave (row rolling), set by the launcher
};
// Up to three convolutions that share a launch configuration (same shapes, different weights /
// outputs: cls+reg head pairs, the three RFB reduce convs) run as one launch, blockIdx.y selects.
struct ConvArgs3 {
  ConvArgs a[3];
};

// Reference-order direct convolution (any layer).  w: [cout][cin/g][k][k].
void launch_conv_direct(const ConvArgs& a, hipStream_t s);
// Pointwise 1x1 on fp32 MFMA.  w: packed by pack_pointwise_weights().
void launch_conv_pointwise_mfma(const ConvArgs* a, int n, hipStream_t s);
// floats needed for the packed pointwise weight image of a cin->cout layer
size_t pointwise_packed_floats(int cin, int cout);
void pack_pointwise_weights(const float* w /*[cout][cin]*/, int cin, int cout, float* packed);
// Depthwise 3x3 (+ReLU) fused into the following pointwise conv.  `a` describes the pair:
// in/ih/iw/cin = depthwise input, oh/ow/cout = pointwise output, w/bias = packed pointwise
// weights, w2/bias2 = depthwise weights; relu applies to the pointwise output.
bool dwpw_supported(const ConvArgs& a, int stride);
// Two consecutive dw->pw blocks (stride 1 then stride 2, 16 -> 32 -> <=32 channels) in one launch:
// `first` / `second` are the ConvArgs of the two pointwise layers as for launch_conv_dwpw_mfma.
// True when launch_conv_dwpw_mfma runs these layers on the cooperative kernel (k_dwpw_coop).
bool dwpw_uses_coop(const ConvArgs* args, int n);
bool dwpw2_supported(const ConvArgs& first, const ConvArgs& second);
void launch_conv_dwpw2_mfma(const ConvArgs& first, const ConvArgs& second, hipStream_t s);
size_t depthwise_packed_floats(int c);
void pack_depthwise_weights(const float* w /*[c][9]*/, const float* bias, int c, float* packed /*[c][12]*/);
void launch_conv_dwpw_mfma(const ConvArgs* a, int n, int stride, hipStream_t s);
// Two independent convs as ONE grid (k_dual_*): group A = n_a merged dw->pw convs (a cls/reg head pair), B = one dw->pw
// block of stride b_stride (cooperative kernel) or, b_stride == 0, one 1x1 conv.  Returns false when that combination of
// kernel instances is not compiled (the caller then issues the two launches one after the other).
bool launch_conv_dual(const ConvArgs* a, int n_a, int a_stride, const ConvArgs* b, int b_stride, hipStream_t s);
// Template arguments of the kernel instance those launchers pick, as rocprofv3 prints them ("<16, 1, true>"): profiling labels.
const char* conv_dual_instance(const ConvArgs* a, int n_a, int a_stride, const ConvArgs* b, int b_stride);
const char* conv_pointwise_instance(const ConvArgs* a, int n);
const char* conv3x3_rows_instance(const ConvArgs* a, int n);
const char* conv_dwpw_instance(const ConvArgs* a, int n, int stride);
const char* conv_dwpw2_instance(const ConvArgs& first, const ConvArgs& second);
// Dense 3x3 (cout <= 16) as implicit GEMM on fp32 MFMA.  w: packed by pack_conv3x3_weights().
void launch_conv3x3_mfma(const ConvArgs* a, int n, hipStream_t s);
// Row variant (16-byte row loads + cross-lane shuffles instead of per-tap gathers).
bool conv3x3_rows_supported(const ConvArgs& a);
// RFB tail as one launch (k_rfb_tail): the three dilated 3x3 convs + relu(ConvLinear(cat) + shortcut(x)); the concat tensor
// never exists.  dil3: the convs in concat order with the row packing of their weights; fin: the summed 1x1 with the tail
// packing of both weight matrices (pack_rfb_tail_weights) and both biases summed.
size_t rfb_tail_packed_floats();
void pack_rfb_tail_weights(const float* w_lin /*[64][48]*/, const float* w_short /*[64][64]*/, float* packed);
bool rfb_tail_supported(const ConvArgs* dil3, const ConvArgs& fin);
void launch_rfb_tail(const ConvArgs* dil3, const ConvArgs& fin, hipStream_t s);
// Stem conv straight from the decoder's 4:2:0 / 4:2:2 sample planes (every frame of the batch at the model
// size): a = the stem's ConvArgs with the row packing of its weights (pack_conv3x3_rows_weights);
// frames whose descriptor does not match (failed frames) read as zero input.
struct StemArgs {
  ConvArgs a;
  const JpegFrameDesc* descs;
  const uint8_t* planes;
  size_t plane_stride;
  const float* lut;  // 3 x 256 normalisation table
};
bool stem_planes_supported(const ConvArgs& a);
void launch_stem_planes_mfma(const StemArgs& sa, hipStream_t s);
void launch_conv3x3_rows_mfma(const ConvArgs* args, int n, hipStream_t s);
size_t conv3x3_rows_packed_floats(int cin);
void pack_conv3x3_rows_weights(const float* w /*[cout][cin][3][3]*/, int cin, int cout, float* packed);
size_t conv3x3_packed_floats(int cin);
void pack_conv3x3_weights(const float* w /*[cout][cin][3][3]*/, int cin, int cout, float* packed);

// ---------------- A6 tail + A7: softmax, prior decode, threshold (post_kernels.hip) ----------------
struct HeadArgs {
  const float* cls[4];  // [B][A*2][fh][fw]
  const float* reg[4];  // [B][A*4][fh][fw]
  int32_t plane[4];     // fh*fw
  int32_t anchors[4];
  int32_t base[5];      // first prior index of each head; base[4] = K
};
// scores [B][K][2], boxes [B][K][4]; candidates conf > min_conf appended to keys[B][key_stride] / counts[B].
void launch_head_decode(const HeadArgs& h, const float* d_priors, uint32_t B, float min_conf, float* d_scores,
                        float* d_boxes, unsigned long long* d_keys, size_t key_stride, uint32_t* d_counts,
                        hipStream_t s);
// threshold only, for caller-provided raw outputs (ufd_debug_postproc)
void launch_threshold(const float* d_scores, uint32_t K, uint32_t B, float min_conf, unsigned long long* d_keys,
                      size_t key_stride, uint32_t* d_counts, hipStream_t s);
// ---------------- A8-A10: sort + greedy NMS, one workgroup per frame ----------------
struct Det {
  float x_tl, y_tl, x_br, y_br, conf;
};
// Frames with more than 256 (and at most 2048) candidates are finished by two more launches: their
// suppression matrix over the whole GPU, then one wave per frame (d_mat: nms_matrix_bytes(B) of
// scratch; nullptr keeps everything inside the first kernel).
// host_out (a frame or a few at a time, round 6): the kernel also hands the frame's results to the slot's pinned host arrays
// -- decode status, detection count, the first max_rows detections -- which was a launch of its own (k_results_out); with
// d_mat == nullptr beside it the whole of A7-A10 is ONE launch instead of four.
struct NmsHostOut {
  const uint32_t* d_status = nullptr;  // device entropy decoder's per-frame flags, or null
  uint32_t* h_status = nullptr;
  uint32_t* h_ndet = nullptr;          // null: no host output from the kernel
  float* h_dets = nullptr;
  uint32_t max_rows = 0;
};
void launch_sort_nms(unsigned long long* d_keys, size_t key_stride, uint32_t* d_counts, const float* d_boxes,
                     uint32_t K, float max_iou, Det* d_dets, uint32_t det_stride, uint32_t* d_ndet, float4* d_sel_spill,
                     unsigned long long* d_mat, uint32_t B, hipStream_t s, const NmsHostOut& host_out = NmsHostOut());
size_t nms_matrix_bytes(uint32_t B, uint32_t K);

// ---------------- N1: rectangles + JPEG re-encode (encode_kernels.hip, encode_host.cpp) ----------------
// inferer.rs:38-40 on frames resident in HBM as tight RGB8 (pitch 3 * width, frame stride rgb_stride).  Frames whose
// descriptor has width 0 (failed decode) are skipped: their output length is 0.
#define UFD_ENC_MAX_FRAMES 1024
// jcdctmgr.c reciprocal quantisation of table t (0 luma, 1 chroma), natural order:
// q = ((|v| + corr) * recip) >> shift, sign restored.
struct EncQuant {
  uint16_t recip[2][64], corr[2][64];
  uint8_t shift[2][64];
};
// Device working set of the encoder for one batch (one per context, allocated on first use).
struct EncBuffers {
  uint8_t* planes = nullptr;     // [frame] Y [16 mcuy][16 mcux] | Cb [8 mcuy][8 mcux] | Cr: MCU-padded sample planes
  size_t plane_stride = 0;
  int16_t* coef = nullptr;       // [frame][mcu][Y00 Y01 Y10 Y11 Cb Cr][64] quantised, zigzag order
  size_t coef_stride = 0;        // int16 per frame
  uint32_t* bits = nullptr;      // [frame][block] code length, then (in place) bit offset
  size_t blk_stride = 0;
  uint32_t* total_bits = nullptr;  // [frame]
  uint32_t* words = nullptr;     // [frame] bit stream before byte stuffing, 32-bit words MSB first
  size_t word_stride = 0;
  uint32_t* chunk_ff = nullptr;  // [frame][4 KiB chunk] 0xFF bytes
  size_t chunk_stride = 0;
  const uint32_t* tables = nullptr;  // [2][272] (size << 16) | code: 16 DC + 256 AC entries, luma / chroma
  const uint8_t* header = nullptr;   // [framing prefix][SOI .. SOS][framing suffix]
  uint32_t pre_len = 0, hdr_len = 0, dim_off = 0, post_len = 0;  // dim_off: offset of SOF0's height field in the header
  uint32_t* out_len = nullptr;   // [frame] bytes of the finished stream (0: skipped)
  uint32_t* out_off = nullptr;   // [frame] offset in `out` (16-byte aligned)
  uint32_t* out_total = nullptr; // [1] bytes used in `out`
  uint8_t* out = nullptr;        // finished streams, packed
};
using EncStageHook = std::function<void(const char* stage, bool begin)>;
// Rectangles and (text) confidence labels of the detections, in detection order, into the RGB frames.  d_ops: scratch of
// label_ops_bytes(frames, det_stride); d_glyphs / d_coverage: the atlas of glyph_atlas.inc (label_atlas) on the device.
size_t label_ops_bytes(uint32_t frames, uint32_t det_stride);
void launch_draw_labels(const JpegFrameDesc* d_descs, const Det* d_dets, uint32_t det_stride, const uint32_t* d_ndet,
                        uint32_t max_dets, void* d_ops, const int* d_glyphs, const float* d_coverage, bool text, uint8_t* d_rgb,
                        size_t rgb_stride, uint32_t max_w, uint32_t max_h, float label_w, float label_h, uint32_t count,
                        hipStream_t s);
// the atlas as host arrays: glyph records {x, y, w, h, offset} for [position][character], coverage floats
void label_atlas(const int** glyphs, size_t* glyph_ints, const float** coverage, size_t* coverage_floats);
void launch_jpeg_encode(const JpegFrameDesc* d_descs, const uint8_t* d_rgb, size_t rgb_stride, uint32_t max_w, uint32_t max_h,
                        uint32_t count, const EncQuant& q, bool ifast, const EncBuffers& e, hipStream_t s,
                        const EncStageHook* hook = nullptr);
// host side (encode_host.cpp): tables of jpeg_set_quality(quality, TRUE), their reciprocal form, the Annex-K code
// tables in the kernels' layout and the marker segments SOI .. SOS (+ optional multipart framing, lib.rs:48-57).
void enc_quant_tables(int quality, uint8_t luma[64], uint8_t chroma[64]);
void enc_make_quant(const uint8_t luma[64], const uint8_t chroma[64], bool ifast, EncQuant* out);
void enc_make_code_tables(uint32_t out[2 * 272]);
size_t enc_make_header(const uint8_t luma[64], const uint8_t chroma[64], bool multipart, uint8_t* out /*>= 768*/, uint32_t* pre_len,
                       uint32_t* hdr_len, uint32_t* dim_off, uint32_t* post_len);
// worst-case sizes for a frame of mcus 16x16 MCUs: entropy-coded bytes before stuffing (26 bits per coefficient)
inline size_t enc_stream_bound(size_t mcus) { return mcus * 6 * 64 * 26 / 8 + 64; }

}  // namespace ufd
