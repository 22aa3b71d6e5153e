// topology.hpp -- UltraFace-RFB ("Mb_Tiny_RFB_fd") layer table: the graph inside the ONNX file
// the reference loads (infer_server/src/nn.rs:143-175) and runs (nn.rs:181).  SURVEY.md 8.1.
#pragma once
#include <cstddef>
#include <cstdint>

namespace ufd {

struct ConvSpec {
  const char* name;
  int cin, cout, k, stride, pad, dil, groups, relu;
  int src;  // conv whose (block) output feeds this one; -1 = network input; -2 = RFB concat
};

constexpr int kNumConv = 52;
constexpr int kRfbCatA = 15, kRfbCatB = 18, kRfbCatC = 22;  // written side by side into the 48-channel concat
constexpr int kRfbLinear = 23, kRfbShortcut = 24;           // out = relu(ConvLinear(cat) + shortcut(x))

inline const ConvSpec* conv_specs() {
  static const ConvSpec t[kNumConv] = {
      {"m0.conv_bn", 3, 16, 3, 2, 1, 1, 1, 1, -1},
      {"m1.dw", 16, 16, 3, 1, 1, 1, 16, 1, 0},
      {"m1.pw", 16, 32, 1, 1, 0, 1, 1, 1, 1},
      {"m2.dw", 32, 32, 3, 2, 1, 1, 32, 1, 2},
      {"m2.pw", 32, 32, 1, 1, 0, 1, 1, 1, 3},
      {"m3.dw", 32, 32, 3, 1, 1, 1, 32, 1, 4},
      {"m3.pw", 32, 32, 1, 1, 0, 1, 1, 1, 5},
      {"m4.dw", 32, 32, 3, 2, 1, 1, 32, 1, 6},
      {"m4.pw", 32, 64, 1, 1, 0, 1, 1, 1, 7},
      {"m5.dw", 64, 64, 3, 1, 1, 1, 64, 1, 8},
      {"m5.pw", 64, 64, 1, 1, 0, 1, 1, 1, 9},
      {"m6.dw", 64, 64, 3, 1, 1, 1, 64, 1, 10},
      {"m6.pw", 64, 64, 1, 1, 0, 1, 1, 1, 11},
      {"rfb.b0.0", 64, 8, 1, 1, 0, 1, 1, 0, 12},
      {"rfb.b0.1", 8, 16, 3, 1, 1, 1, 1, 1, 13},
      {"rfb.b0.2", 16, 16, 3, 1, 2, 2, 1, 0, 14},
      {"rfb.b1.0", 64, 8, 1, 1, 0, 1, 1, 0, 12},
      {"rfb.b1.1", 8, 16, 3, 1, 1, 1, 1, 1, 16},
      {"rfb.b1.2", 16, 16, 3, 1, 3, 3, 1, 0, 17},
      {"rfb.b2.0", 64, 8, 1, 1, 0, 1, 1, 0, 12},
      {"rfb.b2.1", 8, 12, 3, 1, 1, 1, 1, 1, 19},
      {"rfb.b2.2", 12, 16, 3, 1, 1, 1, 1, 1, 20},
      {"rfb.b2.3", 16, 16, 3, 1, 5, 5, 1, 0, 21},
      {"rfb.linear", 48, 64, 1, 1, 0, 1, 1, 0, -2},
      {"rfb.shortcut", 64, 64, 1, 1, 0, 1, 1, 0, 12},
      {"cls0.dw", 64, 64, 3, 1, 1, 1, 64, 1, 24},
      {"cls0.pw", 64, 6, 1, 1, 0, 1, 1, 0, 25},
      {"reg0.dw", 64, 64, 3, 1, 1, 1, 64, 1, 24},
      {"reg0.pw", 64, 12, 1, 1, 0, 1, 1, 0, 27},
      {"m8.dw", 64, 64, 3, 2, 1, 1, 64, 1, 24},
      {"m8.pw", 64, 128, 1, 1, 0, 1, 1, 1, 29},
      {"m9.dw", 128, 128, 3, 1, 1, 1, 128, 1, 30},
      {"m9.pw", 128, 128, 1, 1, 0, 1, 1, 1, 31},
      {"m10.dw", 128, 128, 3, 1, 1, 1, 128, 1, 32},
      {"m10.pw", 128, 128, 1, 1, 0, 1, 1, 1, 33},
      {"cls1.dw", 128, 128, 3, 1, 1, 1, 128, 1, 34},
      {"cls1.pw", 128, 4, 1, 1, 0, 1, 1, 0, 35},
      {"reg1.dw", 128, 128, 3, 1, 1, 1, 128, 1, 34},
      {"reg1.pw", 128, 8, 1, 1, 0, 1, 1, 0, 37},
      {"m11.dw", 128, 128, 3, 2, 1, 1, 128, 1, 34},
      {"m11.pw", 128, 256, 1, 1, 0, 1, 1, 1, 39},
      {"m12.dw", 256, 256, 3, 1, 1, 1, 256, 1, 40},
      {"m12.pw", 256, 256, 1, 1, 0, 1, 1, 1, 41},
      {"cls2.dw", 256, 256, 3, 1, 1, 1, 256, 1, 42},
      {"cls2.pw", 256, 4, 1, 1, 0, 1, 1, 0, 43},
      {"reg2.dw", 256, 256, 3, 1, 1, 1, 256, 1, 42},
      {"reg2.pw", 256, 8, 1, 1, 0, 1, 1, 0, 45},
      {"extra.0", 256, 64, 1, 1, 0, 1, 1, 1, 42},
      {"extra.2.dw", 64, 64, 3, 2, 1, 1, 64, 1, 47},
      {"extra.2.pw", 64, 256, 1, 1, 0, 1, 1, 1, 48},
      {"cls3", 256, 6, 3, 1, 1, 1, 1, 0, 49},
      {"reg3", 256, 12, 3, 1, 1, 1, 1, 0, 49},
  };
  return t;
}

constexpr int kHeadCls[4] = {26, 36, 44, 50};
constexpr int kHeadReg[4] = {28, 38, 46, 51};
constexpr int kHeadAnchors[4] = {3, 2, 2, 3};
constexpr int kStrides[4] = {8, 16, 32, 64};
constexpr double kMinBoxes[4][3] = {{10, 16, 24}, {32, 48, 0}, {64, 96, 0}, {128, 192, 256}};

inline size_t conv_weight_floats(const ConvSpec& s) { return (size_t)s.cout * (s.cin / s.groups) * s.k * s.k; }
inline size_t total_weight_floats() {
  size_t n = 0;
  for (int i = 0; i < kNumConv; i++) n += conv_weight_floats(conv_specs()[i]) + conv_specs()[i].cout;
  return n;
}
inline int conv_out_dim(int n, const ConvSpec& s) { return (n + 2 * s.pad - s.dil * (s.k - 1) - 1) / s.stride + 1; }

}  // namespace ufd
