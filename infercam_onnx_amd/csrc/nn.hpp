// nn.hpp -- header-only C++ mirror of the reference's model interface over the C ABI
// (infer_server/src/nn.rs: Bbox :12, trait InferModel :24-26, UltrafaceVariant :29-42,
// UltrafaceModel::new :55, InferModel::run :178-186; Inferer decode->infer, inferer.rs:35-37).
// Same names, argument meaning and error behaviour (errors -> std::runtime_error where the
// reference returns anyhow::Err).
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ufd.h"

namespace ufd {

using Bbox = std::array<float, 4>;  // [x_top_left, y_top_left, x_bottom_right, y_bottom_right], relative

struct RgbImage {  // image::RgbImage: borrowed interleaved RGB8
  const uint8_t* data;
  uint32_t width, height, pitch;
};

class InferModel {
 public:
  virtual ~InferModel() = default;
  virtual std::vector<std::pair<Bbox, float>> run(const RgbImage& input) = 0;
};

enum class UltrafaceVariant { W640H480 = 640, W320H240 = 320 };
inline std::pair<uint32_t, uint32_t> width_height(UltrafaceVariant v) {
  return v == UltrafaceVariant::W640H480 ? std::make_pair(640u, 480u) : std::make_pair(320u, 240u);
}

class UltrafaceModel : public InferModel {
 public:
  // weights_path == nullptr -> the reference's cache path (nn.rs:144-156)
  UltrafaceModel(UltrafaceVariant variant, float max_iou, float min_confidence, int device_id = 0,
                 uint32_t max_batch = 1, const char* weights_path = nullptr, const float* weights = nullptr,
                 size_t weights_floats = 0) {
    ufd_config cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.variant = static_cast<uint32_t>(variant);
    cfg.max_iou = max_iou;
    cfg.min_confidence = min_confidence;
    cfg.device_id = device_id;
    cfg.max_batch = max_batch;
    cfg.weights_path = weights_path;
    cfg.weights = weights;
    cfg.weights_floats = weights_floats;
    if (ufd_create(&cfg, &h_) != UFD_OK) throw std::runtime_error(std::string("ufd_create: ") + ufd_last_error(nullptr));
  }
  ~UltrafaceModel() override { ufd_destroy(h_); }
  UltrafaceModel(const UltrafaceModel&) = delete;
  UltrafaceModel& operator=(const UltrafaceModel&) = delete;

  std::vector<std::pair<Bbox, float>> run(const RgbImage& input) override {
    std::vector<ufd_det> out(kCap);
    uint32_t n = 0;
    int rc = ufd_infer_rgb(h_, input.data, input.width, input.height, input.pitch, out.data(), kCap, &n);
    if (rc != UFD_OK && rc != UFD_E_TRUNCATED) throw std::runtime_error(std::string("ufd_infer_rgb: ") + ufd_last_error(h_));
    return collect(out, n);
  }
  // Inferer::run: decompress_image(jpeg) then infer_faces(&image)
  std::vector<std::pair<Bbox, float>> infer_jpeg(const uint8_t* jpeg, size_t len) {
    std::vector<ufd_det> out(kCap);
    uint32_t n = 0;
    int rc = ufd_infer_jpeg(h_, jpeg, len, out.data(), kCap, &n, nullptr, nullptr);
    if (rc != UFD_OK && rc != UFD_E_TRUNCATED) throw std::runtime_error(std::string("ufd_infer_jpeg: ") + ufd_last_error(h_));
    return collect(out, n);
  }
  ufd_model* handle() { return h_; }

 private:
  static constexpr uint32_t kCap = 1024;
  static std::vector<std::pair<Bbox, float>> collect(const std::vector<ufd_det>& d, uint32_t n) {
    std::vector<std::pair<Bbox, float>> r;
    for (uint32_t i = 0; i < n && i < kCap; i++) r.push_back({Bbox{d[i].x_tl, d[i].y_tl, d[i].x_br, d[i].y_br}, d[i].conf});
    return r;
  }
  ufd_model* h_ = nullptr;
};

}  // namespace ufd
