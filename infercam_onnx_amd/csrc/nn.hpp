// nn.hpp -- header-only C++ mirror of the reference's model interface over the C ABI
// (infer_server/src/nn.rs: Bbox :12, trait InferModel :24-26, UltrafaceVariant :29-42,
// UltrafaceModel::new :55, InferModel::run :178-186; Inferer decode->infer, inferer.rs:35-37).
// Same names, argument meaning and error behaviour (errors -> std::runtime_error where the
// reference returns anyhow::Err).
#pragma once
#include <array>
#include <cstdint>
#include <memory>
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
  // One model per GPU of `devices` inside ONE process (infer_server.rs:39-68 is one process): UltrafaceModel::new once,
  // the packed weights broadcast to the other GPUs over RCCL (ufd_create_replicas).  Stream i -> models[i % devices.size()].
  static std::vector<std::unique_ptr<UltrafaceModel>> new_replicas(UltrafaceVariant variant, float max_iou, float min_confidence,
                                                                  const std::vector<int32_t>& devices, uint32_t max_batch = 1,
                                                                  const char* weights_path = nullptr, const float* weights = nullptr,
                                                                  size_t weights_floats = 0) {
    ufd_config cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.variant = static_cast<uint32_t>(variant);
    cfg.max_iou = max_iou;
    cfg.min_confidence = min_confidence;
    cfg.max_batch = max_batch;
    cfg.weights_path = weights_path;
    cfg.weights = weights;
    cfg.weights_floats = weights_floats;
    std::vector<ufd_model*> hs(devices.size(), nullptr);
    if (ufd_create_replicas(&cfg, devices.data(), (uint32_t)devices.size(), hs.data()) != UFD_OK)
      throw std::runtime_error(std::string("ufd_create_replicas: ") + ufd_last_error(nullptr));
    std::vector<std::unique_ptr<UltrafaceModel>> models;
    for (ufd_model* h : hs) models.emplace_back(new UltrafaceModel(h));
    return models;
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
  // Inferer::run, the whole iteration (inferer.rs:35-40): decompress_image -> infer_faces -> draw_bboxes_on_image(image,
  // boxes, width, height) -> compress_image(&frame, 95, Sub2x2).  width / height = the slot's labels (router.rs:66-67).
  struct Annotated {
    std::vector<std::pair<Bbox, float>> bboxes_with_confidences;
    std::vector<uint8_t> jpeg;
  };
  Annotated annotate_jpeg(const uint8_t* jpeg, size_t len, uint32_t width, uint32_t height, uint32_t quality = 95,
                          bool multipart = false) {
    std::vector<ufd_det> out(kCap);
    uint32_t n = 0;
    int32_t st = 0;
    uint32_t w = 0, h = 0;
    if (ufd_debug_jpeg_coefficients(jpeg, len, nullptr, 0, &n, &w, &h) != UFD_OK) throw std::runtime_error("not a JPEG");
    Annotated r;
    r.jpeg.resize(ufd_encode_bound(w, h));
    size_t off = 0, out_len = 0;
    ufd_annotate a{};
    a.struct_size = sizeof(a);
    a.label_width = (float)width, a.label_height = (float)height;
    a.quality = quality, a.flags = multipart ? UFD_ANNOT_MULTIPART : 0;
    a.jpeg_out = r.jpeg.data(), a.jpeg_cap = r.jpeg.size(), a.jpeg_off = &off, a.jpeg_len = &out_len;
    n = 0;
    int rc = ufd_annotate_jpeg_batch(h_, &jpeg, &len, 1, &a, out.data(), kCap, &n, &st);
    if (rc != UFD_OK || (st != UFD_OK && st != UFD_E_TRUNCATED))
      throw std::runtime_error(std::string("ufd_annotate_jpeg_batch: ") + ufd_last_error(h_));
    r.jpeg.erase(r.jpeg.begin(), r.jpeg.begin() + (long)off);
    r.jpeg.resize(out_len);
    r.bboxes_with_confidences = collect(out, n);
    return r;
  }
  ufd_model* handle() { return h_; }

 private:
  explicit UltrafaceModel(ufd_model* adopted) : h_(adopted) {}
  static constexpr uint32_t kCap = 1024;
  static std::vector<std::pair<Bbox, float>> collect(const std::vector<ufd_det>& d, uint32_t n) {
    std::vector<std::pair<Bbox, float>> r;
    for (uint32_t i = 0; i < n && i < kCap; i++) r.push_back({Bbox{d[i].x_tl, d[i].y_tl, d[i].x_br, d[i].y_br}, d[i].conf});
    return r;
  }
  ufd_model* h_ = nullptr;
};

}  // namespace ufd
