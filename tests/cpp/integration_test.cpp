// C++ counterpart of the reference's only model test, infer_server/tests/integration_tests.rs:
// load UltraFace-640 with thresholds 0.5 / 0.5, run every resources/test_pics picture through
// `InferModel::run`, compare the number of faces.  The expected counts (3,6,4,3,1,1,10,0) need the
// real version-RFB-640.onnx; without it (no network here) the test runs on the weight blob given
// on the command line and checks the run() / decode->infer paths against each other instead.
//   usage: integration_test <pics dir> <weights.f32> [expect_counts]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../infercam_onnx_amd/csrc/nn.hpp"

static std::vector<uint8_t> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string dir = argv[1];
  const bool expect = argc > 3;
  std::vector<uint8_t> wbytes = slurp(argv[2]);
  const std::pair<const char*, int> images_with_num_faces[] = {
      {"bruce-mars-ZXq7xoo98b0-unsplash.jpg", 3},   {"clarke-sanders-ybPJ47PMT_M-unsplash.jpg", 6},
      {"helena-lopes-e3OUQGT9bWU-unsplash.jpg", 4}, {"kaleidico-d6rTXEtOclk-unsplash.jpg", 3},
      {"michael-dam-mEZ3PoFGs_k-unsplash.jpg", 1},  {"mika-W0i1N6FdCWA-unsplash.jpg", 1},
      {"omar-lopez-T6zu4jFhVwg-unsplash.jpg", 10},  {"ken-cheung-KonWFWUaAuk-unsplash.jpg", 0}};
  try {
    ufd::UltrafaceModel model(ufd::UltrafaceVariant::W640H480, 0.5f, 0.5f, 0, 1, nullptr,
                              reinterpret_cast<const float*>(wbytes.data()), wbytes.size() / 4);
    for (const auto& [filename, expected_num_faces] : images_with_num_faces) {
      std::vector<uint8_t> jpeg = slurp(dir + "/" + filename);
      if (jpeg.empty()) return std::printf("missing %s\n", filename), 1;
      uint32_t w = 0, h = 0;
      std::vector<uint8_t> rgb(640 * 1024 * 3);
      if (ufd_debug_decode_jpeg(model.handle(), jpeg.data(), jpeg.size(), rgb.data(), rgb.size(), &w, &h) != UFD_OK)
        return std::printf("decode failed: %s\n", ufd_last_error(model.handle())), 1;
      auto bboxes_with_confidences = model.run(ufd::RgbImage{rgb.data(), w, h, 3 * w});
      auto via_jpeg = model.infer_jpeg(jpeg.data(), jpeg.size());
      std::printf("%s %ux%u faces=%zu (reference count with the real model: %d)\n", filename, w, h,
                  bboxes_with_confidences.size(), expected_num_faces);
      if (bboxes_with_confidences != via_jpeg) return std::printf("run() and decode->infer disagree\n"), 1;
      if (expect && (int)bboxes_with_confidences.size() != expected_num_faces) return std::printf("face count mismatch\n"), 1;
      for (size_t i = 1; i < via_jpeg.size(); i++)
        if (via_jpeg[i].second > via_jpeg[i - 1].second) return std::printf("not in descending confidence\n"), 1;
      // the rest of the Inferer::run iteration (inferer.rs:38-40): same detections, and a baseline JPEG of the same size back
      auto annotated = model.annotate_jpeg(jpeg.data(), jpeg.size(), 1280, 720);
      if (annotated.bboxes_with_confidences != via_jpeg) return std::printf("annotate and decode->infer disagree\n"), 1;
      uint32_t aw = 0, ah = 0, ncoef = 0;
      if (annotated.jpeg.size() < 700 || annotated.jpeg[0] != 0xFF || annotated.jpeg[1] != 0xD8 ||
          annotated.jpeg[annotated.jpeg.size() - 2] != 0xFF || annotated.jpeg.back() != 0xD9 ||
          ufd_debug_jpeg_coefficients(annotated.jpeg.data(), annotated.jpeg.size(), nullptr, 0, &ncoef, &aw, &ah) != UFD_OK ||
          aw != w || ah != h)
        return std::printf("annotated stream is not a %ux%u JPEG\n", w, h), 1;
    }
    // the single-process multi-GPU constructor of the mirror (one replica here: RCCL forms a one-rank communicator)
    auto replicas = ufd::UltrafaceModel::new_replicas(ufd::UltrafaceVariant::W640H480, 0.5f, 0.5f, {0}, 1, nullptr,
                                                      reinterpret_cast<const float*>(wbytes.data()), wbytes.size() / 4);
    std::vector<uint8_t> jpeg = slurp(dir + "/" + images_with_num_faces[0].first);
    if (replicas.size() != 1 || replicas[0]->infer_jpeg(jpeg.data(), jpeg.size()) != model.infer_jpeg(jpeg.data(), jpeg.size()))
      return std::printf("replica disagrees with the model it was copied from\n"), 1;
  } catch (const std::exception& e) {
    return std::printf("error: %s\n", e.what()), 1;
  }
  std::printf("ok\n");
  return 0;
}
