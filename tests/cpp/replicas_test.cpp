// What a single-process host does at start-up on an N-GPU node (infer_server.rs:39-68 spawns ONE process):
// ufd_create_replicas parses the weights once, creates one handle per GPU and fills all but the first by ncclBroadcast.
// No Python, no torch in this process: exactly the environment of the reference's Rust binary.
//   usage: replicas_test <weights.f32> <frame.jpg> [max devices]
// Prints, per replica, "replica <i> dev <id> pci <bdf> numa <node> cpus <n> dets <count> <x_tl> <y_tl> <x_br> <y_br> <conf> ..."
// (the caller compares with the oracle) and "ok".
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/ufd.h"

static std::vector<uint8_t> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::vector<uint8_t> wbytes = slurp(argv[1]), jpeg = slurp(argv[2]);
  const int max_dev = argc > 3 ? std::atoi(argv[3]) : 8;
  ufd_config cfg{};
  cfg.struct_size = sizeof(cfg);
  cfg.variant = 320, cfg.max_iou = 0.5f, cfg.min_confidence = 0.5f, cfg.max_batch = 2;
  cfg.max_src_width = 320, cfg.max_src_height = 240;
  cfg.weights = reinterpret_cast<const float*>(wbytes.data());
  cfg.weights_floats = wbytes.size() / 4;
  // every GPU of the box: the largest n the library accepts (an id beyond the device count is UFD_E_ARG, out[] untouched NULL)
  std::vector<ufd_model*> hs;
  int n = max_dev, rc = UFD_E_ARG;
  for (; n >= 1; n--) {
    std::vector<int32_t> ids(n);
    for (int i = 0; i < n; i++) ids[i] = i;
    hs.assign(n, reinterpret_cast<ufd_model*>(0x1));
    rc = ufd_create_replicas(&cfg, ids.data(), (uint32_t)n, hs.data());
    if (rc == UFD_OK) break;
    for (auto* h : hs)
      if (h) return std::printf("out[] not cleared on failure\n"), 1;
    if (rc != UFD_E_ARG) return std::printf("ufd_create_replicas(%d): %d %s\n", n, rc, ufd_last_error(nullptr)), 1;
  }
  if (rc != UFD_OK) return std::printf("no replica set accepted: %s\n", ufd_last_error(nullptr)), 1;
  for (int i = 0; i < n; i++) {
    int32_t dev = -1, node = -2;
    uint32_t ncpu = 0, cnt = 0;
    char bdf[64] = "", cpus[512] = "";
    if (ufd_model_placement(hs[i], &dev, &node, &ncpu, bdf, sizeof(bdf), cpus, sizeof(cpus)) != UFD_OK || dev != i) return std::printf("placement\n"), 1;
    std::vector<ufd_det> dets(4420);
    rc = ufd_infer_jpeg(hs[i], jpeg.data(), jpeg.size(), dets.data(), (uint32_t)dets.size(), &cnt, nullptr, nullptr);
    if (rc != UFD_OK) return std::printf("replica %d: %d %s\n", i, rc, ufd_last_error(hs[i])), 1;
    std::printf("replica %d dev %d pci %s numa %d cpus %u dets %u", i, dev, bdf, node, ncpu, cnt);
    for (uint32_t k = 0; k < cnt; k++) std::printf(" %.9g %.9g %.9g %.9g %.9g", dets[k].x_tl, dets[k].y_tl, dets[k].x_br, dets[k].y_br, dets[k].conf);
    std::printf("\n");
  }
  for (auto* h : hs) ufd_destroy(h);
  std::printf("ok\n");
  return 0;
}
