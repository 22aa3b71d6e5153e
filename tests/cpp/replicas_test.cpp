// What a single-process host does at start-up on an N-GPU node (infer_server.rs:39-68 spawns ONE process):
// ufd_create_replicas parses the weights once, creates one handle per GPU and fills all but the first by ncclBroadcast.
// No Python, no torch in this process: exactly the environment of the reference's Rust binary.
//   usage: replicas_test <weights.f32> <frame.jpg> [max devices]
// Prints, per replica, "replica <i> dev <id> pci <bdf> numa <node> cpus <n> dets <count> <x_tl> <y_tl> <x_br> <y_br> <conf> ..."
// (the caller compares with the oracle).  Then the rest of the server's shape: ONE scheduler over the replicas
// (ufd_sched_config.models_320[n]; router.rs:64-71 pushing into it), 2 n + 1 camera streams placed stream i -> GPU i mod n,
// a few frames pushed to each; every delivered frame must carry its stream's replica and the first replica's detections.
// Prints "sched streams <s> frames <f> per-replica <f0> <f1> ..." and "ok".
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/ufd.h"

static std::vector<uint8_t> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

struct Sink {
  std::atomic<uint32_t> frames{0}, bad{0};
  uint32_t n_replicas = 0, expect_n = 0;
  std::vector<ufd_det> expect;
};
static void on_result(void* user, const ufd_frame_result* r) {  // (called from one completion thread per replica)
  Sink* s = static_cast<Sink*>(user);
  bool ok = r->status == UFD_OK && r->variant == 320 && r->replica == (uint32_t)(r->stream_id % s->n_replicas) && r->n == s->expect_n;
  for (uint32_t k = 0; ok && k < r->n; k++) {  // (a batch of two may pick other kernel instances than one frame: fp32 rounding)
    const float *a = &r->dets[k].x_tl, *b = &s->expect[k].x_tl;
    for (int c = 0; c < 5; c++) ok = ok && std::fabs(a[c] - b[c]) <= 1e-5f;
  }
  if (!ok) s->bad++;
  s->frames++;
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
  if (argc > 4 && std::string(argv[4]) == "dup") {
    // The n = 2 form of the broadcast on a ONE-GPU box: device 0 listed twice (the library lets that through only under
    // UFD_FLAG_TEST_DUPLICATE_DEVICES).  RCCL may refuse two ranks on one device: then the outcome is printed, not failed --
    // the point is to execute ncclCommInitAll + the group of two ncclBroadcasts if the runtime allows it at all.
    cfg.flags |= UFD_FLAG_TEST_DUPLICATE_DEVICES;
    int32_t ids[2] = {0, 0};
    ufd_model* h2[2] = {nullptr, nullptr};
    const int rc2 = ufd_create_replicas(&cfg, ids, 2, h2);
    if (rc2 != UFD_OK) {
      if (h2[0] || h2[1]) return std::printf("out[] not cleared on failure\n"), 1;
      std::printf("dup refused rc %d: %s\nok\n", rc2, ufd_last_error(nullptr));
      return 0;
    }
    std::vector<ufd_det> d0(4420), d1(4420);
    uint32_t c0 = 0, c1 = 0;
    if (ufd_infer_jpeg(h2[0], jpeg.data(), jpeg.size(), d0.data(), 4420, &c0, nullptr, nullptr) != UFD_OK ||
        ufd_infer_jpeg(h2[1], jpeg.data(), jpeg.size(), d1.data(), 4420, &c1, nullptr, nullptr) != UFD_OK)
      return std::printf("dup: infer failed\n"), 1;
    // the second handle was created from a ZERO blob: it detects what the first does only if the broadcast filled it
    bool same = c0 == c1 && c0 > 0;
    for (uint32_t k = 0; same && k < c0; k++) same = std::memcmp(&d0[k], &d1[k], sizeof(ufd_det)) == 0;
    ufd_destroy(h2[0]), ufd_destroy(h2[1]);
    if (!same) return std::printf("dup: second replica differs (%u vs %u detections)\n", c0, c1), 1;
    std::printf("dup broadcast executed: 2 ranks on device 0, %u detections on both\nok\n", c0);
    return 0;
  }
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
  // ---- one scheduler over the replicas
  {
    Sink sink;
    sink.n_replicas = (uint32_t)n;
    sink.expect.resize(4420);
    if (ufd_infer_jpeg(hs[0], jpeg.data(), jpeg.size(), sink.expect.data(), 4420, &sink.expect_n, nullptr, nullptr) != UFD_OK) return 1;
    ufd_sched_config sc{};
    sc.struct_size = sizeof(sc);
    sc.models_320 = hs.data(), sc.n_320 = (uint32_t)n, sc.placement = UFD_SCHED_PLACE_ROUND_ROBIN;
    sc.ring_slots = 8, sc.max_wait_us = 500, sc.max_inflight = 3, sc.det_cap = 4420;
    sc.on_result = on_result, sc.user = &sink;
    ufd_sched* sched = nullptr;
    if (ufd_sched_create(&sc, &sched) != UFD_OK) return std::printf("ufd_sched_create over %d replicas failed\n", n), 1;
    const uint32_t streams = 2u * (uint32_t)n + 1u, per_stream = 6;
    std::vector<uint32_t> handle(streams);
    for (uint32_t i = 0; i < streams; i++) {
      ufd_stream_config st{};
      st.struct_size = sizeof(st), st.stream_id = i, st.variant = 320;
      uint32_t where = 99;
      if (ufd_sched_add_stream(sched, &st, &handle[i]) != UFD_OK || ufd_sched_stream_replica(sched, handle[i], &where) != UFD_OK ||
          where != i % (uint32_t)n)
        return std::printf("stream %u placed on %u\n", i, where), 1;
    }
    uint32_t pushed = 0;
    for (uint32_t t = 0; t < per_stream; t++)
      for (uint32_t i = 0; i < streams; i++) {
        int prc;
        while ((prc = ufd_sched_push(sched, handle[i], jpeg.data(), jpeg.size(), t)) == UFD_E_FULL) {
        }  // (the router would drop: the test wants every frame run)
        if (prc != UFD_OK) return std::printf("push: %d\n", prc), 1;
        pushed++;
      }
    if (ufd_sched_flush(sched) != UFD_OK) return 1;
    std::vector<ufd_sched_replica_stats> rs((size_t)n);
    uint32_t nr = 0;
    if (ufd_sched_get_replica_stats(sched, 320, rs.data(), (uint32_t)n, &nr) != UFD_OK || nr != (uint32_t)n) return 1;
    ufd_sched_destroy(sched);
    if (sink.frames != pushed || sink.bad != 0) return std::printf("sched: %u of %u frames, %u wrong\n", sink.frames.load(), pushed, sink.bad.load()), 1;
    std::printf("sched streams %u frames %u per-replica", streams, pushed);
    uint64_t sum = 0;
    for (int i = 0; i < n; i++) std::printf(" %llu", (unsigned long long)rs[(size_t)i].frames), sum += rs[(size_t)i].frames;
    std::printf("\n");
    if (sum != pushed) return std::printf("per-replica counts do not add up\n"), 1;
  }
  for (auto* h : hs) ufd_destroy(h);
  std::printf("ok\n");
  return 0;
}
