// fuzz_host.cpp -- sanitizer target for the host-side parsers of libufacehip.so (no GPU, no HIP): the JPEG header / marker
// scanner and baseline + progressive entropy decoder (csrc/jpeg_host.cpp: what a camera or any client can feed the
// server, inferer.rs:35) and the ONNX protobuf reader (csrc/onnx_loader.cpp: the cached model file, nn.rs:143-175).
// Built with -fsanitize=address,undefined by tools/fuzz/run_host_fuzz.sh; reads a corpus directory, mutates every seed
// `rounds` times (bit flips, byte writes, truncation, splices, marker-length edits) and runs the parsers on each mutant.
// Any memory error or undefined behaviour aborts the process (non-zero exit); a clean run prints a summary.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <vector>

#include "../../infercam_onnx_amd/csrc/jpeg_host.hpp"
#include "../../infercam_onnx_amd/csrc/onnx_loader.hpp"

using namespace ufd;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
  return rng_state;
}

static std::vector<uint8_t> read_file(const std::string& p) {
  std::vector<uint8_t> v;
  if (FILE* f = fopen(p.c_str(), "rb")) {
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
  }
  return v;
}

static void mutate(std::vector<uint8_t>& d, const std::vector<std::vector<uint8_t>>& corpus) {
  if (d.empty()) return;
  const int edits = 1 + (int)(rnd() % 6);
  for (int e = 0; e < edits && !d.empty(); e++) {
    const size_t pos = rnd() % d.size();
    switch (rnd() % 7) {
      case 0: d[pos] ^= (uint8_t)(1u << (rnd() % 8)); break;
      case 1: d[pos] = (uint8_t)rnd(); break;
      case 2: d.resize(pos + 1); break;                                      // truncation
      case 3: d[pos] = 0xFF; if (pos + 1 < d.size()) d[pos + 1] = (uint8_t)(0xC0 + rnd() % 0x40); break;  // stray marker
      case 4: if (pos + 1 < d.size()) d[pos] = (uint8_t)rnd(), d[pos + 1] = (uint8_t)rnd(); break;        // length fields
      case 5: {  // splice a piece of another seed in
        const auto& o = corpus[rnd() % corpus.size()];
        if (o.empty()) break;
        const size_t from = rnd() % o.size(), n = std::min<size_t>(o.size() - from, 1 + rnd() % 256);
        d.insert(d.begin() + pos, o.begin() + from, o.begin() + from + n);
        break;
      }
      default: d.erase(d.begin() + pos, d.begin() + std::min(d.size(), pos + 1 + rnd() % 64)); break;
    }
  }
}

static long g_ok = 0, g_rejected = 0;

static void run_jpeg(const std::vector<uint8_t>& d) {
  JpegFrameDesc desc;
  std::memset(&desc, 0, sizeof(desc));
  if (jpeg_parse_header(d.data(), d.size(), &desc) != kJpegOk) {
    g_rejected++;
    return;
  }
  // the library bounds decoded frames by max_src_width/height before it allocates: same guard here
  if (desc.width > 4096 || desc.height > 4096 || desc.coef_total > (64u << 20)) {
    g_rejected++;
    return;
  }
  std::vector<int16_t> coef((size_t)desc.coef_total + 64);
  JpegFrameDesc d2;
  std::memset(&d2, 0, sizeof(d2));
  const int st = jpeg_decode_coefficients(d.data(), d.size(), &d2, coef.data(), coef.size());
  (st == kJpegOk ? g_ok : g_rejected)++;
  static GpuScanPlan plan, light;  // (large: not on the stack)
  JpegFrameDesc d3, d4;
  std::memset(&d3, 0, sizeof(d3));
  std::memset(&d4, 0, sizeof(d4));
  const int full = jpeg_plan_gpu_scan(d.data(), d.size(), &d3, &plan);
  // the check-and-hash form (no lookup tables built) must agree with the full one on everything but the tables
  const int lite = jpeg_plan_gpu_scan(d.data(), d.size(), &d4, &light, /*build_luts=*/false);
  bool same = full == lite;
  if (same && full == kJpegOk)
    same = !std::memcmp(&d3, &d4, sizeof(d3)) && !std::memcmp(&plan.scan, &light.scan, sizeof(plan.scan)) &&
           plan.n_intervals == light.n_intervals && !std::memcmp(plan.iv, light.iv, sizeof(HuffInterval) * plan.n_intervals) &&
           plan.key_hash == light.key_hash && plan.key_len == light.key_len && !std::memcmp(plan.key_bytes, light.key_bytes, plan.key_len);
  if (!same) {
    fprintf(stderr, "jpeg_plan_gpu_scan: build_luts=false disagrees with the full plan (rc %d vs %d)\n", lite, full);
    abort();
  }
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: fuzz_host <corpus dir> <rounds per seed> [seed]\n");
    return 2;
  }
  const std::string dir = argv[1];
  const int rounds = atoi(argv[2]);
  if (argc > 3) rng_state ^= strtoull(argv[3], nullptr, 0) * 0x2545F4914F6CDD1Dull + 1;
  std::vector<std::vector<uint8_t>> jpegs, onnx;
  if (DIR* dp = opendir(dir.c_str())) {
    while (dirent* e = readdir(dp)) {
      const std::string n = e->d_name;
      if (n.size() > 4 && n.substr(n.size() - 4) == ".jpg") jpegs.push_back(read_file(dir + "/" + n));
      if (n.size() > 5 && n.substr(n.size() - 5) == ".onnx") onnx.push_back(read_file(dir + "/" + n));
    }
    closedir(dp);
  }
  if (jpegs.empty()) {
    fprintf(stderr, "no .jpg seeds in %s\n", dir.c_str());
    return 2;
  }
  long runs = 0;
  for (const auto& seed : jpegs) {
    run_jpeg(seed);
    for (int r = 0; r < rounds; r++, runs++) {
      std::vector<uint8_t> m = seed;
      mutate(m, jpegs);
      run_jpeg(m);
    }
  }
  long onnx_runs = 0;
  for (const auto& seed : onnx) {
    for (int r = 0; r <= rounds; r++, onnx_runs++) {
      std::vector<uint8_t> m = seed;
      if (r) mutate(m, onnx);
      const std::string tmp = dir + "/.fuzz_tmp.onnx";
      if (FILE* f = fopen(tmp.c_str(), "wb")) {
        fwrite(m.data(), 1, m.size(), f);
        fclose(f);
      }
      std::vector<float> blob, priors;
      std::string err;
      (void)load_ultraface_onnx(tmp, 320, 240, &blob, &priors, &err);
      remove(tmp.c_str());
    }
  }
  printf("fuzz_host: %zu jpeg seeds x %d mutants = %ld runs (%ld decoded, %ld rejected), %ld onnx runs: clean\n", jpegs.size(), rounds,
         runs, g_ok, g_rejected, onnx_runs);
  return 0;
}
