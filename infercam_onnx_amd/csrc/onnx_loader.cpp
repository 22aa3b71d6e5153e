// onnx_loader.cpp -- minimal protobuf wire-format reader for the one model family this library
// runs (replaces tract's ONNX front end, infer_server/src/nn.rs:166-172; field numbers from
// onnx.proto3).  No protobuf runtime, no schema: varint / length-delimited /
// fixed32 / fixed64 fields are walked by hand.
#include "onnx_loader.hpp"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>

#include "topology.hpp"

namespace ufd {
namespace {

struct Span {
  const uint8_t* p = nullptr;
  const uint8_t* end = nullptr;
  bool ok = true;
  bool more() const { return ok && p < end; }
};

uint64_t varint(Span& s) {
  uint64_t v = 0;
  for (int shift = 0; shift < 64; shift += 7) {
    if (s.p >= s.end) {
      s.ok = false;
      return 0;
    }
    uint8_t b = *s.p++;
    v |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) return v;
  }
  s.ok = false;
  return 0;
}

struct Field {
  int num = 0, wire = 0;
  uint64_t v = 0;  // varint / fixed
  Span sub;        // length-delimited payload
};

bool next_field(Span& s, Field* f) {
  if (!s.more()) return false;
  uint64_t key = varint(s);
  if (!s.ok) return false;
  f->num = (int)(key >> 3);
  f->wire = (int)(key & 7);
  switch (f->wire) {
    case 0: f->v = varint(s); break;
    case 1:
      if (s.end - s.p < 8) return s.ok = false;
      std::memcpy(&f->v, s.p, 8);
      s.p += 8;
      break;
    case 2: {
      uint64_t n = varint(s);
      if (!s.ok || (uint64_t)(s.end - s.p) < n) return s.ok = false;
      f->sub.p = s.p;
      f->sub.end = s.p + n;
      f->sub.ok = true;
      s.p += n;
      break;
    }
    case 5: {
      if (s.end - s.p < 4) return s.ok = false;
      uint32_t x;
      std::memcpy(&x, s.p, 4);
      f->v = x;
      s.p += 4;
      break;
    }
    default: return s.ok = false;
  }
  return s.ok;
}

std::string str(const Span& s) { return std::string(reinterpret_cast<const char*>(s.p), s.end - s.p); }

struct TensorData {
  std::vector<int64_t> dims;
  int dtype = 0;
  std::vector<float> f;
  std::vector<int64_t> i64;
  std::string name;
};

void repeated_i64(const Field& f, std::vector<int64_t>* out) {
  if (f.wire == 0) {
    out->push_back((int64_t)f.v);
  } else if (f.wire == 2) {
    Span s = f.sub;
    while (s.more()) out->push_back((int64_t)varint(s));
  }
}

bool parse_tensor(Span s, TensorData* t) {
  Field f;
  Span raw;
  while (next_field(s, &f)) {
    switch (f.num) {
      case 1: repeated_i64(f, &t->dims); break;
      case 2: t->dtype = (int)f.v; break;
      case 4:
        if (f.wire == 2) {
          size_t n = (f.sub.end - f.sub.p) / 4;
          size_t o = t->f.size();
          t->f.resize(o + n);
          std::memcpy(t->f.data() + o, f.sub.p, n * 4);
        } else if (f.wire == 5) {
          float x;
          uint32_t u = (uint32_t)f.v;
          std::memcpy(&x, &u, 4);
          t->f.push_back(x);
        }
        break;
      case 7: repeated_i64(f, &t->i64); break;
      case 8: t->name = str(f.sub); break;
      case 9: raw = f.sub; break;
      default: break;
    }
  }
  if (!s.ok) return false;
  if (raw.p && t->dtype == 1) {
    size_t n = (raw.end - raw.p) / 4;
    t->f.resize(n);
    std::memcpy(t->f.data(), raw.p, n * 4);
  } else if (raw.p && t->dtype == 7) {
    size_t n = (raw.end - raw.p) / 8;
    t->i64.resize(n);
    std::memcpy(t->i64.data(), raw.p, n * 8);
  }
  return true;
}

struct Attr {
  std::string name;
  float f = 0;
  int64_t i = 0;
  std::vector<int64_t> ints;
  TensorData t;
  bool has_t = false;
};

struct Node {
  std::string op;
  std::vector<std::string> in, out;
  std::vector<Attr> attrs;
  const Attr* attr(const char* n) const {
    for (const auto& a : attrs)
      if (a.name == n) return &a;
    return nullptr;
  }
};

bool parse_attr(Span s, Attr* a) {
  Field f;
  while (next_field(s, &f)) {
    switch (f.num) {
      case 1: a->name = str(f.sub); break;
      case 2: {
        uint32_t u = (uint32_t)f.v;
        std::memcpy(&a->f, &u, 4);
        break;
      }
      case 3: a->i = (int64_t)f.v; break;
      case 5: a->has_t = parse_tensor(f.sub, &a->t); break;
      case 8: repeated_i64(f, &a->ints); break;
      default: break;
    }
  }
  return s.ok;
}

bool parse_node(Span s, Node* n) {
  Field f;
  while (next_field(s, &f)) {
    switch (f.num) {
      case 1: n->in.push_back(str(f.sub)); break;
      case 2: n->out.push_back(str(f.sub)); break;
      case 4: n->op = str(f.sub); break;
      case 5: {
        Attr a;
        if (!parse_attr(f.sub, &a)) return false;
        n->attrs.push_back(std::move(a));
        break;
      }
      default: break;
    }
  }
  return s.ok;
}

bool fail(std::string* why, const std::string& msg) {
  if (why) *why = msg;
  return false;
}

}  // namespace

bool load_ultraface_onnx(const std::string& path, int width, int height, std::vector<float>* blob,
                         std::vector<float>* priors, std::string* why) {
  std::vector<uint8_t> bytes;
  {
    FILE* fp = std::fopen(path.c_str(), "rb");
    if (!fp) return fail(why, "file not found (the reference downloads it; no network here)");
    std::fseek(fp, 0, SEEK_END);
    long n = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    if (n <= 0 || n > (64L << 20)) {
      std::fclose(fp);
      return fail(why, "implausible file size");
    }
    bytes.resize((size_t)n);
    size_t got = std::fread(bytes.data(), 1, bytes.size(), fp);
    std::fclose(fp);
    if (got != bytes.size()) return fail(why, "short read");
  }
  Span model{bytes.data(), bytes.data() + bytes.size(), true};
  Span graph;
  Field f;
  while (next_field(model, &f))
    if (f.num == 7 && f.wire == 2) graph = f.sub;
  if (!model.ok || !graph.p) return fail(why, "not an ONNX ModelProto (no graph)");

  std::map<std::string, TensorData> tensors;
  std::vector<Node> nodes;
  while (next_field(graph, &f)) {
    if (f.num == 1 && f.wire == 2) {
      Node n;
      if (!parse_node(f.sub, &n)) return fail(why, "malformed NodeProto");
      if (n.op == "Constant" && !n.out.empty()) {
        const Attr* a = n.attr("value");
        if (a && a->has_t) tensors[n.out[0]] = a->t;
      }
      nodes.push_back(std::move(n));
    } else if (f.num == 5 && f.wire == 2) {
      TensorData t;
      if (!parse_tensor(f.sub, &t)) return fail(why, "malformed TensorProto");
      tensors[t.name] = std::move(t);
    }
  }
  if (!graph.ok) return fail(why, "malformed GraphProto");

  // BatchNormalization nodes keyed by their data input
  std::map<std::string, const Node*> bn_of;
  for (const auto& n : nodes)
    if (n.op == "BatchNormalization" && n.in.size() >= 5) bn_of[n.in[0]] = &n;

  const ConvSpec* specs = conv_specs();
  blob->clear();
  blob->reserve(total_weight_floats());
  int ci = 0;
  for (const auto& n : nodes) {
    if (n.op != "Conv") continue;
    if (ci >= kNumConv) return fail(why, "more than 52 Conv nodes: not UltraFace-RFB");
    const ConvSpec& s = specs[ci];
    if (n.in.size() < 2 || !tensors.count(n.in[1])) return fail(why, std::string("Conv ") + s.name + ": weight initializer missing");
    const TensorData& w = tensors[n.in[1]];
    const std::vector<int64_t> want = {s.cout, s.cin / s.groups, s.k, s.k};
    if (w.dims != want || w.f.size() != conv_weight_floats(s))
      return fail(why, std::string("Conv #") + std::to_string(ci) + " (" + s.name + "): weight shape does not match UltraFace-RFB");
    auto ints_eq = [&](const char* name, int64_t v, size_t cnt) {
      const Attr* a = n.attr(name);
      if (!a) return true;  // defaults are validated by the weight shape / output sizes
      if (a->ints.size() != cnt) return false;
      for (int64_t x : a->ints)
        if (x != v) return false;
      return true;
    };
    const Attr* grp = n.attr("group");
    if ((grp ? grp->i : 1) != s.groups || !ints_eq("strides", s.stride, 2) || !ints_eq("dilations", s.dil, 2) ||
        !ints_eq("pads", s.pad, 4) || !ints_eq("kernel_shape", s.k, 2))
      return fail(why, std::string("Conv #") + std::to_string(ci) + " (" + s.name + "): attributes do not match UltraFace-RFB");
    std::vector<float> wv = w.f, bv(s.cout, 0.0f);
    if (n.in.size() >= 3 && !n.in[2].empty()) {
      if (!tensors.count(n.in[2]) || tensors[n.in[2]].f.size() != (size_t)s.cout) return fail(why, "Conv bias shape mismatch");
      bv = tensors[n.in[2]].f;
    }
    auto bn = n.out.empty() ? bn_of.end() : bn_of.find(n.out[0]);
    if (bn != bn_of.end()) {
      const Node& b = *bn->second;
      const TensorData *g = nullptr, *be = nullptr, *mu = nullptr, *var = nullptr;
      const TensorData** dst[4] = {&g, &be, &mu, &var};
      for (int j = 0; j < 4; j++) {
        auto it = tensors.find(b.in[1 + j]);
        if (it == tensors.end() || it->second.f.size() != (size_t)s.cout) return fail(why, "BatchNormalization parameter missing");
        *dst[j] = &it->second;
      }
      const Attr* eps = b.attr("epsilon");
      const float e = eps ? eps->f : 1e-5f;
      const size_t per = conv_weight_floats(s) / s.cout;
      for (int co = 0; co < s.cout; co++) {
        // w' = w * gamma / sqrt(var + eps);  b' = beta + (b - mean) * gamma / sqrt(var + eps)
        const float sc = g->f[co] / std::sqrt(var->f[co] + e);
        for (size_t j = 0; j < per; j++) wv[co * per + j] *= sc;
        bv[co] = be->f[co] + (bv[co] - mu->f[co]) * sc;
      }
    }
    blob->insert(blob->end(), wv.begin(), wv.end());
    blob->insert(blob->end(), bv.begin(), bv.end());
    ci++;
  }
  if (ci != kNumConv) return fail(why, "found " + std::to_string(ci) + " Conv nodes, UltraFace-RFB has 52");

  // priors constant [1,K,4] / [K,4]
  priors->clear();
  int K = 0;
  for (int i = 0; i < 4; i++)
    K += ((width + kStrides[i] - 1) / kStrides[i]) * ((height + kStrides[i] - 1) / kStrides[i]) * kHeadAnchors[i];
  for (const auto& kv : tensors) {
    const TensorData& t = kv.second;
    if (t.f.size() == (size_t)K * 4 && t.dims.size() >= 2 && t.dims.back() == 4 && t.dims[t.dims.size() - 2] == K) {
      *priors = t.f;
      break;
    }
  }
  if (priors->empty()) {
    // torch's tracer evaluates `priors[..., :2]` / `priors[..., 2:]` at export time, so an exported
    // SSD graph holds the priors as [1,K,2] halves: the constant ADDED to the scaled centre offsets
    // is (cx, cy), the constant MULTIPLIED with them (and with exp(size offsets)) is (w, h).
    const TensorData *centres = nullptr, *sizes = nullptr;
    auto half = [&](const std::string& name) -> const TensorData* {
      auto it = tensors.find(name);
      if (it == tensors.end()) return nullptr;
      const TensorData& t = it->second;
      if (t.f.size() != (size_t)K * 2 || t.dims.size() < 2 || t.dims.back() != 2 || t.dims[t.dims.size() - 2] != K) return nullptr;
      return &t;
    };
    for (const auto& n : nodes) {
      if (n.op != "Add" && n.op != "Mul") continue;
      for (const auto& in : n.in)
        if (const TensorData* t = half(in)) (n.op == "Add" ? centres : sizes) = t;
    }
    if (centres && sizes) {
      priors->resize((size_t)K * 4);
      for (int k = 0; k < K; k++) {
        (*priors)[k * 4 + 0] = centres->f[k * 2 + 0];
        (*priors)[k * 4 + 1] = centres->f[k * 2 + 1];
        (*priors)[k * 4 + 2] = sizes->f[k * 2 + 0];
        (*priors)[k * 4 + 3] = sizes->f[k * 2 + 1];
      }
    }
  }
  return true;
}


// ---- get_model's choice of source (nn.rs:143-175): the caller's blob, the caller's path, or the reference's cache path
std::string default_weights_path(int variant) {
  // dirs::cache_dir()/infercam_onnx/ultraface-RFB-{640,320}.onnx (nn.rs:144-156)
  const char* xdg = std::getenv("XDG_CACHE_HOME");
  std::string base;
  if (xdg && *xdg) {
    base = xdg;
  } else {
    const char* home = std::getenv("HOME");
    base = std::string(home ? home : ".") + "/.cache";
  }
  return base + "/infercam_onnx/ultraface-RFB-" + std::to_string(variant) + ".onnx";
}

}  // namespace ufd
