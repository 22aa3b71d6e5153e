// replicas.cpp -- ufd_create_replicas: the multi-GPU start-up of the hot path inside ONE process.
//
// The reference server is one process whose tasks share one model (infer_server.rs:39-68 spawns the
// single Inferer at :48-50); north_star shards independent camera streams one-per-GPU and allows
// exactly one collective: the start-up weight broadcast.  A Rust host therefore calls this once and
// gets one handle per GPU:
//   1. the weight source is read ONCE (get_model, nn.rs:143-175: the .onnx is parsed on the host a
//      single time, not once per GPU);
//   2. the handle of device_ids[0] is created from it: its resident image is the packed form the
//      kernels read (MFMA A-operand order, depthwise [c][12] records, summed / stacked 1x1 pairs);
//   3. the other handles are created with a zero blob (same plan, same image layout, no parsing)
//      and receive that packed image and the priors by ncclBroadcast -- RCCL over xGMI, one
//      communicator per device from ncclCommInitAll, the broadcasts issued as one group.
// Nothing is exchanged afterwards: streams are independent (run(&self) is pure, nn.rs:178-186).
//
// RCCL is loaded with dlopen at the first call: the single-GPU library has no link-time dependency
// on it (librccl brings its own kernels and start-up cost), and a box without it fails here loudly.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/ufd.h"
#include "experiments.hpp"
#include "model_internal.hpp"
#include "onnx_loader.hpp"
#include "topology.hpp"

namespace ufd {
// get_model's parsing step on the host, once for all replicas (nn.rs:143-175): the caller's blob, the caller's path, or the
// reference's cache path
bool load_weights_once(const ufd_config* cfg, std::vector<float>* blob, std::vector<float>* priors, std::string* why) {
  const int W = cfg->variant == 640 ? 640 : 320, H = cfg->variant == 640 ? 480 : 240;
  if (cfg->weights) {
    if (cfg->weights_floats != total_weight_floats()) {
      *why = "weights blob must hold " + std::to_string(total_weight_floats()) + " floats";
      return false;
    }
    blob->assign(cfg->weights, cfg->weights + cfg->weights_floats);
    if (cfg->priors) priors->assign(cfg->priors, cfg->priors + cfg->priors_floats);
  } else {
    const std::string path = cfg->weights_path ? cfg->weights_path : default_weights_path(cfg->variant);
    if (!load_ultraface_onnx(path, W, H, blob, priors, why)) {
      *why = "cannot load " + path + ": " + *why;
      return false;
    }
  }
  if (priors->empty()) gen_priors(W, H, *priors);
  return true;
}
}  // namespace ufd

namespace {

struct Rccl {
  void* so = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;
};

Rccl* load_rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // the librccl beside the HIP runtime THIS library is bound to (a process may hold a second ROCm copy, e.g. the one
    // bundled with PyTorch: RCCL must talk to the runtime that owns the handles' memory), then the usual names
    std::vector<std::string> names;
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
      const std::string hip = info.dli_fname;
      const size_t slash = hip.rfind('/');
      if (slash != std::string::npos) names.push_back(hip.substr(0, slash) + "/librccl.so.1");
    }
    names.insert(names.end(), {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"});
    std::string last_error;
    for (const std::string& name : names) {
      r.so = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
      // (dlerror() hands its message out ONCE and clears it: read it once, right behind the failed dlopen)
      const char* e = dlerror();
      if (e) last_error = e;
    }
    if (!r.so) {
      r.why = "cannot load librccl: " + (last_error.empty() ? std::string("not found") : last_error);
      return;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(r.so, n);
      if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + n;
      return p;
    };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return &r;
}

int replicas(const ufd_config* cfg, const int32_t* device_ids, uint32_t n, ufd_model** out) {
  using namespace ufd;
  if (out)
    for (uint32_t i = 0; i < n && i < UFD_MAX_REPLICAS; i++) out[i] = nullptr;
  if (!cfg || !device_ids || !out || cfg->struct_size != sizeof(ufd_config)) {
    set_create_error("ufd_create_replicas: null argument or struct_size mismatch");
    return UFD_E_ARG;
  }
  if (n < 1 || n > UFD_MAX_REPLICAS) {
    set_create_error("ufd_create_replicas: n must be in 1.." + std::to_string(UFD_MAX_REPLICAS));
    return UFD_E_ARG;
  }
  if (cfg->variant != 640 && cfg->variant != 320) {
    set_create_error("ufd_create_replicas: variant must be 640 or 320");
    return UFD_E_ARG;
  }
  if (const char* knob = stray_experiment_knob()) {
    set_create_error(std::string(knob) + " is set, but this build of libufacehip has no experiment hooks (make EXPERIMENTS=1 builds the one that has)");
    return UFD_E_ARG;
  }
  // (UFD_FLAG_TEST_DUPLICATE_DEVICES: tests/cpp/replicas_test.cpp lists the one GPU of its box twice so that the n = 2 group of
  // ncclBroadcasts below executes at all without a second GPU -- if RCCL forms a communicator with two ranks on one device)
  const bool allow_dup = (cfg->flags & UFD_FLAG_TEST_DUPLICATE_DEVICES) != 0;
  for (uint32_t i = 0; i < n && !allow_dup; i++)
    for (uint32_t j = 0; j < i; j++)
      if (device_ids[i] == device_ids[j]) {
        set_create_error("ufd_create_replicas: device " + std::to_string(device_ids[i]) + " listed twice (one handle per GPU)");
        return UFD_E_ARG;
      }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_create_error("no HIP device: libufacehip needs a gfx950 GPU (there is no CPU fallback)");
    return UFD_E_DEVICE;
  }
  for (uint32_t i = 0; i < n; i++)
    if (device_ids[i] < 0 || device_ids[i] >= ndev) {
      set_create_error("ufd_create_replicas: device id " + std::to_string(device_ids[i]) + " out of range (" + std::to_string(ndev) + " devices)");
      return UFD_E_ARG;
    }
  // the caller's current device is left as it was found (every step below selects the device it works on)
  struct DeviceGuard {
    int saved = -1;
    DeviceGuard() {
      if (hipGetDevice(&saved) != hipSuccess) saved = -1;
    }
    ~DeviceGuard() {
      if (saved >= 0) (void)hipSetDevice(saved);
    }
  } device_guard;
  Rccl* r = load_rccl();
  if (!r->why.empty()) {
    set_create_error(r->why);
    return UFD_E_DEVICE;
  }
  // 1. the weight source, once
  std::vector<float> blob, priors;
  std::string why;
  if (!load_weights_once(cfg, &blob, &priors, &why)) {
    set_create_error(why);
    return UFD_E_WEIGHTS;
  }
  auto destroy_all = [&] {
    for (uint32_t i = 0; i < n; i++) {
      if (out[i]) ufd_destroy(out[i]);
      out[i] = nullptr;
    }
  };
  // 2./3. handle 0 from the blob, the others from zeros (identical plan and image layout, nothing parsed or packed twice
  // that matters: their image is overwritten below)
  std::vector<float> zeros(blob.size(), 0.0f);
  for (uint32_t i = 0; i < n; i++) {
    ufd_config c = *cfg;
    c.device_id = device_ids[i];
    c.weights_path = nullptr;
    c.weights = i == 0 ? blob.data() : zeros.data();
    c.weights_floats = blob.size();
    c.priors = priors.data();
    c.priors_floats = priors.size();
    const int rc = create_handle(&c, &out[i]);
    if (rc != UFD_OK) {
      const std::string msg = "replica on device " + std::to_string(device_ids[i]) + ": " + get_create_error();
      destroy_all();
      set_create_error(msg);
      return rc;
    }
  }
  // the one collective of the path: packed weight image + priors from device_ids[0] to every other device
  std::vector<ncclComm_t> comms(n, nullptr);
  std::vector<hipStream_t> streams(n, nullptr);
  std::vector<int> devs(device_ids, device_ids + n);
  std::string err;
  auto nccl_ok = [&](ncclResult_t rc, const char* what) {
    if (rc == ncclSuccess) return true;
    if (err.empty()) err = std::string(what) + ": " + r->GetErrorString(rc);
    return false;
  };
  auto hip_ok = [&](hipError_t rc, const char* what) {
    if (rc == hipSuccess) return true;
    if (err.empty()) err = std::string(what) + ": " + hipGetErrorString(rc);
    return false;
  };
  bool ok = nccl_ok(r->CommInitAll(comms.data(), (int)n, devs.data()), "ncclCommInitAll");
  for (uint32_t i = 0; ok && i < n; i++)
    ok = hip_ok(hipSetDevice(devs[i]), "hipSetDevice") && hip_ok(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking), "hipStreamCreate");
  if (ok) {
    size_t wf0 = 0, pf0 = 0;
    const bool group_open = nccl_ok(r->GroupStart(), "ncclGroupStart");
    ok = group_open;
    for (uint32_t i = 0; ok && i < n; i++) {
      float *dw = nullptr, *dp = nullptr;
      size_t wf = 0, pf = 0;
      weight_buffers(out[i], &dw, &wf, &dp, &pf);
      if (i == 0) wf0 = wf, pf0 = pf;
      if (wf != wf0 || pf != pf0) {  // (same variant and flags on every device: cannot happen)
        err = "replica images differ in size";
        ok = false;
        break;
      }
      ok = hip_ok(hipSetDevice(devs[i]), "hipSetDevice") &&
           nccl_ok(r->Broadcast(dw, dw, wf, ncclFloat, 0, comms[i], streams[i]), "ncclBroadcast(weights)") &&
           nccl_ok(r->Broadcast(dp, dp, pf, ncclFloat, 0, comms[i], streams[i]), "ncclBroadcast(priors)");
    }
    const bool ended = !group_open || nccl_ok(r->GroupEnd(), "ncclGroupEnd");  // (only a group that was opened is closed)
    ok = ok && ended;
    for (uint32_t i = 0; i < n; i++)
      if (streams[i]) ok = hip_ok(hipSetDevice(devs[i]), "hipSetDevice") && hip_ok(hipStreamSynchronize(streams[i]), "hipStreamSynchronize") && ok;
  }
  for (uint32_t i = 0; i < n; i++) {
    if (streams[i]) {
      (void)hipSetDevice(devs[i]);
      (void)hipStreamDestroy(streams[i]);
    }
    if (comms[i]) (void)r->CommDestroy(comms[i]);
  }
  if (!ok) {
    destroy_all();
    set_create_error("ufd_create_replicas: " + err);
    return UFD_E_DEVICE;
  }
  return UFD_OK;
}

}  // namespace

extern "C" int ufd_create_replicas(const ufd_config* cfg, const int32_t* device_ids, uint32_t n, ufd_model** out) {
  try {
    return replicas(cfg, device_ids, n, out);
  } catch (const std::exception& e) {
    ufd::set_create_error(std::string("exception: ") + e.what());
    return UFD_E_DEVICE;
  } catch (...) {
    ufd::set_create_error("unknown exception");
    return UFD_E_DEVICE;
  }
}
