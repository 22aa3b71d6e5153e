// experiments.hpp -- measurement knobs read from the environment (docs/EXPERIMENTS.md: ablations, run-it-twice, A/B of a
// launch parameter).  They change what a handle computes or how, so the library that ships does not have them: they are
// compiled in only by `make EXPERIMENTS=1` (-DUFD_EXPERIMENTS, libufacehip_exp.so, used by tools/ab/*), and ufd_create /
// ufd_create_replicas of the ordinary build REFUSE to make a handle while one of them is set instead of ignoring it.
#pragma once
#include <cstdlib>

namespace ufd {

inline const char* const* experiment_knobs() {
  static const char* const k[] = {"UFD_ABLATE_LAYERS", "UFD_REPEAT_ENTROPY", "UFD_EXTEND_ROUNDS", "UFD_SUB_SMALL_BYTES", "UFD_PLAN_PARALLEL",
                                  "UFD_NO_DUAL",       "UFD_BAND_SMALL",     "UFD_TEST_DUPLICATE_DEVICES", "UFD_GATE_LAYER", nullptr};
  return k;
}

// The value of an experiment knob, or null (always null in the ordinary build).
inline const char* experiment_env(const char* name) {
#ifdef UFD_EXPERIMENTS
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Ordinary build: the name of an experiment knob that is set in the environment (the handle must not be created), or null.
inline const char* stray_experiment_knob() {
#ifdef UFD_EXPERIMENTS
  return nullptr;
#else
  for (const char* const* k = experiment_knobs(); *k; k++)
    if (std::getenv(*k)) return *k;
  return nullptr;
#endif
}

}  // namespace ufd
