// model_internal.hpp -- what replicas.cpp needs from model.cpp (library-internal, not part of the C ABI).
#pragma once
#include <cstddef>
#include <string>
#include <vector>

#include "../../include/ufd.h"

namespace ufd {
int create_handle(const ufd_config* cfg, ufd_model** out);  // ufd_create without the exception guard
void set_create_error(const std::string& msg);              // what ufd_last_error(NULL) reports on this thread
std::string get_create_error();
// the resident packed weight image (pointwise layers in MFMA A-operand order, ...) and the priors of a handle
void weight_buffers(ufd_model* m, float** d_weights, size_t* weight_floats, float** d_priors, size_t* prior_floats);
// get_model's parsing step (nn.rs:143-175) on the host, once: blob of total_weight_floats() + K*4 priors
void gen_priors(int W, int H, std::vector<float>& out);  // plan.cpp: the SSD priors of SURVEY 8.1 (a file without the constant)
bool load_weights_once(const ufd_config* cfg, std::vector<float>* blob, std::vector<float>* priors, std::string* why);
}  // namespace ufd
