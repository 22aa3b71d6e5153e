// onnx_loader.hpp -- reads the UltraFace-RFB weights out of the ONNX file the reference loads with
// tract_onnx::onnx().model_for_path(..) (infer_server/src/nn.rs:164-172).
#pragma once
#include <string>
#include <vector>

namespace ufd {

// Walks graph.node in order, collects the 52 Conv nodes (folding a following BatchNormalization
// into (w, b)), validates them against the UltraFace-RFB topology table and returns the packed
// blob (for each conv w[cout][cin/g][k][k] then b[cout]).  priors: the [K,4] constant embedded
// in the graph if one is found (else left empty: the caller regenerates them).
bool load_ultraface_onnx(const std::string& path, int width, int height, std::vector<float>* blob,
                         std::vector<float>* priors, std::string* why);

// dirs::cache_dir()/infercam_onnx/ultraface-RFB-{640,320}.onnx (nn.rs:144-156)
std::string default_weights_path(int variant);

}  // namespace ufd
