"""MI355X-native face-detection hot path for infercam_onnx's infer_server.

JPEG decode -> Triangle resize + normalize -> UltraFace-RFB forward -> threshold + NMS, as
hand-written HIP kernels for gfx950 behind a C ABI (include/ufd.h).  `nn` mirrors the
reference's `infer_server/src/nn.rs` interface (InferModel / UltrafaceModel / UltrafaceVariant),
`inferer` mirrors `infer_server/src/inferer.rs` (the per-frame decode -> infer loop).
"""
__version__ = "0.1.0"
