"""Probe for the N > 1 path of bench.py on a one-GPU box: torch.distributed's "nccl" backend (= torch's bundled RCCL on
torch's bundled HIP runtime) with world_size 1, its collectives issued BEFORE and AFTER libufacehip.so (linked against
the system ROCm runtime) has created a handle and run a batch in the same process -- the order bench.py uses.  Prints
"nccl coexist ok" or raises.  Not a measurement; run under gpurun: python tools/nccl_coexist_probe.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from infercam_onnx_amd import nn, parallel, synth  # noqa: E402

w = parallel.broadcast_weights(synth.synthetic_weights(), dist, device=dev)
assert w.dtype == np.float32 and w.size > 0
m = nn.UltrafaceModel(nn.UltrafaceVariant.W320H240, 0.5, 0.5, device_id=0, max_batch=4, weights=w,
                      priors=synth.gen_priors(320, 240), max_src=(320, 240), det_cap=64)
jpegs = synth.synth_jpeg_pool(0, 4, 320, 240, quality=90, subsampling="4:2:0")
dets = m.infer_jpeg_batch(jpegs)
dist.barrier()
torch.cuda.synchronize()
t = parallel.max_over_ranks(1.25, dist, device=dev)
assert t == 1.25
mine = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, device=dev)
out = [torch.empty_like(mine)]
dist.all_gather(out, mine)
assert out[0].cpu().tolist() == [1.0, 2.0, 3.0]
dist.barrier()
dist.destroy_process_group()
print("nccl coexist ok", "(ran a batch: %s)" % (dets is not None))
