#!/usr/bin/env python3
"""Round 6: what does torch.distributed's "nccl" backend in the process cost the handle?  (torch brings its own HIP runtime
and RCCL; every HSA queue beyond the handle's four that merely exists costs throughput -- docs/EXPERIMENTS.md, rounds 1, 4.)
World size 1 on the box's one GPU, bench.py's N > 1 sequence, one mode per process:
  plain      no process group (bench.py at N = 1)
  nccl       init_process_group("nccl"), weight broadcast on the device, dist.barrier() around the timed region (bench.py at N > 1 until round 5)
  nccl+gloo  the weight broadcast over nccl, then that group destroyed; barriers and the max-over-ranks over a gloo group
  gloo       everything over gloo (no GPU work by torch at all)
Prints the 20-step sample (median of 5) and a 300-step steady state, frames/s."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from infercam_onnx_amd import nn, synth  # noqa: E402

mode = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
B, DEPTH = 32, 6
weights = synth.synthetic_weights()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
barrier = lambda: None  # noqa: E731
if mode in ("nccl", "nccl+gloo"):
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    t = torch.from_numpy(weights.copy()).to(dev)
    dist.broadcast(t, src=0)
    weights = t.cpu().numpy()
    if mode == "nccl":
        barrier = dist.barrier
    else:
        del t
        dist.destroy_process_group()
        torch.cuda.empty_cache()
        dist.init_process_group("gloo", rank=0, world_size=1)
        barrier = dist.barrier
elif mode == "gloo":
    dist.init_process_group("gloo", rank=0, world_size=1)
    t = torch.from_numpy(weights.copy())
    dist.broadcast(t, src=0)
    weights = t.numpy()
    barrier = dist.barrier
jpegs = synth.synth_jpeg_pool(0, 256, 640, 480, quality=90, subsampling="4:2:0")
m = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, max_batch=B, weights=np.ascontiguousarray(weights), priors=synth.gen_priors(640, 480),
                      max_src=(640, 480), det_cap=256, host_threads=16)
bts = [m._prep_batch(jpegs[i * B:(i + 1) * B]) for i in range(8)]


def run_steps(k):
    infl = []
    for s in range(k):
        if len(infl) >= DEPTH:
            m.wait(infl.pop(0), collect=False)
        infl.append(m.submit_jpeg_batch(bts[s % len(bts)]))
    for t_ in infl:
        m.wait(t_, collect=False)


samples = []
for _ in range(5):
    run_steps(5)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(20)
    torch.cuda.synchronize()
    barrier()
    samples.append(B * 20 / (time.perf_counter() - t0))
barrier()
t0 = time.perf_counter()
run_steps(300)
torch.cuda.synchronize()
steady = B * 300 / (time.perf_counter() - t0)
print("%-10s 20-step sample median %.0f (%s) | steady state %.0f frames/s" % (mode, statistics.median(samples), " ".join("%.0f" % v for v in samples), steady), flush=True)
m.close()
if dist.is_initialized():
    dist.destroy_process_group()
