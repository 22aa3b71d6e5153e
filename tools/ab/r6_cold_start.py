#!/usr/bin/env python3
"""Round 6: why is bench.py's 20-step sample 0.86 of steady state when the same loop inside a busy process is 0.92-0.95?
One handle; the driver's sample (W warm-up steps, drain, 20 timed steps, drain) after: an idle pause of S seconds, W = 1, 5, 20 or
50.  Prints frames/s per case (three repeats each, in rotation): profiles/r6b/cold_start.txt.
With the argument `who`: also tells GPU from host -- the 0.5 s in front of the region spent sleeping, spinning on the CPU, or
keeping the GPU busy with a torch matmul loop (profiles/r6b/cold_start2.txt; the 64 MB torch tensor and its BLAS kernels in the
process cost the handle 20 % of its rate by themselves -- compare the steady-state lines of the two files -- so the three
cases are comparable with one another, not with the first file)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from infercam_onnx_amd import nn, synth  # noqa: E402

B, DEPTH = 32, 6
weights, priors = synth.synthetic_weights(), synth.gen_priors(640, 480)
jpegs = synth.synth_jpeg_pool(0, 256, 640, 480, quality=90, subsampling="4:2:0")


def run_steps(m, bts, k):
    infl = []
    for s in range(k):
        if len(infl) >= DEPTH:
            m.wait(infl.pop(0), collect=False)
        infl.append(m.submit_jpeg_batch(bts[s % len(bts)]))
    for t in infl:
        m.wait(t, collect=False)


WHO = len(sys.argv) > 1 and sys.argv[1] == "who"
X = torch.randn(4096, 4096, device="cuda") if WHO else None


def sample(m, bts, warm, idle, how="sleep"):
    if idle and how == "sleep":
        time.sleep(idle)
    elif idle and how == "cpu-spin":  # the calling CPU stays busy, the GPU idles
        t = time.perf_counter()
        while time.perf_counter() - t < idle:
            pass
    elif idle and how == "gpu-busy":  # the GPU stays busy with somebody else's work (a torch matmul loop), the handle idles
        t = time.perf_counter()
        while time.perf_counter() - t < idle:
            (X @ X).sum().item()
    run_steps(m, bts, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(m, bts, 20)
    torch.cuda.synchronize()
    return B * 20 / (time.perf_counter() - t0)


for profile in (False,):
    m = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, max_batch=B, weights=weights, priors=priors, max_src=(640, 480), det_cap=256,
                          profile=profile, host_threads=16)
    if profile:
        m.profile_sampling(1 << 30)
    bts = [m._prep_batch(jpegs[i * B:(i + 1) * B]) for i in range(8)]
    first = sample(m, bts, 5, 0)  # the very first batches of the handle
    print("profile armed %s: first sample of a new handle (5 warm-up steps): %.0f" % (profile, first), flush=True)
    for rnd in range(3):
        for warm, idle in ((5, 0), (5, 0.05), (5, 0.5), (5, 3.0), (50, 3.0), (1, 3.0), (20, 0)):
            print("profile armed %s  idle %.2f s  warm-up %2d steps: %.0f" % (profile, idle, warm, sample(m, bts, warm, idle)), flush=True)
        for how in (("sleep", "cpu-spin", "gpu-busy") if WHO else ()):
            print("  %-8s 0.5 s, then warm-up 5 steps: %.0f" % (how, sample(m, bts, 5, 0.5, how)), flush=True)
    t0 = time.perf_counter()
    run_steps(m, bts, 300)
    torch.cuda.synchronize()
    print("steady state: %.0f" % (B * 300 / (time.perf_counter() - t0)), flush=True)
    m.close()
