"""Soak: the bench's submit / wait loop for a given time with the host and device memory of the process sampled -- leaks in
the per-batch paths (event pools, table-set caches, staging) show as growth.  Every 50th batch carries a few corrupt frames
and a stream with other Huffman tables, so the error and cache paths run too.
Usage: python tools/soak.py [seconds] [clean] [spin] [staged] [hostentropy] [annot] [small]
("small": batches of two frames -- the small-batch forms of the host plan: staging block in and results out by kernels)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infercam_onnx_amd import nn, synth

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
opts = sys.argv[2:]
B = 2 if "small" in sys.argv[2:] else 32
weights, pri = synth.synthetic_weights(), synth.gen_priors(640, 480)
m = nn.UltrafaceModel(nn.UltrafaceVariant.W640H480, 0.5, 0.5, weights=weights, priors=pri, max_batch=B, max_src=(640, 480), det_cap=256,
                      extra_flags=nn.UFD_FLAG_SPIN_WAIT if "spin" in opts else 0, host_entropy="hostentropy" in opts)
jp = synth.synth_jpeg_pool(0, 192, 640, 480, quality=90, subsampling="4:2:0")
odd = [synth.encode_jpeg(synth.synth_frame(7, i, 640, 480), optimize=True, quality=60 + i) for i in range(8)]  # per-frame optimised tables
batches = [m._prep_batch(jp[i * B:(i + 1) * B]) for i in range(6)]
bad = list(jp[:B])
for k in (3, 11, 19):
    if k < B:
        bad[k] = bad[k][: len(bad[k]) // 2]
for k in range(8):
    if 20 + k < B:
        bad[20 + k] = odd[k]
if B < 32:
    bad[0], bad[B - 1] = jp[0][: len(jp[0]) // 2], odd[0]
batches.append(m._prep_batch(bad))
if "staged" in opts:
    batches = [m.stage_jpeg_batch(jp[i * B:(i + 1) * B]) for i in range(6)] + [batches[6]]
if "annot" in opts:  # the whole Inferer::run iteration: + rectangles, labels, re-encode, streams copied back
    batches = [m.prep_annotate_batch(jp[i * B:(i + 1) * B], (1280, 720), out_bytes_per_frame=640 * 480) for i in range(6)]
submit = (lambda b: m.submit_annotate_batch(b) if getattr(b, "annot", None) is not None else
          (m.submit_staged(b) if getattr(b, "staged", None) else m.submit_jpeg_batch(b)))


def rss_mb():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmRSS"):
            return int(ln.split()[1]) / 1024.0


def gpu_used_mb():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 1e6


inflight, n, t0, last = [], 0, time.time(), time.time()
first = None
samples = []
interval = min(10, secs / 16)


def late_jump():
    """The largest step of the second half so far sits in its last two intervals: nothing measured behind it yet."""
    if len(samples) < 8:
        return False
    half_ = samples[len(samples) // 2:]
    st = [b_[1] - a_[1] for a_, b_ in zip(half_, half_[1:])]
    i_ = max(range(len(st)), key=lambda k_: st[k_])
    return st[i_] > 8.0 and i_ >= len(st) - 2


deadline, extensions = secs, 0
while True:
    if time.time() - t0 >= deadline:
        # a one-off jump (the runtime enlarging a pool: +190 MB in one interval) at the very end looks like the start of a leak:
        # run four more intervals, at most twice, so that there IS something measured behind it
        if extensions < 2 and late_jump():
            extensions += 1
            deadline += 4 * interval
            print("late step in the resident set: running %.1f s longer (extension %d)" % (4 * interval, extensions), flush=True)
        else:
            break
    if len(inflight) >= 6:
        m.wait(inflight.pop(0), collect=False)
    k = 6 if n % 50 == 49 and not any(b is batches[6] for b in [m._pending[t] for t in inflight]) else n % 6
    if any(m._pending[t] is batches[k] for t in inflight):
        m.wait(inflight.pop(0), collect=False)
        continue
    if "clean" in opts and k == 6:
        k = 0
        if any(m._pending[t] is batches[k] for t in inflight):
            m.wait(inflight.pop(0), collect=False)
            continue
    inflight.append(submit(batches[k]))
    n += 1
    if time.time() - last > interval:
        last = time.time()
        rec = (n, rss_mb(), gpu_used_mb())
        first = first or rec
        samples.append(rec)
        print("batches %7d  rss %8.1f MB  gpu %9.1f MB" % rec, flush=True)
for t in inflight:
    m.wait(t, collect=False)
end = (n, rss_mb(), gpu_used_mb())
print("soak: %d batches in %.0f s; rss %+0.1f MB, gpu %+0.1f MB since the first sample" % (n, time.time() - t0, end[1] - first[1], end[2] - first[2]))
hs = m.host_stats()
print("host:", hs["per_batch_us"], hs["gpu_span_share"])
m.close()
# A leak grows with the batch count to the end of the run; one-off growth (late first touches of pinned pages, allocator arenas
# of threads that start late, the runtime enlarging a pool once: +190 MB in ONE sampling interval -- between the first two
# samples of one run, between the last two of another -- flat before and after) does not.  So the slope is taken over the
# SECOND HALF of the run (round 4's runtime leak was 2.1 KB per batch, in every interval: 45 MB in 12 s), and the largest
# single step between two samples is taken out ONLY when it is an isolated one: the run is long enough that one interval
# cannot hold all of a leak's growth (eight or more second-half samples) and the steps on both sides of it are flat (below the
# per-batch bar).  A leak that arrives in bursts -- a pool enlarged every N batches -- has more than one such step and stays in;
# a late jump with nothing measured behind it stays in too.  The excluded step is printed and reported.
samples.append(end)
half = samples[len(samples) // 2:] if len(samples) >= 4 else [first, end]
steps = [b[1] - a[1] for a, b in zip(half, half[1:])]
batches_of = [b[0] - a[0] for a, b in zip(half, half[1:])]
BAR = 300.0  # bytes per batch


def flat(i):
    return 0 <= i < len(steps) and steps[i] * 1048576.0 / max(batches_of[i], 1) < BAR


excluded = None
if len(steps) >= 7:  # (eight or more second-half samples)
    i = max(range(len(steps)), key=lambda k: steps[k])
    if steps[i] > 0 and flat(i - 1) and flat(i + 1):
        excluded = (half[i][0], half[i + 1][0], steps[i])
growth = sum(steps) - (excluded[2] if excluded else 0.0)
per_batch = growth * 1048576.0 / max(half[-1][0] - half[0][0], 1)
if excluded:
    print("WARNING: one isolated step of %+.1f MB between batches %d and %d left out of the slope (flat before and after)" % (excluded[2], excluded[0], excluded[1]))
print("host memory per batch over the second half (from batch %d on, %d steps%s): %.0f bytes"
      % (half[0][0], len(steps), ", one excluded" if excluded else "", per_batch))
ok = per_batch < BAR and abs(end[2] - first[2]) < 64
import json

print("soak-json " + json.dumps({"batches": n, "per_batch_bytes": round(per_batch), "second_half_growth_mb": round(growth, 1),
                                 "excluded_step": excluded, "gpu_growth_mb": round(end[2] - first[2], 1), "ok": bool(ok)}))
print("ok" if ok else "GROWTH")
sys.exit(0 if ok else 1)
