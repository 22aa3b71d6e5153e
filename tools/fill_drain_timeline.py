#!/usr/bin/env python3
"""Per-batch device timeline of bench.py's timed region from a rocprofv3 kernel trace: for each of the last K batches the
queue (= context) it ran on, the start of its first kernel and the end of its last, relative to the first batch of the
region -- where the 20-step sample's time goes beyond K x the steady-state period (pipeline fill, lockstep start, drain).
Usage: fill_drain_timeline.py <kernel_trace.csv> [K] [SKIP]     (SKIP given: batches SKIP .. SKIP + K of the trace in start
order -- the timed region behind bench.py's warm-up steps; otherwise the last K batches of the trace)"""
import csv
import re
import sys

K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"^void |ufd::\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "0")))
rows.sort()
# a batch = the kernels of one queue from a k_huff_unstuff to the next one on that queue
per_q = {}
for s, e, n, q in rows:
    per_q.setdefault(q, []).append((s, e, n))
batches = []
for q, lst in per_q.items():
    cur = None
    for s, e, n in lst:
        if n.startswith("k_huff_unstuff"):
            if cur:
                batches.append(cur)
            cur = [q, s, e, 0.0]
        if cur:
            cur[2] = max(cur[2], e)
            cur[3] += (e - s) / 1e3
    if cur:
        batches.append(cur)
batches.sort(key=lambda b: b[1])
SKIP = int(sys.argv[3]) if len(sys.argv) > 3 else None
last = batches[-K:] if SKIP is None else batches[SKIP:SKIP + K]
t0 = last[0][1]
print("%-6s %-8s %10s %10s %10s %12s" % ("batch", "queue", "start us", "end us", "span us", "kernels us"))
for i, (q, s, e, busy) in enumerate(last):
    print("%-6d %-8s %10.1f %10.1f %10.1f %12.1f" % (i, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, busy))
ends = sorted(b[2] for b in last)
print("first start -> last end: %.1f us for %d batches; completions %.1f us apart in the middle third" % (
    (ends[-1] - t0) / 1e3, K, (ends[2 * K // 3] - ends[K // 3]) / 1e3 / (2 * K // 3 - K // 3)))
