#!/usr/bin/env python3
"""Per-launch timeline of one frame at a time (bench.py --batch 1 --depth 1) from a rocprofv3 kernel trace: for a batch in the
middle of the run, every kernel with its start offset, duration and the idle gap in front of it -- where the 0.5 ms of the
reference's operating point (one stream, one frame at a time: inferer.rs:23,29-50) go.
Usage: batch1_timeline.py <kernel_trace.csv> [launches per batch, default: detected from k_huff_unstuff]"""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"^void |ufd::\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_huff_unstuff")]
if len(starts) < 8:
    raise SystemExit("fewer than 8 batches in the trace")
# every batch from its k_huff_unstuff to the launch before the next one
spans, busys = [], []
for a, b in zip(starts[3:-1], starts[4:]):
    seg = rows[a:b]
    spans.append((seg[-1][1] - seg[0][0]) / 1e3)
    busys.append(sum(e - s for s, e, _ in seg) / 1e3)
mid = starts[len(starts) // 2]
seg = rows[mid:starts[len(starts) // 2 + 1]]
t0 = seg[0][0]
print("%-36s %9s %8s %8s" % ("kernel", "start us", "dur us", "gap us"))
prev_end = t0
for s, e, n in seg:
    print("%-36s %9.1f %8.1f %8.1f" % (n[:36], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
spans.sort(), busys.sort()
print("launches per batch %d; first kernel start -> last kernel end: median %.1f us, kernels busy %.1f us, gaps %.1f us" % (
    len(seg), spans[len(spans) // 2], busys[len(busys) // 2], spans[len(spans) // 2] - busys[len(busys) // 2]))
