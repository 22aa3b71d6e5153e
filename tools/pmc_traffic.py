"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, csv output) into
per-kernel HBM traffic per launch, applying the gfx950 corrections of MI355X_MICROARCH.md (HBM):
counters are in KiB; FETCH_SIZE reports exactly half of a wide coalesced streaming read (x2);
WRITE_SIZE is exact.  Usage: pmc_traffic.py <fetch counter_collection.csv> <write ...csv> [out.json]"""
import collections
import csv
import json
import re
import sys


def kname(s):
    m = re.search(r"(k_\w+)", s)
    return m.group(1) if m else s.split("(")[0]


def iname(s):
    """Kernel name with its template arguments, as kernel_stats.csv prints it: k_dwpw2_mfma<16, 1, true>."""
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", s)
    return m.group(1) if m else s.split("(")[0]


def load(path, counter, key=kname):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = key(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
fi = load(sys.argv[1], "FETCH_SIZE", iname)
wi = load(sys.argv[2], "WRITE_SIZE", iname)
inst = {}
for k in fi:
    n = fi[k][0]
    fe = 2.0 * fi[k][1] * 1024 / n
    wr = wi[k][1] * 1024 / max(wi[k][0], 1)
    inst[k] = {"launches": n, "fetch_bytes_per_launch": fe, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": fe + wr}
out = {}
print("%-22s %8s %14s %14s %14s" % ("kernel", "launches", "fetch MB (x2)", "write MB", "total MB/launch"))
for k in sorted(f, key=lambda k: -(f[k][1] * 2 + w[k][1])):
    n = f[k][0]
    fe = 2.0 * f[k][1] * 1024 / n
    wr = w[k][1] * 1024 / max(w[k][0], 1)
    out[k] = {"launches": n, "fetch_bytes_per_launch": fe, "write_bytes_per_launch": wr,
              "hbm_bytes_per_launch": fe + wr}
    print("%-22s %8d %14.2f %14.2f %14.2f" % (k, n, fe / 1e6, wr / 1e6, (fe + wr) / 1e6))
print()
for k in sorted(inst, key=lambda k: -inst[k]["hbm_bytes_per_launch"] * inst[k]["launches"]):
    v = inst[k]
    print("%-34s %6d fetch %8.1f MB  write %8.1f MB" % (k[:34], v["launches"], v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))
if len(sys.argv) > 3:
    json.dump({"note": "FETCH_SIZE x2 (gfx950 calibration), WRITE_SIZE exact, KiB units; separate --pmc passes",
               "kernels": out, "instances": inst}, open(sys.argv[3], "w"), indent=1)
