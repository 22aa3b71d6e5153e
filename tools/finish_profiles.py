#!/usr/bin/env python3
"""After tools/collect_profiles.sh <name> has run on the GPU box and gpurun merged gpurun_out/<name>/ back:
copies the summaries the judge reads into profiles/<name>/ and refreshes profiles/pmc_traffic_latest.json, stamped
with the commit and the hash of the kernel sources it was measured on (bench.py reports roofline.traffic only while
that hash matches).  Usage: python tools/finish_profiles.py <name>"""
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", name)
dst = os.path.join(ROOT, "profiles", name)
os.makedirs(dst, exist_ok=True)


def first(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[0]


for f in ("bench.json", "bench_under_rocprof.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
shutil.copy(first("stats/**/*kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
fetch, write = first("pmc_fetch/**/*counter_collection.csv"), first("pmc_write/**/*counter_collection.csv")
shutil.copy(fetch, os.path.join(dst, "pmc_fetch_counter_collection.csv"))
shutil.copy(write, os.path.join(dst, "pmc_write_counter_collection.csv"))
out = os.path.join(dst, "pmc_traffic.json")
print(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), fetch, write, out]).decode())
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the same hash bench.py checks: code only, comments stripped)

d = json.load(open(out))
d["kernel_source_sha"] = bench.kernel_source_sha(ROOT)
d["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
d["profile"] = name
json.dump(d, open(out, "w"), indent=1)
json.dump(d, open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json"), "w"), indent=1)
print("profiles/%s ready; pmc_traffic_latest.json stamped %s @ %s" % (name, d["kernel_source_sha"], d["commit"]))
