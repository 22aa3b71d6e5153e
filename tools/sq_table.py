#!/usr/bin/env python3
"""SQ-counter table of every kernel from the rocprofv3 --pmc passes tools/pmc_kernel.sh collected (one pass per counter
group, kernels alone on the GPU: bench.py --depth 1).  Prints the table and writes <dir>/sq_counters.json:
  instances[kernel<template args>] = {us, mfma_busy, valu_busy, waves_per_simd, wait_share, lds_conflict_share, icache_miss}
    us              mean dispatch duration in the same passes (End - Start of the kernel trace: equal to the alone times of
                    kernel_times_alone.txt; GRBM_GUI_ACTIVE is NOT used -- under counter collection it runs on through the
                    counter read-out, 2.4x the dispatch for the dominant kernel, and round 2's table divided by it)
    SIMD-cycles     = us x 2400 MHz (the MI355X's peak engine clock, MI355X_MICROARCH.md) x 1024 SIMDs
    mfma_busy       SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles: share of SIMD-cycles the matrix pipe was busy (the counter equals
                    SQ_INSTS_MFMA x 64 cycles for v_mfma_f32_32x32x2_f32 exactly)
    valu_busy       4 x (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SIMD-cycles: vector instructions other than MFMA, 4 cycles per wave64 op
    waves_per_simd  4 x SQ_WAVE_CYCLES / SIMD-cycles (the counter ticks in quad-cycles): resident waves per SIMD, averaged
                    over the kernel's time
    wait_share      SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: share of a wave's resident cycles spent waiting on an instruction's
                    dependency (memory, LDS, export)
    lds_conflict_share  SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS
  raw[kernel][counter] = mean over the kernel's dispatches.
Usage: sq_table.py <dir with g*/...counter_collection.csv> [name substring]"""
import collections
import csv
import glob
import json
import re
import sys

out = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(dict)  # kernel -> {(file, dispatch id): ns}
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            key = re.sub(r"^void |ufd::\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"])
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            durs[key][(f, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
means = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
CLOCK_MHZ = 2400.0
inst = {}
print("%-34s %8s %7s %7s %7s %7s %8s %8s" % ("kernel", "us", "mfma%", "valu%", "wait%", "w/simd", "ic_miss%", "ldsconf%"))
mean_us = {k: sum(v.values()) / len(v) / 1e3 for k, v in durs.items() if v}
for k, m in sorted(means.items(), key=lambda kv: -mean_us.get(kv[0], 0) * len(durs[kv[0]])):
    us = mean_us.get(k, 0)
    if us <= 0:
        continue
    simd_cycles = us * CLOCK_MHZ * 1024.0
    e = {"us": round(us, 1), "dispatches": len(durs[k]),
         "mfma_busy": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cycles, 4),
         "valu_busy": round(4.0 * max(m.get("SQ_INSTS_VALU", 0) - m.get("SQ_INSTS_MFMA", 0), 0) / simd_cycles, 4),
         "waves_per_simd": round(4.0 * m.get("SQ_WAVE_CYCLES", 0) / simd_cycles, 3),
         "wait_share": round(m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 4),
         "lds_conflict_share": round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_ACTIVE_INST_LDS", 1), 1), 4),
         "icache_miss": round(m.get("SQC_ICACHE_MISSES", 0) / max(m.get("SQC_ICACHE_REQ", 1), 1), 4)}
    inst[k] = e
    print("%-34s %8.1f %7.1f %7.1f %7.1f %7.2f %8.2f %8.2f" % (k[:34], e["us"], 100 * e["mfma_busy"], 100 * e["valu_busy"], 100 * e["wait_share"],
                                                           e["waves_per_simd"], 100 * e["icache_miss"], 100 * e["lds_conflict_share"]))
json.dump({"note": "rocprofv3 --pmc, one pass per counter group, kernels alone (bench.py --depth 1); definitions in tools/sq_table.py",
           "instances": inst, "raw": means}, open(out + "/sq_counters.json", "w"), indent=1, sort_keys=True)
