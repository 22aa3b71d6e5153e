#!/usr/bin/env python3
"""Launch shapes of every kernel instance from a rocprofv3 kernel trace (the csv tools/kernel_times.sh leaves): workgroups,
threads, LDS, registers AS DISPATCHED, and from them how the launch sits on the GPU's 256 CUs --
  resident = workgroups one CU holds at once = min(LDS: 160 KB / block's LDS; registers: floor(512 / registers per lane,
             rounded up to 8) waves per SIMD x 4 SIMDs / waves per block; 2048 threads / block threads; 32 blocks)
  slots    = 256 x resident;  rounds = workgroups / slots;  last-round fill = share of the slots the last round uses
-- the column of DESIGN.md's kernel table that says whether a launch is one round, a ragged 1.25 rounds or leaves CUs empty
(VERDICT r5 #2).  A kernel instance launched with several grids per batch (one per layer) gets one entry per grid.
Usage: launch_shapes.py <kernel_trace.csv> <out.json>"""
import csv
import json
import re
import sys

CUS, LDS_CU, REGS, THREADS_CU, MAX_BLOCKS = 256, 160 * 1024, 512, 2048, 32


def occupancy(wg, lds, vgpr, agpr):
    waves = (wg + 63) // 64
    regs = max(8, -(-(vgpr + agpr) // 8) * 8)  # unified register file: VGPRs + accumulation registers, allocated in eights
    per_simd = min(8, REGS // regs)
    by_regs = per_simd * 4 // waves if waves <= per_simd * 4 else 0
    by_lds = LDS_CU // lds if lds else MAX_BLOCKS
    return max(1, min(by_regs if by_regs else 1, by_lds, THREADS_CU // wg, MAX_BLOCKS))


def main():
    shapes = {}
    for r in csv.DictReader(open(sys.argv[1])):
        name = re.sub(r"^void |ufd::\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"])
        if name.startswith("__amd"):
            continue
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        key = (grid // wg, wg, int(r["LDS_Block_Size"]), int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["Scratch_Size"]))
        shapes.setdefault(name, {}).setdefault(key, 0)
        shapes[name][key] += 1
    out = {}
    for name, ks in shapes.items():
        rows = []
        for (blocks, wg, lds, vgpr, agpr, scratch), n in sorted(ks.items(), key=lambda kv: -kv[1]):
            res = occupancy(wg, lds, vgpr, agpr)
            slots = CUS * res
            full, rem = divmod(blocks, slots)
            rows.append({"launches": n, "workgroups": blocks, "threads": wg, "lds_bytes": lds, "vgpr": vgpr, "agpr": agpr, "scratch": scratch,
                         "resident_per_cu": res, "slots": slots, "rounds": round(blocks / slots, 3),
                         "last_round_fill": round((rem or (slots if full else 0)) / slots, 3)})
        out[name] = rows
    json.dump({"instances": out, "what": __doc__.split("Usage")[0].strip()}, open(sys.argv[2], "w"), indent=1)
    for name, rows in sorted(out.items()):
        for r in rows:
            print("%-34s x%-4d %5d workgroups of %4d | LDS %6d B, %3d + %3d regs -> %2d per CU, %5d slots: %.2f rounds, last round %3.0f %% full" % (
                name[:34], r["launches"], r["workgroups"], r["threads"], r["lds_bytes"], r["vgpr"], r["agpr"], r["resident_per_cu"], r["slots"],
                r["rounds"], 100 * r["last_round_fill"]))


if __name__ == "__main__":
    main()
