#!/usr/bin/env python3
"""Markdown table of DESIGN.md section 4 from a round's filed profiles: per kernel instance the launches per batch, the
alone time (rocprofv3 kernel trace, one batch in flight), the loaded time (bench.py's fully sampled pass, six batches in
flight), the SQ-counter shares and the PMC traffic.  With --update the table replaces the region between the
KERNEL_TABLE markers of DESIGN.md, so that the measured columns are never typed by hand.
Usage: python tools/design_table.py profiles/r4z [--update]"""
import json
import os
import re
import sys

d = sys.argv[1]
alone = {}
for ln in open(os.path.join(d, "kernel_times_alone.txt")):
    m = re.match(r"(?:void )?(\S.*?)\s+(\d+)\s+([\d.]+) us\s+([\d.]+)%", ln)
    if m:
        alone[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
batches = alone.get("k_stem_planes_mfma", (34, 0))[0]
sq = json.load(open(os.path.join(d, "sq_counters.json"))).get("instances", {}) if os.path.exists(os.path.join(d, "sq_counters.json")) else {}
pmc = json.load(open(os.path.join(d, "pmc_traffic.json"))).get("instances", {})
isa = json.load(open(os.path.join(d, "isa_mix.json"))).get("instances", {}) if os.path.exists(os.path.join(d, "isa_mix.json")) else {}
bench = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
loaded = bench.get("kernels_ms_per_step", {})
shapes = {}  # device kernel name -> {shape text: [layers]} from the bench line's launch_shapes (ufd_profile_shapes: as issued)


def shape_text(sh):
    """workgroups / slots = rounds (last-round fill); mean -> max workgroups per CU"""
    cus = max(1, sh["slots"] // max(sh["resident_per_cu"], 1))
    per_cu = sh["workgroups"] / cus
    mx = -(-sh["workgroups"] // cus)
    return "%d / %d = %.2f (%.0f %%); %.2f → %d per CU" % (sh["workgroups"], sh["slots"], sh["rounds"], 100 * sh["last_round_fill"], per_cu, mx)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402

loaded_by_dev = {}
for k, v in loaded.items():
    loaded_by_dev[B.device_name(k)] = loaded_by_dev.get(B.device_name(k), 0.0) + v * 1e3
for label, sh in bench.get("launch_shapes", {}).items():
    base, _, layers = label.partition(":")
    shapes.setdefault(B.device_name(base), {}).setdefault(shape_text(sh), []).append(layers.split(".")[0] if layers else "")
out_lines = []
_print = print


def print(*a):  # noqa: A001  (collect what is printed: --update writes it into DESIGN.md)
    out_lines.append(" ".join(str(x) for x in a))
    _print(*a)


print("Source: `%s/` (kernel_times_alone.txt, bench.json, sq_counters.json, pmc_traffic.json, isa_mix.json); regenerate with `python tools/design_table.py %s --update`.\n" % (d.rstrip("/"), d.rstrip("/")))
print("| kernel instance | launches / batch | alone µs / launch | loaded µs / batch | MFMA busy | VALU busy | waves / SIMD | waiting | LDS conflicts | traffic MB / launch (fetch + write) | hot loop: vector instr. / MFMA (not fp32 arithmetic) | workgroups / resident slots = rounds (last-round fill); workgroups per CU, mean → max |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
tot = 0.0
pct = lambda x: "%.0f %%" % (100 * x) if x is not None else "–"
for k, (calls, us) in sorted(alone.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if k.startswith("__amd"):
        continue
    per = calls / batches
    tot += per * us
    s = sq.get(k, {})
    p = pmc.get(k, {})
    tr = "%.0f + %.0f" % (p["fetch_bytes_per_launch"] / 1e6, p["write_bytes_per_launch"] / 1e6) if p else "–"
    ld = loaded_by_dev.get(k)
    h = isa.get(k, {}).get("hot_loop")
    mix = "–"
    if h and h.get("MFMA"):
        valu = sum(h.get(c, 0) for c in ("FMA", "mov", "cndmask", "maxmin", "cmp", "int", "cvt"))
        mix = "%.1f (%.1f)" % (valu / h["MFMA"], (valu - h.get("FMA", 0)) / h["MFMA"])
    sk = shapes.get(k)
    if sk is None:  # (a label without its template instance: "idct" for k_idct<true>)
        sk = next((v for n, v in shapes.items() if k.startswith(n + "<")), {})
    shp = "; ".join(t + (" [" + ", ".join(x for x in ls if x) + "]" if len(sk) > 1 and any(ls) else "") for t, ls in sk.items()) or "–"
    print("| `%s` | %.0f | %.1f | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
        k, per, us, "%.0f" % ld if ld else "–", pct(s.get("mfma_busy")), pct(s.get("valu_busy")),
        "%.2f" % s["waves_per_simd"] if s else "–", pct(s.get("wait_share")), pct(s.get("lds_conflict_share")), tr, mix, shp))
print("\nkernels alone per batch: %.0f µs; delivered: %.3f ms per batch (steady state %.0f frames/s)" % (
    tot, bench["host"]["ms_per_batch"] if "host" in bench else bench["ms_per_step"], bench.get("steady_state_fps", 0)))

# ---- the paragraph of DESIGN section 5 about the dominant kernel, from the SAME bench line (so that it cannot drift from the table)
r = bench.get("roofline", {})
dom_lines = []
if r.get("kernel"):
    al = r.get("alone", {})
    fl, by = r.get("algorithmic_flops_per_launch", 0), r.get("algorithmic_bytes_per_launch", 0)
    t = "  Dominant: `%s` (%.2f GFLOP, %.0f MB algorithmic per launch): **%.1f µs alone = %.2f of the fp32-MFMA roof; loaded %.0f µs = %.3f** " \
        "(the launch shares the GPU with three other contexts' kernels)" % (
            r["kernel"], fl / 1e9, by / 1e6, al.get("avg_launch_us", 0), al.get("mfma_frac", 0), r.get("avg_launch_us", 0), r.get("frac", 0))
    if r.get("traffic"):
        t += "; traffic %.0f MB = %.2f× the algorithmic bytes" % (r["traffic"] / 1e6, r.get("traffic_ratio") or 0)
        if r.get("dram_bytes"):
            t += " (%.0f MB addressed to local memory, fabric read latency %.0f L2 clocks)" % (r["dram_bytes"] / 1e6, r.get("ea_read_latency_clk") or 0)
    if r.get("mfma_busy") is not None:
        t += "; MFMA pipe busy %.0f %% and vector ALU %.0f %% of its SIMD-cycles alone at %.1f waves per SIMD" % (
            100 * r["mfma_busy"], 100 * r["valu_busy"], r["waves_per_simd"])
    t += ".  `whole_net_mfma_frac` = frames/s × 798.3 MFLOP ÷ 157.3 TFLOP/s = %.2f (timed region) / %.2f (steady state)." % (
        bench.get("whole_net_mfma_frac", 0), bench.get("steady_state_fps", 0) * 798315520 / 157.3e12)
    t += "  (Generated by `tools/design_table.py` from `%s/bench.json`.)" % d.rstrip("/")
    dom_lines = [t]
    _print("\n" + t)

if "--update" in sys.argv:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "DESIGN.md")
    text = open(path).read()
    a, b = "<!-- KERNEL_TABLE_BEGIN -->", "<!-- KERNEL_TABLE_END -->"
    i, j = text.index(a) + len(a), text.index(b)
    text = text[:i] + "\n" + "\n".join(out_lines) + "\n" + text[j:]
    a, b = "<!-- DOMINANT_BEGIN -->", "<!-- DOMINANT_END -->"
    if dom_lines and a in text:
        i, j = text.index(a) + len(a), text.index(b)
        text = text[:i] + "\n" + "\n".join(dom_lines) + "\n" + text[j:]
    open(path, "w").write(text)
    _print("DESIGN.md updated")
