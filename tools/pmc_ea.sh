#!/bin/bash
# L2 -> fabric read requests of every kernel (alone, bench.py --depth 1): all requests, those addressed to local memory
# ("DRAM": Infinity Cache or HBM behind it -- rocprofv3 exposes no fabric-cache hit counter on this box), and the requests-in-
# flight integral, whose quotient with the request count is the average fabric read latency of the kernel in L2 clocks: a
# kernel whose misses are served by the Infinity Cache shows a lower latency than one that streams from HBM.
# Usage on the box: tools/pmc_ea.sh [out dir under gpurun_out]
set -u
name=${1:-pmc_ea}
out=$GRAFT_REPO_ROOT/gpurun_out/$name
rm -rf $out; mkdir -p $out
(cd $GRAFT_REPO_ROOT && python3 -c "import bench; print(bench.kernel_source_sha())" > $out/source_sha.txt)
cd /tmp && export TMPDIR=/tmp
i=0
for g in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_BUBBLE_sum"; do
  timeout 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out/g$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --depth 1 --no-cpu-baseline --no-extras --pool 64 > /dev/null 2> $out/g$i.err || echo "group $i failed: $(tail -2 $out/g$i.err)"
  i=$((i+1))
done
python3 - $out <<'PY'
import collections, csv, glob, json, re, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void |ufd::\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = {}
print("%-34s %8s %10s %10s %8s %10s %8s" % ("kernel", "us", "rd MB", "dram MB", "dram%", "lat clk", "l2hit%"))
for k, d in sorted(acc.items(), key=lambda kv: -sum(dur[kv[0]])):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    rd, rd32 = m.get("TCC_EA0_RDREQ_sum", 0), m.get("TCC_EA0_RDREQ_32B_sum", 0)
    mb = ((rd - rd32) * 128 + rd32 * 32) / 1e6   # (gfx950: a request is a 128-byte line -- the x2 of MI355X_MICROARCH.md's FETCH_SIZE calibration; equals FETCH_SIZE x 2 of pmc_traffic.json)
    dram = m.get("TCC_EA0_RDREQ_DRAM_sum", 0)
    lat = m.get("TCC_EA0_RDREQ_LEVEL_sum", 0) / rd if rd else 0
    hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
    us = sum(dur[k]) / len(dur[k]) / 1e3
    rows[k] = dict(us=round(us, 1), ea_read_bytes=round(mb * 1e6), dram_bytes=round(dram * 128), ea_read_requests=rd, ea_read_requests_dram=dram, ea_read_latency_clk=round(lat, 1),
                   l2_hit_share=round(hit / (hit + miss), 4) if hit + miss else None, raw=m)
    if us > 4:
        print("%-34s %8.1f %10.1f %10.1f %8.1f %10.1f %8.1f" % (k[:34], us, mb, dram * 128 / 1e6, 100 * dram / rd if rd else 0, lat, 100 * hit / (hit + miss) if hit + miss else 0))
json.dump({"note": "rocprofv3 --pmc TCC_EA0_RDREQ* per kernel, kernels alone (bench.py --depth 1).  dram_bytes = read requests ADDRESSED to local memory x 128 B: "
                   "the Infinity Cache sits behind that address decode, so hits in it are not separable (rocprofv3 --list-avail on this box has no MALL / data-fabric "
                   "event); ea_read_latency_clk = TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ, the average fabric read latency in L2 clocks", "instances": rows},
          open(out + "/ea_reads.json", "w"), indent=1, sort_keys=True)
PY
