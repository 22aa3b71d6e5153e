#!/bin/bash
# Full evidence collection of a round on the GPU box: tools/collect_profiles.sh (C3 bench line, rocprofv3 kernel stats, PMC
# traffic), the driver-flag line, the secondary configs C2 / C5, alone kernel times, the SQ-counter table, the host sweep.
set -u
name=${1:-r4z}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$name
mkdir -p $out
bash tools/collect_profiles.sh $name
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_flags.json 2>> $out/bench.err; echo "driver-flag line rc=$?"
timeout -k 10 300 python3 bench.py --variant 320 --batch 1 --depth 1 --steps 300 --warmup 20 > $out/bench_c2_320_batch1.json 2>> $out/bench.err; echo "C2 rc=$?"
timeout -k 10 300 python3 bench.py --src 1280x720 --batch 16 --steps 100 --warmup 10 > $out/bench_c5_1280x720_batch16.json 2>> $out/bench.err; echo "C5 rc=$?"
bash tools/kernel_times.sh > $out/kernel_times_alone.txt 2>&1
bash tools/pmc_kernel.sh k_ > $out/sq_counters.txt 2>&1
timeout -k 10 500 python3 tools/host_scaling.py $out/host_scaling.json --cpus 0,16,8,4,2 --steps 300 > $out/host_scaling.log 2>&1
for f in bench.json bench_driver_flags.json bench_c2_320_batch1.json bench_c5_1280x720_batch16.json; do python3 - $out/$f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    print(sys.argv[1].split('/')[-1], d['value'], d.get('steady_state_fps'), 'roof', r.get('kernel'), r.get('frac'), r.get('traffic_ratio'), r.get('mfma_busy'), 'lat', (d.get('latency_ms_batch1') or {}).get('median'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
tail -8 $out/host_scaling.log
