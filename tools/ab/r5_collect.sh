#!/bin/bash
# Full evidence collection of round 5 on the GPU box: tools/collect_profiles.sh (C3 bench line, rocprofv3 kernel stats, PMC
# traffic), the driver-flag line, the secondary configs C2 / C5, the 4:2:2 / DRI / DHT-less C3 lines, the reference server's
# shape, alone kernel times, the SQ-counter table, the fabric read requests, the host sweep.
set -u
name=${1:-r5z}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$name
mkdir -p $out
bash tools/collect_profiles.sh $name
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_flags.json 2>> $out/bench.err; echo "driver-flag line rc=$?"
timeout -k 10 300 python3 bench.py --variant 320 --batch 1 --depth 1 --steps 300 --warmup 20 > $out/bench_c2_320_batch1.json 2>> $out/bench.err; echo "C2 rc=$?"
timeout -k 10 300 python3 bench.py --src 1280x720 --batch 16 --steps 100 --warmup 10 > $out/bench_c5_1280x720_batch16.json 2>> $out/bench.err; echo "C5 rc=$?"
timeout -k 10 300 python3 bench.py --subsampling 4:2:2 --steps 80 --warmup 8 --no-cpu-baseline > $out/bench_c3_422.json 2>> $out/bench.err; echo "4:2:2 rc=$?"
timeout -k 10 300 python3 bench.py --subsampling 4:2:2 --restart-rows 1 --steps 80 --warmup 8 --no-cpu-baseline > $out/bench_c3_422_dri1.json 2>> $out/bench.err; echo "4:2:2 DRI rc=$?"
timeout -k 10 300 python3 bench.py --subsampling 4:2:2 --no-dht --steps 80 --warmup 8 --no-cpu-baseline > $out/bench_c3_422_nodht.json 2>> $out/bench.err; echo "4:2:2 no DHT rc=$?"
common="--variant 320 --src 1280x720 --subsampling 4:2:2 --no-dht --annotate"
timeout -k 10 400 python3 bench.py $common --batch 1 --depth 1 --steps 400 --warmup 20 --pool 64 > $out/bench_server_default_one_at_a_time.json 2>> $out/bench.err; echo "server, one at a time rc=$?"
timeout -k 10 400 python3 bench.py $common --one-process --gpus 1 --streams 8 --batch 8 --depth 6 --steps 60 --warmup 6 --pool 64 > $out/bench_server_default_8_cameras_one_sched.json 2>> $out/bench.err; echo "server, 8 cameras rc=$?"
bash tools/kernel_times.sh > $out/kernel_times_alone.txt 2>&1
bash tools/pmc_kernel.sh k_ > $out/sq_counters.txt 2>&1
bash tools/pmc_ea.sh pmc_ea > $out/ea_reads.txt 2>&1
timeout -k 10 500 python3 tools/host_scaling.py $out/host_scaling.json --cpus 0,16,8,4,2 --steps 300 > $out/host_scaling.log 2>&1
for f in $out/bench*.json; do python3 - $f <<'PY'
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
