#!/bin/bash
# re-collects the bench LINES of a round (kernels unchanged: PMC / SQ / kernel stats of the same name stay) + the host contention A/B
set -u
name=${1:-r4z}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$name
mkdir -p $out
python3 -c "import bench; print(bench.kernel_source_sha())" > $out/source_sha_lines.txt
timeout -k 10 600 python3 bench.py --steps 80 --warmup 8 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_flags.json 2>> $out/bench.err; echo "driver-flag line rc=$?"
timeout -k 10 300 python3 bench.py --variant 320 --batch 1 --depth 1 --steps 300 --warmup 20 > $out/bench_c2_320_batch1.json 2>> $out/bench.err; echo "C2 rc=$?"
timeout -k 10 300 python3 bench.py --src 1280x720 --batch 16 --steps 100 --warmup 10 > $out/bench_c5_1280x720_batch16.json 2>> $out/bench.err; echo "C5 rc=$?"
timeout -k 10 500 python3 tools/host_scaling.py $out/host_scaling.json --cpus 0,16,8,4,2 --steps 300 > $out/host_scaling.log 2>&1
bash tools/ab/r4_ab_noisy.sh UFD_PLAN_PARALLEL=1 16 > $out/host_contention_16hogs.txt 2>&1
bash tools/ab/r4_ab_noisy.sh UFD_PLAN_PARALLEL=1 32 > $out/host_contention_32hogs.txt 2>&1
for f in bench.json bench_driver_flags.json bench_c2_320_batch1.json bench_c5_1280x720_batch16.json; do python3 - $out/$f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}; h=d.get('host',{})
print(sys.argv[1].split('/')[-1], d['value'], d.get('steady_state_fps'), 'roof', r.get('kernel'), r.get('frac'), r.get('traffic_ratio'), r.get('mfma_busy'), 'lat', (d.get('latency_ms_batch1') or {}).get('median'), 'gaps', h.get('gpu_idle_gap_us_per_batch'))
PY
done
tail -6 $out/host_scaling.log; cat $out/host_contention_16hogs.txt $out/host_contention_32hogs.txt
