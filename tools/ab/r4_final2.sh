#!/bin/bash
# round-end pass after the batch-1 work: the driver's steps (pytest -m gpu, smoke), the four bench lines, the batch-1 timelines
set -u
name=${1:-r4z2}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$name
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -4 $out/pytest_gpu.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 -c "import bench; print(bench.kernel_source_sha())" > $out/source_sha_lines.txt
timeout -k 10 600 python3 bench.py --steps 80 --warmup 8 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_flags.json 2>> $out/bench.err; echo "driver-flag line rc=$?"
timeout -k 10 300 python3 bench.py --variant 320 --batch 1 --depth 1 --steps 300 --warmup 20 > $out/bench_c2_320_batch1.json 2>> $out/bench.err; echo "C2 rc=$?"
timeout -k 10 300 python3 bench.py --src 1280x720 --batch 16 --steps 100 --warmup 10 > $out/bench_c5_1280x720_batch16.json 2>> $out/bench.err; echo "C5 rc=$?"
for f in bench.json bench_driver_flags.json bench_c2_320_batch1.json bench_c5_1280x720_batch16.json; do python3 - $out/$f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}; h=d.get('host',{})
print(sys.argv[1].split('/')[-1], d['value'], d.get('steady_state_fps'), 'roof', r.get('kernel'), r.get('frac'), r.get('traffic_ratio'), r.get('mfma_busy'), 'lat', (d.get('latency_ms_batch1') or {}).get('median'), 'gaps', h.get('gpu_idle_gap_us_per_batch'))
PY
done
bash tools/ab/r4_lat.sh $name | head -4
