#!/bin/bash
# the four-context form as built: GPU suite, smoke, bench lines (incl. annotate and latency), soak
set -u
name=${1:-r4ctx}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$name
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest -m gpu rc=$rc"; tail -4 $out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python3 bench.py --steps 80 --warmup 8 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_flags.json 2>> $out/bench.err; echo "driver-flag line rc=$?"
for f in bench.json bench_driver_flags.json; do python3 - $out/$f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}; h=d.get('host',{})
print(sys.argv[1].split('/')[-1], d['value'], d.get('steady_state_fps'), 'roof', r.get('kernel'), r.get('frac'), 'lat', (d.get('latency_ms_batch1') or {}), 'annot', (d.get('annotate') or {}).get('fps'), 'gaps', h.get('gpu_idle_gap_us_per_batch'), 'hbm', d['config'].get('hbm_resident_fps'))
PY
done
timeout -k 10 200 python3 tools/soak.py 30 clean | tail -4
timeout -k 10 200 python3 tools/soak.py 20 annot | tail -3
