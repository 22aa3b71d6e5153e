#!/bin/bash
# 4:2:2 / restart-marker / DHT-less streams as bench inputs: steady-state lines beside the 4:2:0 one, same box.
set -u
name=${1:-r5d}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$name
mkdir -p $out
run() {  # <file> <bench args...>
  f=$1; shift
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $out/$f.json 2>> $out/bench.err
  python3 - $out/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], d['value'], 'steady', d.get('steady_state_fps'), 'roof', r.get('kernel'), r.get('frac'), 'verified', (d.get('verified') or {}).get('max_abs_err'), (d.get('verified') or {}).get('unexplained_detections'))
PY
}
run bench_420
run bench_422 --subsampling 4:2:2
run bench_422_dri1 --subsampling 4:2:2 --restart-rows 1
run bench_422_nodht --subsampling 4:2:2 --no-dht
run bench_422_nodht_dri1 --subsampling 4:2:2 --no-dht --restart-rows 1
run bench_420_dri1 --restart-rows 1
