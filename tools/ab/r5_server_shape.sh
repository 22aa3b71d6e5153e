#!/bin/bash
# The reference server's own operating point as measured lines: 1280x720 slots (router.rs:66-67), 4:2:2 MJPG
# without DHT segments (what cam_sender captures, sensors.rs:18-68) -> UltraFace-320 (inferer.rs:23) -> rectangles + labels +
# JPEG q95 re-encode (inferer.rs:38-46).  (a) one frame at a time through ufd_submit_annotate_batch / ufd_wait,
# (b) eight such cameras through ONE ufd_sched on one GPU.
set -u
name=${1:-r5s}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$name
mkdir -p $out
common="--variant 320 --src 1280x720 --subsampling 4:2:2 --no-dht --annotate"
timeout -k 10 400 python3 bench.py $common --batch 1 --depth 1 --steps 400 --warmup 20 --pool 64 > $out/bench_server_default_one_at_a_time.json 2>> $out/bench.err; echo "one at a time rc=$?"
timeout -k 10 400 python3 bench.py $common --one-process --gpus 1 --streams 8 --batch 8 --depth 6 --steps 60 --warmup 6 --pool 64 > $out/bench_server_default_8_cameras_one_sched.json 2>> $out/bench.err; echo "8 cameras rc=$?"
timeout -k 10 400 python3 bench.py $common --batch 8 --depth 6 --steps 200 --warmup 20 --pool 64 > $out/bench_server_default_batch8.json 2>> $out/bench.err; echo "batch 8 rc=$?"
for f in $out/bench_server_default_*.json; do python3 - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    print(sys.argv[1].split('/')[-1], d['value'], 'steady', d.get('steady_state_fps'), 'roof', r.get('kernel'), r.get('bound'), r.get('frac'), 'lat', (d.get('latency_ms_batch1') or {}).get('median'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'ver', d.get('verified'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
tail -5 $out/bench.err
