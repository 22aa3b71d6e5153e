#!/bin/bash
# bench.py --depth 6 against --depth 8, quiet and with 16 spinning processes beside it
set -u
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 100 python3 bench.py --host-only --steps 200 --warmup 10 --depth $1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host']; print('$2 depth $1: value', d['value'], 'steady', d['steady_state_fps'], 'span share', h['gpu_span_share'], 'gap', h['gpu_idle_gap_us_per_batch'])"; }
for r in 1 2; do run 6 quiet; run 8 quiet; done
pids=""
for i in $(seq 1 16); do python3 -c "
import time
t=time.time()
while time.time()-t < 120: pass" & pids="$pids $!"; done
sleep 1
for r in 1 2; do run 6 hogs16; run 8 hogs16; done
kill $pids 2>/dev/null; wait 2>/dev/null
