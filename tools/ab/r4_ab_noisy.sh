#!/bin/bash
# the same A/B with the host's CPUs busy elsewhere: N spinning processes beside the bench (what eight ranks and other tenants do to one box)
set -u
knob=${1:-UFD_PLAN_PARALLEL=1}
hogs=${2:-16}
cd $GRAFT_REPO_ROOT
pids=""
for i in $(seq 1 $hogs); do python3 -c "
import time
t=time.time()
while time.time()-t < 170: pass" & pids="$pids $!"; done
sleep 1
for r in 1 2; do
  for v in base knob; do
    pre=""; [ $v = knob ] && pre="env $knob"
    $pre timeout -k 10 100 python3 bench.py --host-only --steps 200 --warmup 10 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host']; print('$hogs hogs, $v round $r: value', d['value'], 'steady', d['steady_state_fps'], h['per_batch_us'], 'span share', h['gpu_span_share'], 'gap', h['gpu_idle_gap_us_per_batch'])"
  done
done
kill $pids 2>/dev/null; wait 2>/dev/null
