#!/bin/bash
# same-box A/B of an environment knob of the library: steady-state frames/s + the host object, alternating, 3 rounds
set -u
knob=${1:-UFD_PLAN_PARALLEL=1}
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in base knob; do
    pre=""; [ $v = knob ] && pre="env $knob"
    $pre timeout -k 10 200 python3 bench.py --host-only --steps 300 --warmup 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host']; print('$v round $r: value', d['value'], 'steady', d['steady_state_fps'], h['per_batch_us'], 'span share', h['gpu_span_share'], 'gap', h['gpu_idle_gap_us_per_batch'])"
  done
done
