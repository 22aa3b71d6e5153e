#!/bin/bash
# entropy chain (consecutive batches' decoders one behind the other): 20-step value, steady state, small batches
set -u
cd $GRAFT_REPO_ROOT
run() { local label=$1 pre=$2; shift 2
  $pre timeout -k 10 200 python3 bench.py --host-only "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host']; print('$label:', 'value', d['value'], 'steady', d['steady_state_fps'], 'gap', h['gpu_idle_gap_us_per_batch'])"
}
for r in 1 2 3; do
run "base 20 steps r$r" "" --steps 20 --warmup 5
run "chain 20 steps r$r" "env UFD_ENTROPY_CHAIN=1" --steps 20 --warmup 5
done
for r in 1 2; do
run "base 300 r$r" "" --steps 300 --warmup 20
run "chain 300 r$r" "env UFD_ENTROPY_CHAIN=1" --steps 300 --warmup 20
done
run "base 320 b32" "" --variant 320 --steps 300 --warmup 20
run "chain 320 b32" "env UFD_ENTROPY_CHAIN=1" --variant 320 --steps 300 --warmup 20
run "base b1 d6" "" --batch 1 --steps 600 --warmup 20
run "chain b1 d6" "env UFD_ENTROPY_CHAIN=1" --batch 1 --steps 600 --warmup 20
