#!/bin/bash
# four contexts / no copy stream / kernel staging against the shipped form on the other configurations
set -u
cd $GRAFT_REPO_ROOT
run() { # label, env prefix, bench args
  local label=$1 pre=$2; shift 2
  $pre timeout -k 10 200 python3 bench.py --host-only "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host']; print('$label:', 'value', d['value'], 'steady', d['steady_state_fps'], 'gap', h['gpu_idle_gap_us_per_batch'])"
}
NEW="env UFD_NUM_CTX=4 UFD_STAGE_KERNEL_MAX=100000000 UFD_NO_COPY_STREAM=1"
for r in 1 2; do
run "base 20 steps r$r" "" --steps 20 --warmup 5
run "new  20 steps r$r" "$NEW" --steps 20 --warmup 5
run "base C5 r$r" "" --src 1280x720 --batch 16 --steps 200 --warmup 10
run "new  C5 r$r" "$NEW" --src 1280x720 --batch 16 --steps 200 --warmup 10
run "base 320 b32 r$r" "" --variant 320 --steps 300 --warmup 20
run "new  320 b32 r$r" "$NEW" --variant 320 --steps 300 --warmup 20
run "base b8 r$r" "" --batch 8 --steps 600 --warmup 20
run "new  b8 r$r" "$NEW" --batch 8 --steps 600 --warmup 20
done
