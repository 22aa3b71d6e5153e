#!/bin/bash
# kernel-composition choices re-checked under four contexts: dual launches, k_rfb_tail
set -u
cd $GRAFT_REPO_ROOT
run() { local label=$1 pre=$2; shift 2
  $pre timeout -k 10 200 python3 bench.py --host-only --steps 300 --warmup 20 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label:', 'value', d['value'], 'steady', d['steady_state_fps'])"
}
for r in 1 2 3; do
run "base r$r" ""
run "no dual r$r" "env UFD_NO_DUAL=1"
run "no rfb tail r$r" "" --no-rfb-tail
done
