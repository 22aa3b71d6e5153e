#!/bin/bash
# same-box A/B of a bench.py flag: steady-state frames/s, alternating, 3 rounds of 300 steps each
set -u
flag=${1:---no-rfb-tail}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for r in 1 2 3; do
  for v in base flag; do
    extra=""; [ $v = flag ] && extra=$flag
    timeout -k 10 200 python3 bench.py --host-only --steps 300 --warmup 20 $extra 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v round $r: value', d['value'], 'steady', d['steady_state_fps'])"
  done
done
bash tools/kernel_times.sh 2>&1 | head -12
