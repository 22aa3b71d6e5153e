#!/bin/bash
# more hardware queues (GPU_MAX_HW_QUEUES) and more contexts than four?
set -u
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for cfg in "4 4 6" "4 8 6" "5 8 8" "6 8 8" "3 4 6" "2 4 6"; do
set -- $cfg
UFD_NUM_CTX=$1 GPU_MAX_HW_QUEUES=$2 timeout -k 10 200 python3 bench.py --host-only --steps 300 --warmup 20 --depth $3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host']; print('ctx $1 hwq $2 depth $3 round $r: value', d['value'], 'steady', d['steady_state_fps'], 'span share', h['gpu_span_share'])"
done; done
