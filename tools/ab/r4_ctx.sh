#!/bin/bash
# contexts x staging form: does a fourth context pay once the H2D goes through a kernel on the context's stream and NO copy stream exists?
set -u
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for cfg in "3 262144 6 0" "4 100000000 6 1" "4 100000000 8 1" "3 100000000 6 1"; do
set -- $cfg
pre=""; [ $4 = 1 ] && pre="env UFD_NO_COPY_STREAM=1"
UFD_NUM_CTX=$1 UFD_STAGE_KERNEL_MAX=$2 $pre timeout -k 10 200 python3 bench.py --host-only --steps 300 --warmup 20 --depth $3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host']; print('ctx $1 stage_max $2 depth $3 nocopystream $4 round $r: value', d['value'], 'steady', d['steady_state_fps'], 'span share', h['gpu_span_share'], 'gap', h['gpu_idle_gap_us_per_batch'])"
done; done
