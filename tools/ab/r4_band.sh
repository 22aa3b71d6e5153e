#!/bin/bash
# band height of the m1->m2 chained kernel under the LOADED pipeline (alone: 5 rows is best, one round of resident waves)
set -u
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for band in 5 3 4 6 8 2; do
UFD_BAND_BIG=$band timeout -k 10 200 python3 bench.py --host-only --steps 300 --warmup 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('band $band round $r: value', d['value'], 'steady', d['steady_state_fps'])"
done; done
