#!/bin/bash
# (UFD_DWPW2_LDS_PAD existed only for this measurement: docs/EXPERIMENTS.md round 5, "Room beside the chained kernel?")
# one block of the chained kernel per CU (its LDS request padded beyond half a CU's LDS): does the room it leaves pay?
# pad 0: as shipped (m1->m2: 80 KB, two blocks per CU; m3->m4: 25 KB); 8192: m1->m2 one block per CU, m3->m4 unchanged in effect;
# 90000: both one block per CU
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
for r in 1 2; do
  for pad in 0 8192 90000; do
    UFD_DWPW2_LDS_PAD=$pad timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('pad', $pad, d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5m/pad.txt
  done
done
