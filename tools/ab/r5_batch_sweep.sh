#!/bin/bash
# Steady-state frame rate against frames per batch (wave-tile granularity: k_rfb_tail has 100 tiles per frame on 3 072 slots,
# the chained kernel 64 on 2 048, ...).  Usage: tools/ab/r5_batch_sweep.sh <out name> b1 b2 ...
set -u
name=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$name
for r in 1 2; do
  for b in "$@"; do
    timeout -k 10 200 python3 bench.py --batch $b --steps 300 --warmup 10 --no-cpu-baseline --no-variants --no-extras 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('batch $b', d['value'], d['ms_per_step'])" | tee -a gpurun_out/$name/fps.txt
  done
done
