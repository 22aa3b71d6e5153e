#!/bin/bash
# What the entropy chain costs the LOADED pipeline: steady-state frame rate with the chain run 0 / 1 / 2 more times per batch.
set -u
name=${1:-r5f}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$name
for r in 1 2; do
  for n in 0 1 2; do
    UFD_REPEAT_ENTROPY=$n timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('entropy x', 1 + $n, d['value'], d['ms_per_step'])" | tee -a gpurun_out/$name/repeat.txt
  done
done
