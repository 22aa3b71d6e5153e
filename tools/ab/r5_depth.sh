#!/bin/bash
# batches in flight over the four contexts, round-5 kernels: steady-state frame rate at depth 4 / 6 / 8 / 12
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
for r in 1 2; do
  for d in 4 6 8 12; do
    timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --depth $d --no-cpu-baseline --no-extras 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('depth', $d, d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5q/depth.txt
  done
done
