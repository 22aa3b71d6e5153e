#!/bin/bash
# band height of the 32-channel chained instance (m3 -> m4: 2560 waves at band 1 = 1.25 rounds of the 2048 resident ones) with
# round 5's kernel: UFD_BAND_SMALL = 1 / 2 / 3, alone times and steady-state frame rate
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
for b in 1 2 3; do
  export UFD_BAND_SMALL=$b
  bash tools/kernel_times.sh > gpurun_out/r5p/kernel_times_band$b.txt 2>&1
  echo "== band $b"; grep -E "dwpw2" gpurun_out/r5p/kernel_times_band$b.txt
done
for r in 1 2; do
  for b in 1 2 3; do
    UFD_BAND_SMALL=$b timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('band', $b, d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5p/fps.txt
  done
done
