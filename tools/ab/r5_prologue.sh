#!/bin/bash
# Builds ab/<v>.so on the same box: alone kernel times, steady-state frame rate (2 alternating rounds) and one frame at a
# time (batch 1, depth 1: ms per step) at 640 and 320.  Usage: tools/ab/r5_prologue.sh <out name> v1 v2 ...
set -u
name=$1; shift
cd $GRAFT_REPO_ROOT
lib=infercam_onnx_amd/libufacehip.so
cp $lib ab/_orig.so
mkdir -p gpurun_out/$name
for v in "$@"; do
  cp ab/$v.so $lib
  bash tools/kernel_times.sh > gpurun_out/$name/kernel_times_$v.txt 2>&1
  echo "== $v"; grep -E "total" gpurun_out/$name/kernel_times_$v.txt
done
for r in 1 2; do
  for v in "$@"; do
    cp ab/$v.so $lib
    timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-variants --no-extras 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/$name/fps.txt
    for var in 640 320; do
      timeout -k 10 200 python3 bench.py --variant $var --batch 1 --depth 1 --steps 300 --warmup 20 --no-cpu-baseline --no-extras --no-variants --pool 32 2>/dev/null |
        python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v batch1 $var ms', d['ms_per_step'])" | tee -a gpurun_out/$name/fps.txt
    done
  done
done
cp ab/_orig.so $lib
