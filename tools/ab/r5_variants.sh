#!/bin/bash
# Alone kernel times (rocprofv3 kernel trace, one batch in flight) and steady-state frame rate of several builds ab/<v>.so on
# the same box.  Usage: tools/ab/r5_variants.sh <out name> <grep pattern> v1 v2 ...   (the library in the tree is restored)
set -u
name=$1; pat=$2; shift 2
cd $GRAFT_REPO_ROOT
lib=infercam_onnx_amd/libufacehip.so
cp $lib ab/_orig.so
mkdir -p gpurun_out/$name
for v in "$@"; do
  cp ab/$v.so $lib
  bash tools/kernel_times.sh > gpurun_out/$name/kernel_times_$v.txt 2>&1
  echo "== $v"; grep -E "$pat" gpurun_out/$name/kernel_times_$v.txt
done
for r in 1 2; do
  for v in "$@"; do
    cp ab/$v.so $lib
    timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-variants --no-extras 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/$name/fps.txt
  done
done
cp ab/_orig.so $lib
