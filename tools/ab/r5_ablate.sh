#!/bin/bash
# What each launch costs the LOADED pipeline: steady-state frame rate with the launch(es) of the given layer indices
# skipped (UFD_ABLATE_LAYERS, results garbage, timing only).  Usage: tools/ab/r5_ablate.sh <out name> "set1" "set2" ...
# ("-" = nothing skipped; a set is a comma-separated list of layer indices: 0 stem, 4 m1->m2, 8 m3->m4, 10 m5, 12 m6, 13 /
# 14 / 21 / 24 the RFB launches, 26 heads0|m8, 32 m9, 34 m10, 36 heads1|m11, 42 m12, 44 heads2|extra.0, 48-50 the tail)
set -u
name=$1; shift
cd $GRAFT_REPO_ROOT
# (round 6: the knob exists only in the measurement build)
make -s -C infercam_onnx_amd/csrc EXPERIMENTS=1 -j16 && export UFD_LIBRARY=$PWD/infercam_onnx_amd/libufacehip_exp.so || exit 1
mkdir -p gpurun_out/$name
for r in 1 2; do
  for s in "$@"; do
    if [ "$s" = "-" ]; then unset UFD_ABLATE_LAYERS; else export UFD_ABLATE_LAYERS=$s; fi
    timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('skip', '$s', d['value'], d['ms_per_step'])" | tee -a gpurun_out/$name/ablate.txt
  done
done
