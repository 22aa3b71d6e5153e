#!/bin/bash
# throw-away: alone kernel times of ab/base.so against ab/alt.so (an EXPERIMENT build whose results may be garbage)
set -u
cd $GRAFT_REPO_ROOT
lib=infercam_onnx_amd/libufacehip.so
cp $lib ab/_orig.so
for v in base alt; do
  cp ab/$v.so $lib
  bash tools/kernel_times.sh > gpurun_out/exp_$v.txt 2>&1
  echo "== $v"; grep -E "${1:-dwpw2}" gpurun_out/exp_$v.txt
done
cp ab/_orig.so $lib
