#!/bin/bash
# Alone kernel times (rocprofv3 kernel trace, one batch in flight) of ab/base.so against ab/alt.so on the same box, then the
# SQ counters of the kernels matching $1 under ab/alt.so.  Usage: tools/ab/r5_ab_alone.sh [kernel substring] [out name]
set -u
pat=${1:-k_dwpw2_mfma}
name=${2:-r5ab}
cd $GRAFT_REPO_ROOT
lib=infercam_onnx_amd/libufacehip.so
cp $lib ab/_orig.so
mkdir -p gpurun_out/$name
for v in base alt; do
  cp ab/$v.so $lib
  bash tools/kernel_times.sh > gpurun_out/$name/kernel_times_$v.txt 2>&1
  echo "== $v"; grep -E "dwpw|stem|rfb|rows|total" gpurun_out/$name/kernel_times_$v.txt
done
cp ab/alt.so $lib
bash tools/pmc_kernel.sh "$pat" > gpurun_out/$name/sq_alt.txt 2>&1; tail -12 gpurun_out/$name/sq_alt.txt
cp gpurun_out/pmc_kernel/sq_counters.json gpurun_out/$name/sq_counters_alt.json
cp ab/_orig.so $lib
