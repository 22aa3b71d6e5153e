#!/bin/bash
# (UFD_DWPW_DEPTH and the ab/tail3.so build existed only for this measurement: docs/EXPERIMENTS.md round 5, "not kept")
# round 5, third kernel step: k_rfb_tail at three waves per SIMD; k_dwpw_mfma with four k-steps of windows in flight
set -u
cd $GRAFT_REPO_ROOT
lib=infercam_onnx_amd/libufacehip.so
cp $lib ab/_orig.so
mkdir -p gpurun_out/r5h
run() {  # <label> <lib> <depth>
  cp ab/$2.so $lib
  export UFD_DWPW_DEPTH=$3
  bash tools/kernel_times.sh > gpurun_out/r5h/kernel_times_$1.txt 2>&1
  echo "== $1"; grep -E "rfb|dwpw_mfma|dual|total" gpurun_out/r5h/kernel_times_$1.txt
}
run base base 2
run depth4 base 4
run tail3 tail3 2
for r in 1 2; do
  for v in "base base 2" "depth4 base 4" "tail3 tail3 2" "both tail3 4"; do
    set -- $v
    cp ab/$2.so $lib
    UFD_DWPW_DEPTH=$3 timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5h/fps.txt
  done
done
cp ab/_orig.so $lib
