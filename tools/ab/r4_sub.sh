#!/bin/bash
# experiment: shorter subsequences for batches of <= 4 frames (UFD_SUB_MIN_BYTES), parity of the entropy decoder + batch-1 timeline
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4sub
for sub in ${SUBS:-16 24}; do
export UFD_SUB_MIN_BYTES=$sub
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "jpeg or entropy or sync or restart or mjpg or segment" > gpurun_out/r4sub/pytest_$sub.log 2>&1; echo "sub $sub pytest rc=$?"; tail -2 gpurun_out/r4sub/pytest_$sub.log
bash tools/ab/r4_lat.sh r4sub/s$sub | head -12
done
