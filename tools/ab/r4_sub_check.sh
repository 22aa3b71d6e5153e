#!/bin/bash
# the 32-byte subsequence floor for small batches: parity tests, whole-file fuzz under it, batch-1 timeline and latency
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4sub2
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "jpeg or entropy or sync or restart or mjpg or segment or rfb_tail or floor" > gpurun_out/r4sub2/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4sub2/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python3 tools/fuzz_gpu.py 1200 > gpurun_out/r4sub2/fuzz_small.log 2>&1; echo "fuzz small rc=$?"; tail -1 gpurun_out/r4sub2/fuzz_small.log
timeout -k 10 400 python3 tools/fuzz_gpu.py 400 big > gpurun_out/r4sub2/fuzz_big.log 2>&1; echo "fuzz big rc=$?"; tail -1 gpurun_out/r4sub2/fuzz_big.log
bash tools/ab/r4_lat.sh r4sub2
