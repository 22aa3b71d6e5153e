#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4lat2
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_configs.py -x -q -k "not bench_script and not rccl and not reference_pictures and not postproc" > gpurun_out/r4lat2/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4lat2/pytest.log
[ $rc -ne 0 ] && exit 1
bash tools/ab/r4_lat.sh r4lat2
