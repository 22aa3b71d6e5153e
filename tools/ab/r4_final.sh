#!/bin/bash
# round-end rehearsal on the GPU box: the driver's three steps (pytest -m gpu, smoke, bench) and the evidence collection
set -u
name=${1:-r4z}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$name
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/$name/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -4 gpurun_out/$name/pytest_gpu.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/ab/r4_collect.sh $name
