#!/bin/bash
# Round-5 kernel iteration on the GPU box: conv parity subset, steady-state bench, alone kernel times.
# Usage: tools/ab/r5_kern.sh <name> [pytest -k expression]
set -u
name=${1:-r5k}
kexpr=${2:-"not bench_script and not rccl and not reference_pictures"}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$name
mkdir -p $out
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_configs.py -x -q -k "$kexpr" > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $out/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 bench.py --host-only --steps 300 --warmup 20 > $out/bench_host_only.json 2> $out/bench.err; python3 -c "
import json; d=json.loads(open('$out/bench_host_only.json').read().strip().splitlines()[-1]); print('value', d['value'], 'steady', d['steady_state_fps'], d['host']['gpu_span_ms_per_batch'])"
bash tools/kernel_times.sh > $out/kernel_times_alone.txt 2>&1; head -34 $out/kernel_times_alone.txt
