#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout -k 10 600 python3 tools/host_scaling.py $out/host_scaling.json --cpus 0,8,4,2 --steps 300
timeout -k 10 200 python3 bench.py --host-only --steps 300 --spin-wait > $out/bench_spin.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('$out/bench_spin.json').read().strip().splitlines()[-1]); print('spin', d['value'], d['steady_state_fps'], d['host']['per_batch_us'])"
