#!/bin/bash
# round-4 first GPU call: host-stats test, driver-flag bench line, host scaling sweep, alone kernel times, SQ counters
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4a
mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_bench_configs.py -x -q -k "host_stats or c3_batch32_bench_pipeline" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_flags.json 2> $out/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4a/bench_driver_flags.json').read().strip().splitlines()[-1])
print(d['value'], d.get('steady_state_fps'), json.dumps(d.get('host')))
PY
timeout -k 10 600 python3 tools/host_scaling.py $out/host_scaling.json --cpus 0,16,8,4,2 --steps 300
