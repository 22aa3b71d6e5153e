#!/bin/bash
# kernel iteration on the GPU box: conv parity subset, steady-state bench, alone kernel times, PMC fetch/write passes
set -u
name=${1:-r4k}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$name
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_configs.py -x -q -k "not bench_script and not rccl and not reference_pictures" > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $out/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 bench.py --host-only --steps 300 --warmup 20 > $out/bench_host_only.json 2> $out/bench.err; python3 -c "
import json; d=json.loads(open('$out/bench_host_only.json').read().strip().splitlines()[-1]); print('value', d['value'], 'steady', d['steady_state_fps'], d['host']['gpu_span_ms_per_batch'])"
bash tools/kernel_times.sh > $out/kernel_times_alone.txt 2>&1; cat $out/kernel_times_alone.txt | head -30
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_fetch -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --depth 1 --no-cpu-baseline --no-variants --pool 64 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_write -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --depth 1 --no-cpu-baseline --no-variants --pool 64 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out/pmc_fetch -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $f $w $out/pmc_traffic.json | tail -40
