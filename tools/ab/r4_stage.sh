#!/bin/bash
# k_stage_in (small batches fetch their staging block by a kernel on the context's stream): parity, A/B of the batch-1 latency, soak
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4stage
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_mirrors.py -x -q -k "jpeg or entropy or sync or restart or mjpg or segment or floor or end_to_end or soak or concurrent" > gpurun_out/r4stage/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4stage/pytest.log
[ $rc -ne 0 ] && exit 1
for r in 1 2 3; do
for v in kernel copy; do
pre=""; [ $v = copy ] && pre="env UFD_NO_STAGE_KERNEL=1"
for b in 1 4; do
$pre timeout -k 10 120 python3 bench.py --variant 640 --batch $b --depth 1 --steps 400 --warmup 20 --no-cpu-baseline --host-only --pool 64 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v round $r batch $b depth 1: ms', d['ms_per_step'], d['host']['per_batch_us'], d['host']['launches_per_batch'])"
done
$pre timeout -k 10 120 python3 bench.py --variant 640 --batch 1 --depth 6 --steps 600 --warmup 20 --no-cpu-baseline --host-only --pool 64 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v round $r batch 1 depth 6: fps', d['steady_state_fps'])"
done; done
