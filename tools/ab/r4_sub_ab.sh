#!/bin/bash
# A/B of the subsequence floor (64 bytes always vs 32 bytes always) over batch sizes and depths: where the small-batch rule should end
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4sub
out=gpurun_out/r4sub/ab.txt
: > $out
for b in 1 2 4 8 16 32; do
for depth in 1 6; do
for thr in 0 100000000; do
export UFD_SUB_SMALL_BYTES=$thr
steps=$((2400 / b)); [ $steps -gt 600 ] && steps=600
r=$(timeout -k 10 120 python3 bench.py --variant 640 --batch $b --depth $depth --steps $steps --warmup 20 --no-cpu-baseline --host-only --pool 64 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['steady_state_fps'])")
echo "batch $b depth $depth floor $([ $thr = 0 ] && echo 64 || echo 32): ms_per_step fps $r" | tee -a $out
done; done; done
