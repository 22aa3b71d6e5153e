#!/bin/bash
# speculation rounds of the entropy decoder for one frame at a time: resolve's misses (avg / max) and the mean latency
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r4rounds
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in 320 640; do
for rounds in 2 3 4; do
export UFD_EXTEND_ROUNDS=$rounds
rm -rf $out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --variant $v --batch 1 --depth 1 --steps 400 --warmup 20 --no-extras --no-cpu-baseline --pool 128 > /dev/null 2> $out/err.txt
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
echo "variant $v rounds $rounds:"; grep "k_huff_resolve\|k_huff_extend" $f | awk -F'",' '{print "   ", substr($1,1,60), $2}'
rm -rf $out/prof
( cd $GRAFT_REPO_ROOT && timeout -k 10 120 python3 bench.py --variant $v --batch 1 --depth 1 --steps 600 --warmup 20 --no-cpu-baseline --host-only --pool 128 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('    mean ms per frame', d['ms_per_step'])" )
done; done
