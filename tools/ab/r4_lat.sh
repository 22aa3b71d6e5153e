#!/bin/bash
set -u
name=${1:-r4lat}
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in 640 320; do
rm -rf $out/trace$v
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace$v -o k -- python3 $GRAFT_REPO_ROOT/bench.py --variant $v --batch 1 --depth 1 --steps 40 --warmup 5 --no-extras --no-cpu-baseline --pool 32 > $out/bench_b1_$v.json 2>/dev/null
f=$(find $out/trace$v -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/batch1_timeline.py $f > $out/timeline_$v.txt; tail -1 $out/timeline_$v.txt
rm -rf $out/trace$v
done
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 bench.py --variant 640 --batch 1 --depth 1 --steps 200 --warmup 20 --no-cpu-baseline --host-only 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1 640:', d['ms_per_step'], d['host']['per_batch_us'], d['host']['launches_per_batch'])"
cat $out/timeline_640.txt
