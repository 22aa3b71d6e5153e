#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/fill
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/t -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench.json 2>/dev/null
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
head -1 $f
python3 $GRAFT_REPO_ROOT/tools/fill_drain_timeline.py $f 20 5 | tee $out/timeline.txt
tail -1 $out/bench.json | cut -c1-150
rm -rf $out/t
