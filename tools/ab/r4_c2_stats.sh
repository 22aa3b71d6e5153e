#!/bin/bash
# rocprofv3 --kernel-trace --stats of the C2 configuration (UltraFace-320, one 320x240 frame at a time): per-kernel summary
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r4c2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --variant 320 --batch 1 --depth 1 --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $out/bench_c2_under_rocprof.json 2> $out/err.txt; echo "rc=$?"
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_c2_batch1.csv; head -12 $out/kernel_stats_c2_batch1.csv
rm -rf $out/prof
