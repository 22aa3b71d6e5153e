#!/bin/bash
# Runs on the MI355X box (gpurun): bench line, rocprofv3 kernel stats and the two PMC passes for
# HBM traffic, all into gpurun_out/<name>/.  Usage: tools/collect_profiles.sh <name>
set -u
name=${1:-prof}
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 -c "import bench; print(bench.kernel_source_sha())" > $out/source_sha.txt  # what THIS run measures (finish_profiles.py stamps with it)
timeout 600 python3 bench.py --steps 80 --warmup 8 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-variants --profile-every 1 > $out/bench_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --depth 1 --no-cpu-baseline --no-variants --pool 64 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --depth 1 --no-cpu-baseline --no-variants --pool 64 > /dev/null 2>&1
find $out -name "*.csv" < /dev/null | head -20
# then, back in the build container: python tools/finish_profiles.py <name>
