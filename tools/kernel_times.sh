#!/bin/bash
# Per-kernel average durations with one batch in flight (rocprofv3 kernel trace, --depth 1).
# Usage on the box: tools/kernel_times.sh [extra bench args]
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/ktimes
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 4 --depth 1 --no-cpu-baseline --no-variants "$@" > $out/bench.json 2>/dev/null
f=$(find $out -name "*kernel_stats.csv" < /dev/null | head -1)
if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    name = re.sub(r'ufd::\(anonymous namespace\)::', '', r['Name']).split('(')[0][:44]
    calls = int(r['Calls']); avg = float(r['AverageNs']) / 1e3
    tot += float(r['TotalDurationNs'])
    if float(r['Percentage']) > 0.4: print(f'{name:46s} {calls:5d} {avg:9.1f} us {float(r["Percentage"]):5.1f}%')
print('total kernel ms', tot / 1e6)
PY
fi
