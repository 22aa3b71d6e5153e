#!/bin/bash
# Same-box A/B of two builds of the library (frame rates of different gpurun boxes differ by +-4 %): alternates
# ab/base.so and ab/alt.so in place of infercam_onnx_amd/libufacehip.so and runs the bench ROUNDS times each.
# Usage on the box: tools/ab_bench.sh [rounds] [bench args...]; the library in the tree is restored at the end.
set -u
rounds=${1:-3}; shift || true
lib=infercam_onnx_amd/libufacehip.so
mkdir -p ab && cp $lib ab/_orig.so  # (ab/base.so and ab/alt.so: the two builds, copied there by hand)
mkdir -p gpurun_out/ab
: > gpurun_out/ab/log.txt
for r in $(seq 1 $rounds); do
  for v in base alt; do
    cp ab/$v.so $lib
    timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-variants --no-extras "$@" 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['value'], d['ms_per_step'])" >> gpurun_out/ab/log.txt
  done
done
cp ab/_orig.so $lib
cat gpurun_out/ab/log.txt
python3 - <<'PY'
import collections
v = collections.defaultdict(list)
for l in open('gpurun_out/ab/log.txt'):
    k, fps, ms = l.split()
    v[k].append(float(fps))
for k in v: print(k, 'mean %.0f' % (sum(v[k]) / len(v[k])), 'n', len(v[k]))
PY
