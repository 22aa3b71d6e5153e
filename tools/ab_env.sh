#!/bin/bash
# Same-box A/B of an environment knob: alternates `bench.py` with and without VAR=VALUE, ROUNDS times each.
# Usage on the box: tools/ab_env.sh VAR=VALUE [rounds] [bench args...]
set -u
kv=$1; rounds=${2:-3}; shift; shift || true
mkdir -p gpurun_out/ab
: > gpurun_out/ab/env.txt
run() { timeout -k 10 200 "$@" python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-variants --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$tag', d['value'], d['ms_per_step'])" >> gpurun_out/ab/env.txt; }
for r in $(seq 1 $rounds); do
  tag=base; run env
  tag=$kv; run env $kv
done
cat gpurun_out/ab/env.txt
