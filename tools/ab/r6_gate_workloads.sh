#!/bin/bash
# Round 6: the cross-context gate (csrc/pipeline_gate.cpp) on the other workloads -- bench.py's driver sample (20 steps) and
# steady state, with and without UFD_FLAG_NO_GATE, alternating, two rounds.  Usage on the box: tools/ab/r6_gate_workloads.sh <out dir>
set -u
out=gpurun_out/${1:-r6b}
mkdir -p $out
: > $out/gate_workloads.txt
run() {  # tag, bench args...
  tag=$1; shift
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --host-only --no-cpu-baseline "$@" 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$tag', d['value'], d.get('steady_state_fps'))" >> $out/gate_workloads.txt
}
for r in 1 2; do
  run "c3 gate" ; run "c3 nogate" --no-gate
  run "c3-422 gate" --subsampling 4:2:2; run "c3-422 nogate" --subsampling 4:2:2 --no-gate
  run "320-b32 gate" --variant 320; run "320-b32 nogate" --variant 320 --no-gate
  run "c5 gate" --src 1280x720 --batch 16; run "c5 nogate" --src 1280x720 --batch 16 --no-gate
  run "annotate gate" --annotate; run "annotate nogate" --annotate --no-gate
  run "b8 gate" --batch 8; run "b8 nogate" --batch 8 --no-gate
done
cat $out/gate_workloads.txt
