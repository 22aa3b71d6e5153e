# Throughput against the number of HSA hardware queues the HIP runtime may use (GPU_MAX_HW_QUEUES),
# the number of contexts (UFD_CTX) and the other pipeline-level tuning knobs.  Usage on the GPU box:
# bash tools/hwq_probe.sh [set]   (set: queues | chunk)
set -u
B="python bench.py --no-variants --no-cpu-baseline --steps 300 --warmup 30"
run() { echo "== $1"; shift; env "$@" $B 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'])"; }
case "${1:-queues}" in
queues)
  for c in 1 2 3 4; do run "ctx$c" UFD_CTX=$c; done
  for q in 2 3 5 8; do run "hwq$q ctx3" GPU_MAX_HW_QUEUES=$q UFD_CTX=3; done
  ;;
chunk)
  run base A=1
  for ch in 4 8 16; do for l in 5 9; do run "chunk$ch layers<$l" UFD_CHUNK=$ch UFD_CHUNK_LAYERS=$l; done; done
  ;;
esac
