# Throughput against the number of HSA hardware queues the HIP runtime may use (GPU_MAX_HW_QUEUES)
# and the number of contexts (UFD_CTX).  Usage on the GPU box: bash tools/hwq_probe.sh
set -u
B="python bench.py --no-variants --no-cpu-baseline --steps 300 --warmup 30 --depth 8"
run() { echo "== $1"; shift; env "$@" $B 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'])"; }
for c in 3 4 6; do run "nocopy hwq3 ctx$c" UFD_COPY_STREAM=0 GPU_MAX_HW_QUEUES=3 UFD_CTX=$c; done
for c in 3 4 5 6 8; do run "nocopy hwq4 ctx$c" UFD_COPY_STREAM=0 GPU_MAX_HW_QUEUES=4 UFD_CTX=$c; done
for c in 5 6; do run "copy hwq4 ctx$c" GPU_MAX_HW_QUEUES=4 UFD_CTX=$c; done
