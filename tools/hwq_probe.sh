# Throughput against the number of HSA hardware queues the HIP runtime may use (GPU_MAX_HW_QUEUES)
# and the number of contexts (UFD_CTX).  Usage on the GPU box: bash tools/hwq_probe.sh
set -u
B="python bench.py --no-variants --no-cpu-baseline --steps 300 --warmup 30"
run() { echo "== $1"; shift; env "$@" $B 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'])"; }
B0="$B"
for d in 3 6 8; do B="$B0 --depth $d"; run "ctx3 depth$d" A=1; done
for d in 4 8; do B="$B0 --depth $d"; run "ctx4 depth$d" UFD_CTX=4; done
B="$B0 --depth 6"; run "ctx3 depth6 again" A=1
