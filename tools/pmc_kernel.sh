#!/bin/bash
# SQ counters of one kernel (name substring), one rocprofv3 --pmc pass per counter group, kernel alone on
# the GPU (--depth 1).  Usage on the box: tools/pmc_kernel.sh <substring>; prints per-counter means.
set -u
pat=${1:-k_dwpw2_mfma}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_kernel
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
groups=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum")
i=0
for g in "${groups[@]}"; do
  timeout 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out/g$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --depth 1 --no-cpu-baseline --no-variants --pool 64 > /dev/null 2> $out/g$i.err || echo "group $i failed: $(tail -2 $out/g$i.err)"
  i=$((i+1))
done
python3 - "$pat" $out <<'PY'
import csv, glob, sys, collections
pat, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            key = r['Kernel_Name'].split('(')[0][-40:]
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f'  {c:36s} n={len(v):3d} mean={sum(v)/len(v):.4g}')
PY
