#!/bin/bash
# SQ counters of one kernel (name substring), one rocprofv3 --pmc pass per counter group, kernel alone on
# the GPU (--depth 1).  Usage on the box: tools/pmc_kernel.sh <substring>; prints per-counter means.
set -u
pat=${1:-k_dwpw2_mfma}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_kernel
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
groups=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_INST_CYCLES_VMEM_RD" "SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL")
i=0
for g in "${groups[@]}"; do
  timeout 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out/g$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --depth 1 --no-cpu-baseline --no-variants --pool 64 > /dev/null 2> $out/g$i.err || echo "group $i failed: $(tail -2 $out/g$i.err)"
  i=$((i+1))
done
python3 - "$pat" $out <<'PY'
import csv, glob, sys, collections, json, re
pat, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            key = re.sub(r'^void |ufd::\(anonymous namespace\)::|\(.*$', '', r['Kernel_Name'])
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
means = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(means, open(out + '/sq_counters.json', 'w'), indent=1, sort_keys=True)
# per kernel: share of the SIMD-cycles the kernel held the GPU (GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs)
print('%-34s %8s %7s %7s %7s %7s %7s %8s %8s' % ('kernel', 'us', 'mfma%', 'valu%', 'coex%', 'wait%', 'occ', 'ic_miss%', 'ldsconf%'))
for k, m in sorted(means.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0)):
    g = m.get('GRBM_GUI_ACTIVE', 0) / 8.0
    if g <= 0:
        continue
    simd_cycles = g * 1024.0
    pct = lambda x, d=simd_cycles: 100.0 * m.get(x, 0) / d if d else 0.0
    # SQ_VALU_MFMA_BUSY_CYCLES / SQ_ACTIVE_INST_VALU count in quad-cycles per SIMD on gfx9 (x4 -> cycles); kept raw in the json
    print('%-34s %8.1f %7.1f %7.1f %7.1f %7.1f %7.2f %8.2f %8.2f' % (
        k[:34], g / 2400.0, pct('SQ_VALU_MFMA_BUSY_CYCLES'), pct('SQ_ACTIVE_INST_VALU') * 4, pct('SQ_VALU_MFMA_COEXEC_CYCLES'),
        100.0 * m.get('SQ_WAIT_INST_ANY', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1), m.get('SQ_WAVE_CYCLES', 0) / max(simd_cycles / 4, 1) / 4,
        100.0 * m.get('SQC_ICACHE_MISSES', 0) / max(m.get('SQC_ICACHE_REQ', 1), 1),
        100.0 * m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_ACTIVE_INST_LDS', 1), 1)))
PY
