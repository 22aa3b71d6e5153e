#!/bin/bash
# SQ counters of one kernel (name substring), one rocprofv3 --pmc pass per counter group, kernel alone on
# the GPU (--depth 1).  Usage on the box: tools/pmc_kernel.sh <substring>; prints per-counter means.
set -u
pat=${1:-k_dwpw2_mfma}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_kernel
rm -rf $out; mkdir -p $out
(cd $GRAFT_REPO_ROOT && python3 -c "import bench; print(bench.kernel_source_sha())" > $out/source_sha.txt)
cd /tmp && export TMPDIR=/tmp
groups=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_INST_CYCLES_VMEM_RD" "SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL")
i=0
for g in "${groups[@]}"; do
  timeout 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out/g$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --depth 1 --no-cpu-baseline --no-extras --pool 64 > /dev/null 2> $out/g$i.err || echo "group $i failed: $(tail -2 $out/g$i.err)"
  i=$((i+1))
done
python3 $GRAFT_REPO_ROOT/tools/sq_table.py $out "$pat"
