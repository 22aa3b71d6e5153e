#!/bin/bash
# (the dw -> pw family diet this A/B measured was not kept: docs/EXPERIMENTS.md round 5)
# round 5, fourth kernel step: the stride-1 dw -> pw family on dw_taps_cm; then the SQ counters of every kernel (alt build)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_configs.py -x -q -k "not bench_script and not rccl and not reference_pictures" > gpurun_out/r5i/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r5i/pytest.log
[ $rc -ne 0 ] && exit 1
bash tools/ab/r5_variants.sh r5i "dwpw_mfma|dwpw_coop|dual|total" base alt
bash tools/pmc_kernel.sh k_ > gpurun_out/r5i/sq_counters.txt 2>&1; tail -30 gpurun_out/r5i/sq_counters.txt
cp gpurun_out/pmc_kernel/sq_counters.json gpurun_out/r5i/sq_counters.json
