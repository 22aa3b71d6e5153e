#!/bin/bash
# round 5, second kernel step: parity subset, alone times and frame rates base vs alt, then the entropy decoder's speculation rounds
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_configs.py -x -q -k "not bench_script and not rccl and not reference_pictures" > gpurun_out/r5g/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r5g/pytest.log
[ $rc -ne 0 ] && exit 1
bash tools/ab/r5_variants.sh r5g "dwpw2|stem|rfb|rows|total" base alt
for r in 1 2; do
  for n in 2 1 3; do
    UFD_EXTEND_ROUNDS=$n timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('extend rounds', $n, d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5g/extend_rounds.txt
  done
done
