#!/bin/bash
# PC sampling of the pipeline with one batch in flight (rocprofv3 beta feature): where the waves of each kernel are when sampled
set -u
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pcs
rm -rf $out; mkdir -p $out
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout -k 10 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 65536 --kernel-trace --output-format csv -d $out/st -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 2 --depth 1 --no-extras --no-cpu-baseline --pool 64 > $out/st.log 2>&1
echo "stochastic rc=$?"; tail -5 $out/st.log; ls -la $out/st 2>/dev/null | head
if ! ls $out/st/*pc_sampling* > /dev/null 2>&1; then
timeout -k 10 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 --kernel-trace --output-format csv -d $out/ht -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 2 --depth 1 --no-extras --no-cpu-baseline --pool 64 > $out/ht.log 2>&1
echo "host_trap rc=$?"; tail -5 $out/ht.log; ls -la $out/ht 2>/dev/null | head
fi
find $out -name "*.csv" -size +60M -delete
du -sh $out
