python tools/wide_msb_bench.py 28 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_wide_prof -- python3 $GRAFT_REPO_ROOT/tools/wide_msb_bench.py 26 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r2_wide_prof -name "*kernel_trace.csv" -delete; head -20 $(find gpurun_out/r2_wide_prof -name "*kernel_stats.csv" | head -1) | cut -c1-160
