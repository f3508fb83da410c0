#!/bin/bash
# rocprofv3 per-kernel statistics of bench.py's cpw_iso leg (config 3's solver loop); summary -> gpurun_out/r06_cpw_ref_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cpwprof -o cpw -- python $GRAFT_REPO_ROOT/scripts/cpw_only.py > $GRAFT_REPO_ROOT/gpurun_out/r06_cpw_ref_profile.log 2>&1
cd $GRAFT_REPO_ROOT
grep -v rocprofv3 gpurun_out/r06_cpw_ref_profile.log | tail -5
find gpurun_out/cpwprof | head
f=$(find gpurun_out/cpwprof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r06_cpw_ref_kernel_stats.csv
python scripts/trace_by_grid.py $(find gpurun_out/cpwprof -name "*kernel_trace.csv" | head -1) 10 > gpurun_out/r06_cpw_ref_by_level.txt
rm -rf gpurun_out/cpwprof
