#!/bin/bash
# round 5, GPU call 2: where the time of the device-resident Gram-Schmidt goes (kernel table of one column at config 3's size)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python scripts/time_orthog.py ) > gpurun_out/r2_orthog.log 2>&1
cat gpurun_out/r2_orthog.log | cut -c1-300
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2_prof -o orthog -- python $GRAFT_REPO_ROOT/scripts/time_orthog.py 2180208 100 ) > $GRAFT_REPO_ROOT/gpurun_out/r2_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r2_prof -name "*kernel_stats*" | head; f=$(find gpurun_out/r2_prof -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -c1-260
find gpurun_out/r2_prof -name "*.csv" ! -name "*stats*" -size +1M -delete
