#!/bin/bash
# round 4, GPU call 11: kernel table of the per-rank proxy of the 8-GPU case (one Mult loop + PCG with halos on every level)
cd "$GRAFT_REPO_ROOT"
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_proxy
cd /tmp
PYTHONPATH=$REPO PCG=50 REPS=200 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_proxy -- python $REPO/scripts/time_halo_mult.py > $REPO/gpurun_out/prof_proxy.log 2>&1
echo "proxy profile exit $?"
cd $REPO
find gpurun_out/prof_proxy -type f ! -name '*stats*' -delete
grep -E "slab|it/s" gpurun_out/prof_proxy.log
head -30 gpurun_out/prof_proxy/*/*kernel_stats.csv | cut -c1-220
