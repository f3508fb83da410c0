#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes, counters only with --kernel-trace) of config 3's solver loop -> gpurun_out/r06_cpw_iso_pmc.csv
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/cpwpmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cpwpmc_$c -- python $GRAFT_REPO_ROOT/scripts/cpw_iso_only.py > $GRAFT_REPO_ROOT/gpurun_out/r06_cpw_iso_pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_by_kernel.py gpurun_out/cpwpmc_FETCH_SIZE gpurun_out/cpwpmc_WRITE_SIZE 100 > gpurun_out/r06_cpw_iso_pmc.csv
rm -rf gpurun_out/cpwpmc_FETCH_SIZE gpurun_out/cpwpmc_WRITE_SIZE
head -30 gpurun_out/r06_cpw_iso_pmc.csv
