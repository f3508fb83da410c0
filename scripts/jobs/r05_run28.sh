#!/bin/bash
# round 5, GPU call 28: kernel trace of the headline apply (curl-curl) and of K + M on the final library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for op in curl curlmass; do
  OP=$op REPS=200 PYTHONPATH=$R timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$op -- python $R/scripts/profile_apply.py > /dev/null 2> $R/gpurun_out/prof_$op.err
  f=$(find $R/gpurun_out/prof_$op -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r05_final_apply_${op}_kernel_stats.csv && head -4 $R/gpurun_out/r05_final_apply_${op}_kernel_stats.csv | cut -c1-160
  rm -rf $R/gpurun_out/prof_$op
done
