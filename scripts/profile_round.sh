#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the default bench command (no CPU leg)
#   2. kernel trace + stats of 20 curl-curl+mass applies
#   3. PMC passes (separate runs, counters only with --kernel-trace) of the curl-curl apply
#   4. the same for the p = 4 streaming kernel (kernel stats + FETCH / WRITE passes)
# Raw output goes to gpurun_out/prof_*; scripts/summarize_profiles.py turns it into profiles/.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
rm -rf $OUT/prof_bench $OUT/prof_curlmass $OUT/prof_pmc* && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
timeout 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $OUT/prof_bench -- python $REPO/bench.py --no-cpu --no-tets --no-p4 > $OUT/prof_bench.log 2>&1
OP=curlmass REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_curlmass -- python $REPO/scripts/profile_apply.py > $OUT/prof_curlmass.log 2>&1
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  OP=curl REPS=10 CAL8=1 timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/prof_pmc$i -- python $REPO/scripts/profile_apply.py > $OUT/prof_pmc$i.log 2>&1
done
# 4. the p = 4 streaming kernel (bench.py's p4 leg): kernel stats of curl-curl and curl-curl + mass, FETCH / WRITE passes of both
rm -rf $OUT/prof_p4* 
ORDER=4 OP=curl REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_p4_curl -- python $REPO/scripts/profile_apply.py > $OUT/prof_p4_curl.log 2>&1
ORDER=4 OP=curlmass REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_p4_curlmass -- python $REPO/scripts/profile_apply.py > $OUT/prof_p4_curlmass.log 2>&1
i=0
for op in curl curlmass; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    ORDER=4 OP=$op REPS=10 timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/prof_p4pmc_${op}_$pmc -- python $REPO/scripts/profile_apply.py > $OUT/prof_p4pmc$i.log 2>&1
  done
done
cd $REPO
grep -h '^done' $OUT/prof_pmc1.log | awk '{print $2}' > $OUT/prof_cal_n.txt
# then, back in the work tree: CAL_N=$(cat gpurun_out/prof_cal_n.txt) python scripts/summarize_profiles.py
