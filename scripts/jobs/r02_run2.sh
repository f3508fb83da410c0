#!/bin/bash
# Round 2, GPU call 2: streaming kernel after the ticket / wait fixes: parity, A/B timing, kernel trace, PMC passes.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02_2
rm -rf $OUT && mkdir -p $OUT
export PYTHONPATH=$REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
for cfg in "1 1 0" "1 0 0" "1 2 0" "0 0 0" "1 1 4" "1 1 5"; do
  set -- $cfg
  echo "== STREAM=$1 GPOS=$2 WG=$3" >> $OUT/time_apply.log
  PALACE_AMD_STREAM=$1 PALACE_AMD_STREAM_GPOS=$2 PALACE_AMD_STREAM_WG=$3 timeout 300 python scripts/time_apply.py >> $OUT/time_apply.log 2>&1
done
grep -v "Warning\|amdgpu.ids" $OUT/time_apply.log
cd /tmp && export TMPDIR=/tmp
for op in curl curlmass; do
  OP=$op REPS=10 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$op -- python $REPO/scripts/profile_apply.py > $OUT/prof_$op.log 2>&1
done
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  OP=curl REPS=5 CAL8=1 timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/prof_pmc$i -- python $REPO/scripts/profile_apply.py > $OUT/prof_pmc$i.log 2>&1
  tail -2 $OUT/prof_pmc$i.log
done
cd $REPO
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -6 $f | cut -c1-160; done
python scripts/summarize_pmc.py $OUT
