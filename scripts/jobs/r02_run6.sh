#!/bin/bash
# Round 2, GPU call 6: warm A/B of the occupancy / placement variants, kernel trace and PMC of y = A x.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02_6
rm -rf $OUT && mkdir -p $OUT
export PYTHONPATH=$REPO
L=$REPO/palace_amd/lib
for cfg in "libpalace_amd.so 1 0" "libpalace_amd.so 2 0" "libpalace_amd.so 0 0" "libpalace_amd_minw2.so 1 0" "libpalace_amd_minw2early.so 1 0" "libpalace_amd_minw2early.so 0 0" "libpalace_amd.so 1 5" "libpalace_amd.so 1 1"; do
  set -- $cfg
  echo "== $1 GPOS=$2 WG=$3" | tee -a $OUT/time_apply.log
  PALACE_AMD_LIB=$L/$1 PALACE_AMD_STREAM_GPOS=$2 PALACE_AMD_STREAM_WG=$3 timeout 300 python scripts/time_apply.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/time_apply.log
done
cd /tmp && export TMPDIR=/tmp
for op in curl curlmass; do
  OP=$op REPS=10 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$op -- python $REPO/scripts/profile_apply.py > $OUT/prof_$op.log 2>&1
done
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  OP=curl REPS=5 CAL8=1 timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/prof_pmc$i -- python $REPO/scripts/profile_apply.py > $OUT/prof_pmc$i.log 2>&1
done
cd $REPO
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -4 $f | cut -c1-140; done
python scripts/summarize_pmc.py $OUT
