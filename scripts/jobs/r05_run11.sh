#!/bin/bash
# round 5, GPU call 13: rocprofv3 evidence on the round's library -- kernel tables (bench, K + M apply, order 4, tetrahedra, config 3's
# reference leg, eigenmode leg), PMC passes of the headline pair and of the tetrahedral kernels
cd "$GRAFT_REPO_ROOT"
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
rm -rf $O/prof_bench $O/prof_curlmass $O/prof_pmc* $O/prof_p4* $O/prof_tet* $O/prof_cpw $O/prof_eigen
cd /tmp
export PYTHONPATH=$REPO
timeout 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $REPO/$O/prof_bench -- python $REPO/bench.py --no-cpu --no-tets --no-p4 --no-traffic > $REPO/$O/prof_bench.log 2>&1
OP=curlmass REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof_curlmass -- python $REPO/scripts/profile_apply.py > $REPO/$O/prof_curlmass.log 2>&1
ORDER=4 OP=curl REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof_p4_curl -- python $REPO/scripts/profile_apply.py > $REPO/$O/prof_p4_curl.log 2>&1
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  OP=curl REPS=10 CAL8=1 timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $REPO/$O/prof_pmc$i -- python $REPO/scripts/profile_apply.py > $REPO/$O/prof_pmc$i.log 2>&1
done
grep -h '^done' $REPO/$O/prof_pmc1.log | awk '{print $2}' > $REPO/$O/prof_cal_n.txt
timeout 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $REPO/$O/prof_cpw -- python $REPO/scripts/profile_cpw.py > $REPO/$O/prof_cpw.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof_eigen -- python $REPO/scripts/time_eigen.py 3 1.0e6 30 > $REPO/$O/prof_eigen.log 2>&1
cd $REPO
N=36 REPS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tet -- python scripts/time_tet.py > $O/prof_tet.log 2>&1
N=36 REPS=5 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $O/prof_tet_pmc1 -- python scripts/time_tet.py > $O/prof_tet_pmc1.log 2>&1
N=36 REPS=5 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_tet_pmc2 -- python scripts/time_tet.py > $O/prof_tet_pmc2.log 2>&1
N=36 REPS=5 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_tet_pmc3 -- python scripts/time_tet.py > $O/prof_tet_pmc3.log 2>&1
find $O/prof_bench $O/prof_curlmass $O/prof_p4_curl $O/prof_tet $O/prof_tet_pmc1 $O/prof_tet_pmc2 $O/prof_tet_pmc3 $O/prof_pmc1 $O/prof_pmc2 $O/prof_pmc3 $O/prof_pmc4 $O/prof_pmc5 $O/prof_cpw $O/prof_eigen -type f ! -name '*stats*' ! -name '*counter_collection*' ! -name '*marker*' -delete 2>/dev/null
find $O -name '*marker*' -size +8M -delete 2>/dev/null
du -sh $O; for d in prof_bench prof_tet prof_cpw prof_eigen; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -8 "$f" | cut -c1-180; done
grep "^cpw:" $O/prof_cpw.log | head -1 | cut -c1-400; grep "mult" $O/prof_tet.log
