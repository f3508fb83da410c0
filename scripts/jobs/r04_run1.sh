#!/bin/bash
# round 4, GPU call 1: multi-rank hardening tests (stress, rehearsal), A/B of the LDS element stride and K + M occupancy,
# E-vector cache pricing, LDS counters, cpw profile, bench-size rehearsal with 8 ranks
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 1500 python -m pytest -q -x -m gpu tests/test_peer_gpu.py tests/test_halo_gpu.py tests/test_split_gpu.py tests/test_apply_gpu.py tests/test_h1_gpu.py tests/test_complex_gpu.py tests/test_solvers_gpu.py tests/test_ams_gpu.py ) > $O/r1_tests.log 2>&1
echo "tests exit $?" >> $O/r1_tests.log; tail -5 $O/r1_tests.log
( time timeout 1500 python -m pytest -q -x -m gpu tests/test_rehearse_gpu.py ) > $O/r1_rehearse_tests.log 2>&1
echo "rehearse tests exit $?" >> $O/r1_rehearse_tests.log; tail -5 $O/r1_rehearse_tests.log
L=$PWD/palace_amd/lib
for v in default oldlds km3 default; do
  if [ $v = default ]; then unset PALACE_AMD_LIB; else export PALACE_AMD_LIB=$L/libpalace_amd_$v.so; fi
  TAG=$v timeout 400 python scripts/time_k.py 2>&1 | tail -1 | tee -a $O/r1_time_k.log
done
unset PALACE_AMD_LIB
timeout 400 python scripts/price_evec_cache.py 2>&1 | tail -2 | tee $O/r1_price.log
cd /tmp
for v in default oldlds; do
  if [ $v = default ]; then unset PALACE_AMD_LIB; else export PALACE_AMD_LIB=$L/libpalace_amd_$v.so; fi
  rm -rf $GRAFT_REPO_ROOT/$O/prof_lds_$v
  PYTHONPATH=$GRAFT_REPO_ROOT OP=curl REPS=10 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_lds_$v -- python $GRAFT_REPO_ROOT/scripts/profile_apply.py > $GRAFT_REPO_ROOT/$O/prof_lds_$v.log 2>&1
  echo "lds pmc $v exit $?"
done
unset PALACE_AMD_LIB
rm -rf $GRAFT_REPO_ROOT/$O/prof_cpw
PYTHONPATH=$GRAFT_REPO_ROOT timeout 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_cpw -- python $GRAFT_REPO_ROOT/scripts/profile_cpw.py > $GRAFT_REPO_ROOT/$O/prof_cpw.log 2>&1
echo "cpw profile exit $?"; tail -2 $GRAFT_REPO_ROOT/$O/prof_cpw.log | cut -c1-600
cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python bench.py --rehearse 8 --no-cpu ) > $O/r1_rehearse8.json 2> $O/r1_rehearse8.err
echo "rehearse8 exit $?"; tail -c 1500 $O/r1_rehearse8.err; head -c 3000 $O/r1_rehearse8.json
# keep the profiles small: only the stats / counter csv files travel back
find $O/prof_cpw $O/prof_lds_default $O/prof_lds_oldlds -type f ! -name '*stats*' ! -name '*counter_collection*' ! -name '*marker*' -delete 2>/dev/null
du -sh $O
