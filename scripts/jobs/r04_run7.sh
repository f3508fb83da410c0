#!/bin/bash
# round 4, GPU call 7: slot-pattern dictionary of the five-point kernel + conflict-free staging rows of the four-point one:
# correctness, then A/B against the library at the previous commit (head) on one box, alternating
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 1500 python -m pytest -q -x -m gpu tests/test_apply_gpu.py tests/test_stream5_gpu.py tests/test_complex_gpu.py tests/test_split_gpu.py tests/test_solvers_gpu.py tests/test_fullsize_gpu.py tests/test_halo_gpu.py ) > $O/r7_tests.log 2>&1
echo "tests exit $?"; tail -4 $O/r7_tests.log
L=$PWD/palace_amd/lib
for v in head default head default; do
  if [ $v = default ]; then unset PALACE_AMD_LIB; else export PALACE_AMD_LIB=$L/libpalace_amd_$v.so; fi
  timeout 400 python scripts/price_evec_cache.py 2>&1 | tail -1 | sed "s/^/[$v] /" | tee -a $O/r7_price.log
  TAG=$v timeout 400 python scripts/time_k.py 2>&1 | tail -1 | tee -a $O/r7_time_k.log
  P4_VARIANTS=default timeout 400 python scripts/time_p4.py 2>&1 | tail -1 | sed "s/^/[$v] /" | tee -a $O/r7_time_p4.log
done
