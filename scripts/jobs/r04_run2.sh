#!/bin/bash
# round 4, GPU call 2: the whole -m gpu suite on the current library (LDS stride, H1 / dense split forms, plan slots), A/B of the
# q-data-ahead variant
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 2400 python -m pytest -q -m gpu tests -x ) > $O/r2_tests.log 2>&1
echo "tests exit $?" >> $O/r2_tests.log; tail -8 $O/r2_tests.log
L=$PWD/palace_amd/lib
for v in default qahead default qahead; do
  if [ $v = default ]; then unset PALACE_AMD_LIB; else export PALACE_AMD_LIB=$L/libpalace_amd_$v.so; fi
  TAG=$v timeout 400 python scripts/time_k.py 2>&1 | tail -1 | tee -a $O/r2_time_k.log
done
for v in default qahead; do
  if [ $v = default ]; then unset PALACE_AMD_LIB; else export PALACE_AMD_LIB=$L/libpalace_amd_$v.so; fi
  TAG=$v DOFS=10e6 timeout 600 python scripts/time_pcg.py 2>&1 | tail -3 | tee -a $O/r2_time_pcg.log
done
