#!/bin/bash
# round 4, GPU call 5: run gather with chunk masks (thread per dof, no code words): correctness + A/B of element kernel and gather
# against the library of call 3 (base), alternating on one box
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 1500 python -m pytest -q -x -m gpu tests/test_apply_gpu.py tests/test_stream5_gpu.py tests/test_h1_gpu.py tests/test_complex_gpu.py tests/test_split_gpu.py tests/test_solvers_gpu.py tests/test_fullsize_gpu.py tests/test_halo_gpu.py ) > $O/r5_tests.log 2>&1
echo "tests exit $?"; tail -4 $O/r5_tests.log
L=$PWD/palace_amd/lib
for v in base default base default base default; do
  if [ $v = default ]; then unset PALACE_AMD_LIB; else export PALACE_AMD_LIB=$L/libpalace_amd_$v.so; fi
  timeout 400 python scripts/price_evec_cache.py 2>&1 | tail -1 | sed "s/^/[$v] /" | tee -a $O/r5_price.log
done
