#!/bin/bash
# round 4, GPU call 4: slot-pattern dictionary + header-walking run gather (correctness, A/B against the library of call 3),
# the replicated coarse solver assembled by the C++ layer (two processes)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 1800 python -m pytest -q -x -m gpu tests/test_apply_gpu.py tests/test_stream5_gpu.py tests/test_h1_gpu.py tests/test_complex_gpu.py tests/test_split_gpu.py tests/test_solvers_gpu.py tests/test_halo_gpu.py tests/test_fullsize_gpu.py tests/test_multirank_local_gpu.py tests/test_hiptmair_gpu.py tests/test_peer_gpu.py ) > $O/r4_tests.log 2>&1
echo "tests exit $?"; tail -6 $O/r4_tests.log
L=$PWD/palace_amd/lib
for v in base default base default; do
  if [ $v = default ]; then unset PALACE_AMD_LIB; else export PALACE_AMD_LIB=$L/libpalace_amd_$v.so; fi
  TAG=$v timeout 400 python scripts/time_k.py 2>&1 | tail -1 | tee -a $O/r4_time_k.log
done
unset PALACE_AMD_LIB
timeout 400 python scripts/price_evec_cache.py 2>&1 | tail -1 | tee $O/r4_price.log
