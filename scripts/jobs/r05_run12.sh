#!/bin/bash
# round 5, GPU call 14: push form of the dense E-vector (stores at their places in the dof-major copy list, contiguous gather): tests, A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest -q -x -m gpu tests/test_tet_gpu.py tests/test_dense_gpu.py tests/test_rt_gpu.py tests/test_2d_gpu.py tests/test_line_gpu.py tests/test_cpw_gpu.py tests/test_spheres_gpu.py tests/test_ams_gpu.py ) > gpurun_out/r12_tests.log 2>&1
echo "tests exit $?"; tail -5 gpurun_out/r12_tests.log | cut -c1-300
for push in 1 0 1 0; do
  echo "== PALACE_AMD_DENSE_PUSH=$push"
  ( PALACE_AMD_DENSE_PUSH=$push N=36 timeout 300 python scripts/time_tet.py ) 2>&1 | grep "mult" | cut -c1-120
done > gpurun_out/r12_push_ab.log 2>&1
cat gpurun_out/r12_push_ab.log
( PALACE_AMD_DENSE_PUSH=1 N=36 P=2 timeout 300 python scripts/time_tet.py ) 2>&1 | grep "mult" | cut -c1-120
