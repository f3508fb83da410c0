#!/bin/bash
# round 5, GPU call 6: dense (tetrahedral) kernel after the E^T / spill fixes: tests of every dense path, then timings
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest -q -x -m gpu tests/test_tet_gpu.py tests/test_dense_gpu.py tests/test_split_gpu.py tests/test_rt_gpu.py tests/test_2d_gpu.py "tests/test_complex_gpu.py::test_fused_complex_apply_tets" tests/test_cpw_gpu.py ) > gpurun_out/r5_tests.log 2>&1
echo "tests exit $?"; tail -8 gpurun_out/r5_tests.log | cut -c1-300
( N=36 timeout 300 python scripts/time_tet.py ) > gpurun_out/r5_tet.log 2>&1; grep "mult\|affine" gpurun_out/r5_tet.log | cut -c1-200
( N=36 P=2 timeout 300 python scripts/time_tet.py ) > gpurun_out/r5_tet_p2.log 2>&1; grep "mult\|setup" gpurun_out/r5_tet_p2.log | cut -c1-200
