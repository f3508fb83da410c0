#!/bin/bash
# round 5, GPU call 4: eigenmode loop (test on the reference mesh against eig.csv + the ~1M-dof leg), dense complex with fused
# essential dofs, per-rank proxy with the balanced grid A/B, cpw leg with the fence-free reductions
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest -q -x -m gpu tests/test_eigen_gpu.py tests/test_orthog_gpu.py "tests/test_complex_gpu.py" ) > gpurun_out/r3_tests.log 2>&1
echo "tests exit $?"; tail -12 gpurun_out/r3_tests.log | cut -c1-400
( time timeout 600 python scripts/time_eigen.py 3 1.0e6 30 ) > gpurun_out/r3_eigen.log 2>&1
echo "eigen exit $?"; grep "^eigen:" gpurun_out/r3_eigen.log | cut -c1-2500; tail -4 gpurun_out/r3_eigen.log | cut -c1-300
for bal in 0 3; do
  ( PALACE_AMD_STREAM_BALANCE=$bal PCG=50 timeout 300 python scripts/time_halo_mult.py 2>&1 | sed "s/^/[balance=$bal] /" ) >> gpurun_out/r3_proxy.log 2>&1
done
cat gpurun_out/r3_proxy.log | cut -c1-300
( time timeout 600 python scripts/profile_cpw.py ) > gpurun_out/r3_cpw.log 2>&1
echo "cpw exit $?"; grep "^cpw:" gpurun_out/r3_cpw.log | cut -c1-2200
