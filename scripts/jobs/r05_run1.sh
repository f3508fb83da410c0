#!/bin/bash
# round 5, GPU call 1: device-resident Gram-Schmidt (orthog.hip): its tests, the Krylov parity tests, then config 3's leg
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest -q -x -m gpu tests/test_orthog_gpu.py tests/test_parity_r02_gpu.py tests/test_complex_gpu.py tests/test_cpw_gpu.py tests/test_peer_gpu.py ) > gpurun_out/r1_tests.log 2>&1
echo "tests exit $?"; tail -15 gpurun_out/r1_tests.log | cut -c1-300
( time timeout 600 python scripts/profile_cpw.py ) > gpurun_out/r1_cpw.log 2>&1
echo "cpw exit $?"; grep "^cpw:" gpurun_out/r1_cpw.log | cut -c1-3000; tail -5 gpurun_out/r1_cpw.log | cut -c1-300
