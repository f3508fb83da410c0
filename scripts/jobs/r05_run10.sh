#!/bin/bash
# round 5, GPU call 12: fused complex apply with the surface sub-operators applied after the pass (cpw reference operator)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest -q -x -s -m gpu tests/test_cpw_gpu.py tests/test_complex_gpu.py ) > gpurun_out/r10_tests.log 2>&1
echo "tests exit $?"; grep -v "^$" gpurun_out/r10_tests.log | tail -8 | cut -c1-600
( time timeout 900 python scripts/profile_cpw.py ) > gpurun_out/r10_cpw.log 2>&1
echo "cpw exit $?"; grep "^cpw:" gpurun_out/r10_cpw.log | head -1 | cut -c1-2500
