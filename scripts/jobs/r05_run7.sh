#!/bin/bash
# round 5, GPU call 8: config 3 as the reference defines it -- test against port-S.csv on the reference's mesh and order, then the leg
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest -q -x -s -m gpu tests/test_cpw_gpu.py ) > gpurun_out/r7_tests.log 2>&1
echo "tests exit $?"; grep -v "^$" gpurun_out/r7_tests.log | tail -25 | cut -c1-600
( time timeout 900 python scripts/profile_cpw.py ) > gpurun_out/r7_cpw.log 2>&1
echo "cpw exit $?"; grep "^cpw:" gpurun_out/r7_cpw.log | cut -c1-3500; tail -5 gpurun_out/r7_cpw.log | cut -c1-300
