#!/bin/bash
# round 4, GPU call 9: the multi-rank C++ driver (KspSolver + AMS on a space with a halo) as two processes against one
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest -q -x -m gpu tests/test_cxx_host_gpu.py ) > gpurun_out/r9_tests.log 2>&1
echo "tests exit $?"; tail -30 gpurun_out/r9_tests.log | cut -c1-400
