#!/bin/bash
# round 5, GPU call 17: one-pass complex apply on meshes with affine AND curved blocks (list form of the complex kernel)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest -q -x -m gpu tests/test_complex_gpu.py tests/test_tet_gpu.py ) > gpurun_out/r15_tests.log 2>&1
echo "tests exit $?"; tail -6 gpurun_out/r15_tests.log | cut -c1-400
