#!/bin/bash
# round 5, GPU call 16: geometry-from-nodes form as a tested option: its test, the streaming kernel tests, timing of both variants
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest -q -x -m gpu tests/test_apply_gpu.py tests/test_split_gpu.py tests/test_solvers_gpu.py ) > gpurun_out/r14_tests.log 2>&1
echo "tests exit $?"; tail -5 gpurun_out/r14_tests.log | cut -c1-300
( timeout 300 python scripts/geomn_ab.py 2>&1 | grep "GEOM" ) > gpurun_out/r14_geomn.log
for v in w2g1 w2g2; do
  ( PALACE_AMD_STREAM_GEOM=nodes PALACE_AMD_GEOMN_VARIANT=$v timeout 300 python scripts/geomn_ab.py 2>&1 | grep "GEOM\|rror" | tail -3 ) >> gpurun_out/r14_geomn.log
done
cat gpurun_out/r14_geomn.log | cut -c1-250
