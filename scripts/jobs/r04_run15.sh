#!/bin/bash
# round 4, GPU call 15 (the last minutes of the budget): the QFunction families added last against the oracle, then the test
# files of every path the change touched (dense 2-D / line / RT blocks, two-space operators, estimators, boundary forms)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 200 python -m pytest tests/test_qf_rest_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tail -60 ) > gpurun_out/r15_new.log 2>&1
echo "new tests exit ${PIPESTATUS[0]}" >> gpurun_out/r15_new.log
tail -5 gpurun_out/r15_new.log
( time timeout 280 python -m pytest tests/test_2d_gpu.py tests/test_line_gpu.py tests/test_mixed_grad_gpu.py tests/test_rt_gpu.py \
    tests/test_estimator_gpu.py tests/test_cxx_estimator_gpu.py tests/test_cxx_boundary_gpu.py tests/test_dense_gpu.py \
    -q -m gpu --tb=short -x -p no:cacheprovider 2>&1 | tail -30 ) > gpurun_out/r15_regress.log 2>&1
tail -5 gpurun_out/r15_regress.log
