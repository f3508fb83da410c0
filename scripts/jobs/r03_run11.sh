#!/bin/bash
# round 3, GPU call 11: solver-layer fixes (initial residual, deferred failures, recording epoch, shared CSR, group abort),
# AMS / BoomerAMG through KspSolver, in-place contraction buffers for the p = 3 curl-curl kernel
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest -q -m gpu tests/test_solvers_gpu.py tests/test_assemble_gpu.py tests/test_cxx_host_gpu.py \
  tests/test_multirank_local_gpu.py tests/test_apply_gpu.py tests/test_fullsize_gpu.py tests/test_ams_gpu.py \
  tests/test_halo_gpu.py tests/test_rap_gpu.py tests/test_hiptmair_gpu.py > gpurun_out/r11_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r11_tests.log
tail -15 gpurun_out/r11_tests.log
timeout 300 python scripts/time_k.py > gpurun_out/r11_time_k.log 2>&1
tail -8 gpurun_out/r11_time_k.log
