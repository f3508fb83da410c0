#!/bin/bash
# round 3, GPU call 16: split-vector apply and the direct form of the multi-rank ParOperator::Mult
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu tests/test_split_gpu.py tests/test_peer_gpu.py tests/test_halo_gpu.py tests/test_apply_gpu.py > gpurun_out/r16_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r16_tests.log
grep -n "passed\|failed\|FAILED\|Error\|error" gpurun_out/r16_tests.log | tail -12
{
for d in 1 0; do
  PALACE_AMD_HALO_DIRECT=$d HALO_MODE=peer PCG=50 timeout 250 python scripts/time_halo_mult.py 2>&1 | grep "^\[\|rror" | sed "s/^/direct=$d /"
done
} | tee gpurun_out/r16_halo_direct.log
