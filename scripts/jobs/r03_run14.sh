#!/bin/bash
# round 3, GPU call 14: the whole -m gpu suite, then the round's rocprofv3 evidence (scripts/profile_round.sh)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest -q -m gpu tests > gpurun_out/r14_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r14_tests.log
tail -6 gpurun_out/r14_tests.log
timeout 1500 bash scripts/profile_round.sh
ls gpurun_out | head -40
tail -3 gpurun_out/prof_bench.log | cut -c 1-600
