#!/bin/bash
# round 5, GPU call 23: the set-up gather over the peer transport (Comm::AllGatherVHost), per-device transport settings, the
# plan-churn test repeated after the descriptor fix
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_peer_gpu.py tests/test_cxx_host_gpu.py tests/test_rehearse_gpu.py -q -m gpu -x ) > gpurun_out/r05_run23_tests.log 2>&1
tail -4 gpurun_out/r05_run23_tests.log
for i in 1 2 3; do timeout 200 python -m pytest tests/test_peer_gpu.py -q -m gpu -k "fences or stress" 2>&1 | tail -1; done | tee gpurun_out/r05_run23_churn.log
