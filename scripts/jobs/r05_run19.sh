#!/bin/bash
# round 5, GPU call 19: the algebraic coarse solves distributed over the ranks (amg_dist.hpp), first run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_cxx_host_gpu.py tests/test_ams_gpu.py tests/test_multirank_local_gpu.py tests/test_peer_gpu.py tests/test_rehearse_gpu.py -x -q -m gpu > gpurun_out/r05_run19_tests.log 2>&1
tail -30 gpurun_out/r05_run19_tests.log
