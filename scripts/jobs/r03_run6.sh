#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_multirank_local_gpu.py tests/test_peer_gpu.py tests/test_tet_solvers_gpu.py tests/test_halo_gpu.py -x -q -m gpu > gpurun_out/r03_mr_t.log 2>&1; echo "rc=$?" >> gpurun_out/r03_mr_t.log
tail -30 gpurun_out/r03_mr_t.log
