#!/bin/bash
# round 3: peer transport -- two processes on one GPU, rank threads; regression of the other multi-rank tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_peer_gpu.py -x -q -m gpu > gpurun_out/r03_peer_t.log 2>&1; echo "peer rc=$?" >> gpurun_out/r03_peer_t.log
tail -40 gpurun_out/r03_peer_t.log
timeout 900 python -m pytest tests/test_multirank_local_gpu.py tests/test_halo_gpu.py tests/test_ams_gpu.py -q -m gpu > gpurun_out/r03_peer_t2.log 2>&1; echo "regress rc=$?" >> gpurun_out/r03_peer_t2.log
tail -5 gpurun_out/r03_peer_t2.log
