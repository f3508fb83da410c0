#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_peer_gpu.py -x -q -m gpu > gpurun_out/r03_peer_t.log 2>&1; echo "peer rc=$?" >> gpurun_out/r03_peer_t.log
tail -3 gpurun_out/r03_peer_t.log
for m in rccl peer; do HALO_MODE=$m PCG=50 timeout 600 python scripts/time_halo_mult.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r03_halo_time.log 2>&1
cat gpurun_out/r03_halo_time.log
