#!/bin/bash
# round 5, GPU call 31: the two-process BOOMER_AMG test on a level 0 large enough for three algebraic levels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time PALACE_AMD_COARSE_VERBOSE=1 timeout 130 python -m pytest tests/test_cxx_host_gpu.py -q -m gpu -k distributed_amg -s ) 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r05_run31.log
