#!/bin/bash
# round 5, GPU call 21: the whole GPU suite on the library with the packed complex kernel and the distributed coarse solves
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -q -m gpu -x ) > gpurun_out/r05_gpu_tests_call21.log 2>&1
tail -8 gpurun_out/r05_gpu_tests_call21.log
