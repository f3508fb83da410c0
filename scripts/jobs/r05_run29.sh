#!/bin/bash
# round 5, GPU call 29: the streaming kernels after the argument structs became value-initialised
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 170 python -m pytest tests/test_apply_gpu.py tests/test_split_gpu.py "tests/test_complex_gpu.py::test_fused_complex_apply" -x -q -m gpu ) 2>&1 | tail -6 | tee gpurun_out/r05_run29.log
