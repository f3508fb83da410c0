#!/bin/bash
# round 3, call 1: the five-point streaming kernel -- parity tests, then timings against the one-shot kernel
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_stream5_gpu.py -x -q -m gpu > gpurun_out/r03_t1.log 2>&1; echo "stream5 rc=$?" >> gpurun_out/r03_t1.log
timeout 600 python -m pytest tests/test_apply_gpu.py tests/test_halo_gpu.py tests/test_solvers_gpu.py -q -m gpu > gpurun_out/r03_t2.log 2>&1; echo "apply rc=$?" >> gpurun_out/r03_t2.log
timeout 600 python scripts/time_p4.py > gpurun_out/r03_p4.log 2>&1; echo "p4 rc=$?" >> gpurun_out/r03_p4.log
tail -5 gpurun_out/r03_t1.log; tail -3 gpurun_out/r03_t2.log; cat gpurun_out/r03_p4.log
