#!/bin/bash
# round 3, call: native AMG / AMS coarse solvers -- tests, then iteration counts and timings at the bench size
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ams_gpu.py -x -q -m gpu > gpurun_out/r03_ams_t.log 2>&1; echo "ams rc=$?" >> gpurun_out/r03_ams_t.log
tail -30 gpurun_out/r03_ams_t.log
timeout 900 python scripts/time_ams.py ${AMS_DOFS:-10.0e6} > gpurun_out/r03_ams_time.log 2>&1; echo "rc=$?" >> gpurun_out/r03_ams_time.log
cat gpurun_out/r03_ams_time.log
