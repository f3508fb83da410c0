#!/bin/bash
set -u
export PYTHONPATH=$(pwd)
timeout 900 python -m pytest tests/test_h1_gpu.py tests/test_hiptmair_gpu.py tests/test_parity_r02_gpu.py tests/test_solvers_gpu.py tests/test_fullsize_gpu.py tests/test_cxx_host_gpu.py -x -q > gpurun_out/r32_pytest.log 2>&1
tail -4 gpurun_out/r32_pytest.log | cut -c1-300; grep -n "^E \|Fatal" gpurun_out/r32_pytest.log | head
DOFS=10e6 SLAB=1 timeout 300 python scripts/time_pcg.py 2>&1 | grep -v amdgpu.ids
PALACE_AMD_STREAM=0 DOFS=10e6 SLAB=1 ITS=20 timeout 300 python scripts/time_pcg.py 2>&1 | grep -v amdgpu.ids | tail -1
