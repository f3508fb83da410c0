#!/bin/bash
set -u
export PYTHONPATH=$(pwd)
mkdir -p gpurun_out
PALACE_AMD_GRAPH_DEBUG=1 timeout 600 python -m pytest tests/test_solvers_gpu.py -x -q > gpurun_out/r11_pytest.log 2>&1
grep -n "palace_amd graph\|passed\|failed\|Error\|error" gpurun_out/r11_pytest.log | head -40
grep -n "Fatal\|Segmentation\|File \"/root/repo" gpurun_out/r11_pytest.log | head -20
for cfg in "0 1" "1 1" "0 0"; do
  set -- $cfg
  echo "== PALACE_AMD_CG_HOST=$1 PALACE_AMD_GRAPH=$2"
  PALACE_AMD_GRAPH_DEBUG=1 PALACE_AMD_CG_HOST=$1 PALACE_AMD_GRAPH=$2 SLAB=8 timeout 300 python scripts/time_pcg.py 2>&1 | grep -v amdgpu.ids | head -20
done
