#!/bin/bash
# Round 2, GPU call 10: device-resident PCG scalars + HIP graphs: tests, then it/s against the host loop / no graphs.
set -u
export PYTHONPATH=$(pwd)
timeout 900 python -m pytest tests/test_solvers_gpu.py tests/test_parity_r02_gpu.py tests/test_cpw_gpu.py tests/test_spheres_gpu.py -x -q 2>&1 | tail -15
for cfg in "0 1" "1 1" "0 0"; do
  set -- $cfg
  echo "== PALACE_AMD_CG_HOST=$1 PALACE_AMD_GRAPH=$2"
  PALACE_AMD_CG_HOST=$1 PALACE_AMD_GRAPH=$2 SLAB=8 timeout 300 python scripts/time_pcg.py 2>&1 | grep -v amdgpu.ids
done
echo "== 10M default"; DOFS=10e6 SLAB=1 timeout 300 python scripts/time_pcg.py 2>&1 | grep -v amdgpu.ids
