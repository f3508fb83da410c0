#!/bin/bash
set -u
REPO=$(pwd)
export PYTHONPATH=$REPO
mkdir -p gpurun_out
PALACE_AMD_GRAPH_DEBUG=1 timeout 900 python -m pytest tests/test_solvers_gpu.py tests/test_parity_r02_gpu.py tests/test_cpw_gpu.py tests/test_spheres_gpu.py tests/test_linalg_gpu.py -x -q > gpurun_out/r12_pytest.log 2>&1
grep -n "passed\|failed\|Error\|error" gpurun_out/r12_pytest.log | head -20
grep -c "palace_amd graph: recorded" gpurun_out/r12_pytest.log; grep "not recordable" gpurun_out/r12_pytest.log | sort | uniq -c | head
for cfg in "0 1" "1 1" "0 0"; do
  set -- $cfg
  echo "== PALACE_AMD_CG_HOST=$1 PALACE_AMD_GRAPH=$2"
  PALACE_AMD_GRAPH_DEBUG=1 PALACE_AMD_CG_HOST=$1 PALACE_AMD_GRAPH=$2 SLAB=8 timeout 300 python scripts/time_pcg.py 2>&1 | grep -v amdgpu.ids | head -20
done
cd /tmp && export TMPDIR=/tmp
PALACE_AMD_GRAPH=0 SLAB=8 ITS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_pcg_small -- python $REPO/scripts/time_pcg.py > $REPO/gpurun_out/prof_pcg_small.log 2>&1
cd $REPO
f=$(find gpurun_out/prof_pcg_small -name "*kernel_stats.csv" | head -1); head -30 $f | cut -c1-150
