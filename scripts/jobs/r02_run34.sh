#!/bin/bash
set -u
export PYTHONPATH=$(pwd)
timeout 900 python -m pytest tests/test_h1_gpu.py tests/test_hiptmair_gpu.py tests/test_parity_r02_gpu.py tests/test_solvers_gpu.py tests/test_cxx_host_gpu.py tests/test_apply_gpu.py -x -q > gpurun_out/r34_pytest.log 2>&1
tail -4 gpurun_out/r34_pytest.log | cut -c1-300; grep -n "^E \|Fatal" gpurun_out/r34_pytest.log | head
for h1 in 1 0 1 0; do
  echo "== PALACE_AMD_STREAM_H1=$h1"
  PALACE_AMD_STREAM_H1=$h1 DOFS=10e6 SLAB=1 timeout 300 python scripts/time_pcg.py 2>&1 | grep -v amdgpu.ids | tail -1
done
cd /tmp && export TMPDIR=/tmp
DOFS=10e6 SLAB=1 ITS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_pcg_h1b -- python /root/repo/scripts/time_pcg.py > /root/repo/gpurun_out/prof_pcg_h1b.log 2>&1
cd /root/repo
f=$(find gpurun_out/prof_pcg_h1b -name "*kernel_stats.csv" | head -1); grep -i "h1_hex\|et_gather_kernel\|run_gather" $f | cut -c1-150
