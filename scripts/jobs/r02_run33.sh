#!/bin/bash
set -u
export PYTHONPATH=$(pwd)
for h1 in 1 0 1 0; do
  echo "== PALACE_AMD_STREAM_H1=$h1"
  PALACE_AMD_STREAM_H1=$h1 DOFS=10e6 SLAB=1 timeout 300 python scripts/time_pcg.py 2>&1 | grep -v amdgpu.ids | tail -2
done
cd /tmp && export TMPDIR=/tmp
DOFS=10e6 SLAB=1 ITS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_pcg_h1 -- python /root/repo/scripts/time_pcg.py > /root/repo/gpurun_out/prof_pcg_h1.log 2>&1
cd /root/repo
f=$(find gpurun_out/prof_pcg_h1 -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-130
