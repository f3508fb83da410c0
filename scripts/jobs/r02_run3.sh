#!/bin/bash
# Round 2, GPU call 3: where does the streaming kernel lose its time?  Timeline + ablations.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02_3
rm -rf $OUT && mkdir -p $OUT
export PYTHONPATH=$REPO
for v in trace static_trace; do
  echo "== variant $v" | tee -a $OUT/trace.log
  PALACE_AMD_LIB=$REPO/palace_amd/lib/libpalace_amd_$v.so timeout 300 python scripts/trace_stream.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/trace.log
done
echo "== variant trace curlmass" | tee -a $OUT/trace.log
OP=curlmass PALACE_AMD_LIB=$REPO/palace_amd/lib/libpalace_amd_trace.so timeout 300 python scripts/trace_stream.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/trace.log
echo "== variant static (no trace)" | tee -a $OUT/trace.log
PALACE_AMD_LIB=$REPO/palace_amd/lib/libpalace_amd_static.so timeout 300 python scripts/time_apply.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/trace.log
