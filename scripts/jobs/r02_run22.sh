#!/bin/bash
set -u
export PYTHONPATH=$(pwd)
timeout 900 python -m pytest tests/test_apply_gpu.py tests/test_fullsize_gpu.py tests/test_solvers_gpu.py tests/test_hiptmair_gpu.py -x -q > gpurun_out/r22_pytest.log 2>&1
tail -4 gpurun_out/r22_pytest.log | cut -c1-300; grep -n "^E " gpurun_out/r22_pytest.log | head
timeout 300 python scripts/time_apply.py 2>&1 | grep -v amdgpu.ids | tail -4
PALACE_AMD_STREAM_GPOS=2 timeout 300 python scripts/time_apply.py 2>&1 | grep -v amdgpu.ids | tail -4
python bench.py --no-cpu --no-tets --no-p4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ms_per_step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'frac %.3f' % d['roofline']['frac'], {k: round(v['iters_per_s'],1) for k,v in d['pcg'].items() if k!='config'})"
