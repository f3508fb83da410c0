#!/bin/bash
# round 3, GPU call 17: three-term form of the Chebyshev smoothers: parity tests, PCG rates of both forms
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest -q -m gpu tests/test_solvers_gpu.py tests/test_parity_r02_gpu.py tests/test_hiptmair_gpu.py tests/test_tet_solvers_gpu.py tests/test_multirank_local_gpu.py tests/test_cxx_host_gpu.py tests/test_ams_gpu.py > gpurun_out/r17_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r17_tests.log
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r17_tests.log | tail -12
for f in e d; do
PALACE_AMD_CHEBY_FORM=$f timeout 600 python bench.py --no-cpu --no-tets --no-p4 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('form $f', d['value'], {k:(round(v['iters_per_s'],1), v.get('iterations_to_1e-8'), round(v.get('seconds_to_1e-8',0),3)) for k,v in d['pcg'].items() if isinstance(v,dict) and 'iters_per_s' in v})
"
done | tee gpurun_out/r17_cheby_forms.log
