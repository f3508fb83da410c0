#!/bin/bash
# round 5, GPU call 18: row-list gather of small blocks (surface terms), rank-thread test of the device Gram-Schmidt; cpw leg again
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest -q -x -m gpu tests/test_orthog_gpu.py tests/test_cpw_gpu.py tests/test_tet_gpu.py tests/test_2d_gpu.py tests/test_line_gpu.py tests/test_cxx_boundary_gpu.py tests/test_multirank_local_gpu.py ) > gpurun_out/r16_tests.log 2>&1
echo "tests exit $?"; tail -6 gpurun_out/r16_tests.log | cut -c1-400
( time timeout 900 python scripts/profile_cpw.py ) > gpurun_out/r16_cpw.log 2>&1
echo "cpw exit $?"; grep "^cpw:" gpurun_out/r16_cpw.log | head -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()[5:])
print({k: (d[k].get('ms') or (d[k].get('iterations_to_1e-8'), round(d[k].get('seconds', 0), 3), round(d[k].get('iters_per_s', 0), 2))) for k in ('complex_apply', 'fgmres', 'fgmres_host_driven_mgs', 'fgmres_second_solve', 'fgmres_cgs2')}, d['parity'])
"
