#!/bin/bash
# round 3, GPU call 13: deferred-status test, cpw driven solver test, the cpw leg of bench.py alone
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest -q -m gpu tests/test_solvers_gpu.py tests/test_cpw_gpu.py -s > gpurun_out/r13_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r13_tests.log
grep -n "cpw p=\|passed\|failed\|FAILED\|Error" gpurun_out/r13_tests.log | tail -12
timeout 900 python - > gpurun_out/r13_cpw.log 2>&1 <<'PY'
import json, sys, time
sys.path.insert(0, ".")
import bench
t0 = time.time()
print(json.dumps(bench.cpw_leg(3, 1)))
print("leg seconds", time.time() - t0)
PY
tail -c 2500 gpurun_out/r13_cpw.log
