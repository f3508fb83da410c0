#!/bin/bash
# round 3, GPU call 15: complex form of the five-point kernel (tests + timing), spheres p-MG + AMG test
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest -q -m gpu tests/test_complex_gpu.py tests/test_spheres_gpu.py tests/test_stream5_gpu.py -s > gpurun_out/r15_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r15_tests.log
grep -n "spheres p=\|passed\|failed\|FAILED\|Error" gpurun_out/r15_tests.log | tail -12
timeout 600 python - > gpurun_out/r15_p4.log 2>&1 <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
from palace_amd import linalg
ctx = linalg.Context()
out = bench.p4_leg(ctx, 10.0e6, pcg_iters=0, parity=False)
print(json.dumps(out))
PY
tail -c 1500 gpurun_out/r15_p4.log
