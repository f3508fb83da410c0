#!/bin/bash
# round 3, GPU call 12: the tests that failed in call 11 + a full bench.py run with the new legs (h1, tets parity)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu tests/test_solvers_gpu.py tests/test_cxx_host_gpu.py > gpurun_out/r12_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r12_tests.log
tail -12 gpurun_out/r12_tests.log
timeout 1500 python bench.py > gpurun_out/r12_bench.json 2> gpurun_out/r12_bench.err
echo "bench exit $?"
tail -c 3000 gpurun_out/r12_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r12_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"])
for k in ("pcg", "p4", "h1", "complex", "tets_mfma", "parity", "cpu_baseline"):
    print(k, json.dumps(d.get(k))[:1500])
PY
