#!/bin/bash
# round 3, GPU call 18: the whole -m gpu suite, the default bench line, the round's rocprofv3 evidence
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest -q -m gpu tests > gpurun_out/r18_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r18_tests.log
tail -4 gpurun_out/r18_tests.log
timeout 1500 python bench.py > gpurun_out/r18_bench.json 2> gpurun_out/r18_bench.err
echo "bench exit $?"
timeout 1500 bash scripts/profile_round.sh
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r18_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"].get("consistent_with_ms_per_step"))
for k in ("pcg", "p4", "h1", "cpw", "complex", "tets_mfma"):
    print(k, json.dumps(d.get(k))[:900])
PY
