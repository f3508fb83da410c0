#!/bin/bash
# round 3, GPU call 20: the whole -m gpu suite and the default bench line on the final library
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest -q -m gpu tests > gpurun_out/r20_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r20_tests.log
tail -4 gpurun_out/r20_tests.log
timeout 1500 python bench.py > gpurun_out/r20_bench.json 2> gpurun_out/r20_bench.err
echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r20_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step")}, r["frac"], r["kernel_ms"], r["kernel_reps"], r["consistent_with_ms_per_step"])
print({k: (round(v["iters_per_s"], 1), v.get("iterations_to_1e-8")) for k, v in d["pcg"].items() if isinstance(v, dict) and "iters_per_s" in v})
for k in ("p4", "h1", "cpw", "tets_mfma", "complex"):
    print(k, "error" in json.dumps(d.get(k)), json.dumps(d.get(k))[:300])
PY
