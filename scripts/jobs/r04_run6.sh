#!/bin/bash
# round 4, GPU call 6: the whole -m gpu suite (with durations) and the default bench line on the current library
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 2400 python -m pytest -q -m gpu tests --durations=25 ) > $O/r6_tests.log 2>&1
echo "tests exit $?" >> $O/r6_tests.log; grep -E "passed|failed" $O/r6_tests.log | tail -2
( time timeout 1500 python bench.py ) > $O/r6_bench.json 2> $O/r6_bench.err
echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step")}, r["frac"], r["kernel_ms"], r["traffic"], r.get("traffic_over_algorithmic"))
print({k: (round(v["iters_per_s"], 1), v.get("iterations_to_1e-8")) for k, v in d["pcg"].items() if isinstance(v, dict) and "iters_per_s" in v})
print("p4", d["p4"]["curlcurl"]["ms"], d["p4"]["curlcurl"]["hbm_frac"], d["p4"]["curlcurl_mass"]["ms"], "complex", d["complex"]["ms"], d["complex"]["hbm_frac"])
print("tets", d["tets_mfma"]["curlcurl"]["ms"], d["tets_mfma"]["curlcurl_mass"]["ms"], "cpw", d["cpw"]["fgmres"], "h1", d["h1"]["apply"])
PY
