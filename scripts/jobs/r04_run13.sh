#!/bin/bash
# round 4, GPU call 13: the library as committed (CSR form of the dense gather by default): dense-path tests and the default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 900 python -m pytest -q -x -m gpu tests/test_dense_gpu.py tests/test_tet_gpu.py tests/test_tet_solvers_gpu.py tests/test_spheres_gpu.py tests/test_cpw_gpu.py tests/test_2d_gpu.py tests/test_rt_gpu.py tests/test_split_gpu.py ) > $O/r13_tests.log 2>&1
echo "tests exit $?"; grep -E "passed|failed" $O/r13_tests.log | tail -1
( time timeout 1500 python bench.py ) > $O/r13_bench.json 2> $O/r13_bench.err
echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r13_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step")}, r["frac"], r["kernel_ms"], r["traffic"])
print({k: (round(v["iters_per_s"], 1), v.get("iterations_to_1e-8")) for k, v in d["pcg"].items() if isinstance(v, dict) and "iters_per_s" in v})
t = d["tets_mfma"]
print("tets", t["curlcurl"]["ms"], t["curlcurl_mass"]["ms"], "cpw", d["cpw"]["fgmres"], "spheres", d["spheres"]["p3"]["apply"]["ms"], d["spheres"]["p3"]["rel_dev_from_terminal_C_csv"])
PY
