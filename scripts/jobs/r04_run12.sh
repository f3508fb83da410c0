#!/bin/bash
# round 4, GPU call 12: run form of the dense path's E^T gather: the whole -m gpu suite, A/B of the tetrahedral applies against the
# CSR form (PALACE_AMD_DENSE_GATHER=csr), the default bench line on the final library
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 2400 python -m pytest -q -m gpu tests --durations=5 ) > $O/r12_tests.log 2>&1
echo "tests exit $?" >> $O/r12_tests.log; grep -E "passed|failed" $O/r12_tests.log | tail -2
for g in csr runs csr runs; do  # (call 12 ran with the run form as the default)
  PALACE_AMD_DENSE_GATHER=$g N=36 REPS=200 timeout 300 python scripts/time_tet.py 2>&1 | grep -E "mult" | tr '\n' ' ' | sed "s/^/[$g] /" | tee -a $O/r12_time_tet.log; echo | tee -a $O/r12_time_tet.log
done
( time timeout 1500 python bench.py ) > $O/r12_bench.json 2> $O/r12_bench.err
echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r12_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step")}, r["frac"], r["kernel_ms"], r["traffic"])
print({k: (round(v["iters_per_s"], 1), v.get("iterations_to_1e-8")) for k, v in d["pcg"].items() if isinstance(v, dict) and "iters_per_s" in v})
t = d["tets_mfma"]
print("tets", t["curlcurl"]["ms"], t["curlcurl"]["hbm_frac"], t["curlcurl_mass"]["ms"], t["complex"]["ms"], t["pcg_hiptmair_ams"], t["parity"]["rel_l2_y_full"])
print("cpw", d["cpw"]["complex_apply"]["ms"], d["cpw"]["fgmres"], d["cpw"]["parity"])
print("spheres", d["spheres"]["p3"]["rel_dev_from_terminal_C_csv"], d["spheres"]["p3"]["apply"])
PY
