#!/bin/bash
# round 4, GPU call 10: the fenced tier of the peer transport under the stress test and in the per-rank proxy; a second default bench
# line on another box
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 900 python -m pytest -q -x -m gpu tests/test_peer_gpu.py -k "fences or stress" ) > $O/r10_tests.log 2>&1
echo "tests exit $?"; tail -5 $O/r10_tests.log | cut -c1-300
for f in 0 1; do
  PALACE_AMD_PEER_FENCE=$f PCG=50 timeout 600 python scripts/time_halo_mult.py 2>&1 | grep -E "slab|it/s|PCG" | sed "s/^/[fence=$f] /" | tee -a $O/r10_halo_proxy.log
done
( time timeout 1500 python bench.py ) > $O/r10_bench.json 2> $O/r10_bench.err
echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r10_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step")}, r["frac"], r["kernel_ms"], r["traffic"])
print({k: (round(v["iters_per_s"], 1), v.get("iterations_to_1e-8")) for k, v in d["pcg"].items() if isinstance(v, dict) and "iters_per_s" in v})
print(d["p4"]["curlcurl"]["ms"], d["p4"]["curlcurl"]["hbm_frac"], d["cpw"]["fgmres"], d["tets_mfma"]["curlcurl"]["ms"])
PY
