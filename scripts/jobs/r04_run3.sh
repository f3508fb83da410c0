#!/bin/bash
# round 4, GPU call 3: the default bench line with the new legs (spheres, magnetostatic, complex parity, in-run traffic), the per-rank
# proxy of the 8-GPU case with the merged P^T kernel and the H1 split forms, tests of the areas touched since call 2
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 1500 python bench.py ) > $O/r3_bench.json 2> $O/r3_bench.err
echo "bench exit $?"; tail -c 600 $O/r3_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step")}, r["frac"], r["kernel_ms"], r["traffic"], str(r["traffic_note"])[:300])
print({k: (round(v["iters_per_s"], 1), v.get("iterations_to_1e-8")) for k, v in d["pcg"].items() if isinstance(v, dict) and "iters_per_s" in v})
for k in ("p4", "h1", "complex", "cpw", "spheres", "magnetostatic", "tets_mfma", "parity"):
    print(k, json.dumps(d.get(k))[:900])
PY
for m in 1 0; do
  PALACE_AMD_HALO_MERGED=$m PCG=50 timeout 600 python scripts/time_halo_mult.py 2>&1 | grep -E "slab|it/s|PCG" | sed "s/^/[merged=$m] /" | tee -a $O/r3_halo_proxy.log
done
( time timeout 1500 python -m pytest -q -x -m gpu tests/test_tet_gpu.py tests/test_tet_solvers_gpu.py tests/test_dense_gpu.py tests/test_halo_gpu.py tests/test_peer_gpu.py tests/test_spheres_gpu.py tests/test_rap_gpu.py tests/test_complex_gpu.py tests/test_multirank_local_gpu.py tests/test_split_gpu.py ) > $O/r3_tests.log 2>&1
echo "tests exit $?"; tail -4 $O/r3_tests.log
