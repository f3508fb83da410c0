#!/bin/bash
# round 5, GPU call 10: checkpoint -- the whole GPU suite, then the default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest -q -x -m gpu tests/ ) > gpurun_out/r8_tests.log 2>&1
echo "tests exit $?"; tail -12 gpurun_out/r8_tests.log | cut -c1-300
( time timeout 1200 python bench.py ) > gpurun_out/r8_bench.json 2> gpurun_out/r8_bench.err
echo "bench exit $?"; tail -3 gpurun_out/r8_bench.err | cut -c1-300; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r8_bench.json").read().strip().splitlines()[-1])
    def g(k, *ks):
        v = d.get(k)
        for q in ks:
            v = v.get(q) if isinstance(v, dict) else None
        return v
    print("value", d["value"], "ms", d["ms_per_step"], "blocks", d.get("timed_blocks"), "first", d.get("first_block_ms_per_step"))
    print("roofline", {k: d["roofline"][k] for k in ("frac", "kernel_ms", "traffic", "traffic_over_algorithmic") if k in d["roofline"]})
    print("cpu", d["cpu_baseline"])
    print("pcg", {k: (v.get("iters_per_s"), v.get("iterations_to_1e-8")) if isinstance(v, dict) else v for k, v in (d.get("pcg") or {}).items() if k != "config"})
    print("p4", {k: d["p4"].get(k) for k in ("curlcurl", "curlcurl_mass", "pcg_chebyshev", "config5_size_one_gpu", "error")})
    print("tets", {k: d["tets_mfma"].get(k) for k in ("curlcurl", "curlcurl_mass", "complex", "pcg_hiptmair", "pcg_hiptmair_ams", "error")})
    print("eigen", d.get("eigenmode"))
    print("cpw", {k: d["cpw"].get(k) for k in ("complex_apply", "fgmres", "fgmres_host_driven_mgs", "fgmres_second_solve", "fgmres_cgs2", "parity", "error")})
    print("cpw_iso", {k: d["cpw_iso"].get(k) for k in ("complex_apply", "fgmres", "parity", "error")})
    print("complex", d.get("complex")); print("h1", d.get("h1")); print("spheres", d.get("spheres")); print("mag", d.get("magnetostatic"))
except Exception as e:
    print("parse failed", e)
PY
