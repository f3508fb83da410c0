#!/bin/bash
# Round 2, GPU call 8: the default bench line, then the rocprofv3 evidence of the round (scripts/profile_round.sh).
set -u
REPO=$(pwd)
mkdir -p gpurun_out
( time timeout 900 python bench.py ) > gpurun_out/r02_bench_n1.log 2>&1
grep '^{"metric"' gpurun_out/r02_bench_n1.log > gpurun_out/r02_bench_n1.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_bench_n1.json"))
print({k: d[k] for k in ("value", "ms_per_step", "scaling")})
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "kernel_ms", "traffic", "measured_stream_GBps")})
print("parity", d["parity"])
print("cpu", d["cpu_baseline"])
print("pcg", {k: (v if isinstance(v, str) else {a: v[a] for a in ("iters_per_s", "iterations_to_1e-8")}) for k, v in d["pcg"].items() if k != "config"})
print("p4", d["p4"])
print("tets", {k: v for k, v in d["tets_mfma"].items() if k != "workload"})
print("setup_s", d["setup_s"])
PY
tail -6 gpurun_out/r02_bench_n1.log | cut -c1-300
time bash scripts/profile_round.sh
