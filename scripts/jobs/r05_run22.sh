#!/bin/bash
# round 5, GPU call 22: final evidence -- the whole GPU suite, the default bench line, the kernel trace of the packed complex apply
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu ) > gpurun_out/r05_gpu_tests_final.log 2>&1
tail -4 gpurun_out/r05_gpu_tests_final.log
( time timeout 600 python bench.py ) > gpurun_out/r05_bench_n1.json 2> gpurun_out/r05_bench_n1.err
tail -4 gpurun_out/r05_bench_n1.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_caniso -- python $GRAFT_REPO_ROOT/scripts/time_complex_aniso.py > $GRAFT_REPO_ROOT/gpurun_out/r05_complex_aniso_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r05_complex_aniso_rocprof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_caniso -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_complex_aniso_kernel_stats.csv && head -6 gpurun_out/r05_complex_aniso_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/prof_caniso
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_n1.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "steps", "timed_blocks") if k in d})
print("roofline", {k: d["roofline"][k] for k in ("frac", "achieved", "traffic") if k in d["roofline"]})
print("pcg", {k: (v.get("iters_per_s"), v.get("iterations_to_1e-8")) for k, v in d["pcg"].items() if isinstance(v, dict)})
for k in ("complex", "complex_aniso"):
    print(k, d.get(k))
print("tets", {k: d["tets_mfma"][k].get("ms") for k in ("curlcurl", "curlcurl_mass", "complex") if k in d["tets_mfma"]})
print("cpw", {k: d["cpw"].get(k) for k in ("fgmres", "s_parameters", "parity", "error") if k in d["cpw"]})
print("eigen", d.get("eigenmode"))
PY
