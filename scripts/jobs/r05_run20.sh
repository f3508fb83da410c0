#!/bin/bash
# round 5, GPU call 20: the N-process rehearsal with the distributed level-0 leg (test sizes, then the bench size with 2 and 8 processes)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_rehearse_gpu.py -x -q -m gpu ) > gpurun_out/r05_run20_tests.log 2>&1
tail -6 gpurun_out/r05_run20_tests.log
( time timeout 600 python bench.py --rehearse 2 --no-cpu --no-nranks-legs ) > gpurun_out/r05_rehearse2_bench_size.json 2> gpurun_out/r05_rehearse2.err
tail -3 gpurun_out/r05_rehearse2.err
( time timeout 900 python bench.py --rehearse 8 --no-cpu --no-nranks-legs --pcg-iters 10 ) > gpurun_out/r05_rehearse8_bench_size.json 2> gpurun_out/r05_rehearse8.err
tail -3 gpurun_out/r05_rehearse8.err
python - <<'PY'
import json
for n in (2, 8):
    try:
        d = json.loads(open(f"gpurun_out/r05_rehearse{n}_bench_size.json").read().strip().splitlines()[-1])
        p = d["pcg"]
        print(n, {k: {kk: v for kk, v in p[k].items() if kk in ("iters_per_s", "iterations_to_1e-8", "replicated_level0", "distributed_level0", "error")} for k in p if k != "config"})
    except Exception as e:
        print(n, "failed", e)
PY
