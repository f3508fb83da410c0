#!/bin/bash
# Round 2, GPU call 1: parity of the streaming kernel, A/B timing against the one-shot kernel, kernel trace, stream probe.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02_1
rm -rf $OUT && mkdir -p $OUT
export PYTHONPATH=$REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
for cfg in "1 0" "0 0" "1 4" "1 5" "1 8"; do
  set -- $cfg
  echo "== STREAM=$1 WG=$2" >> $OUT/time_apply.log
  PALACE_AMD_STREAM=$1 PALACE_AMD_STREAM_WG=$2 timeout 300 python scripts/time_apply.py >> $OUT/time_apply.log 2>&1
done
cat $OUT/time_apply.log | grep -v Warning
timeout 120 scripts/probes/stream_probe > $OUT/stream_probe.log 2>&1
cat $OUT/stream_probe.log
cd /tmp && export TMPDIR=/tmp
for st in 1 0; do
  PALACE_AMD_STREAM=$st OP=curl REPS=10 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_curl_s$st -- python $REPO/scripts/profile_apply.py > $OUT/prof_curl_s$st.log 2>&1
  PALACE_AMD_STREAM=$st OP=curlmass REPS=10 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_curlmass_s$st -- python $REPO/scripts/profile_apply.py > $OUT/prof_curlmass_s$st.log 2>&1
done
rocprofv3 -L 2>/dev/null | grep -i "TCC_EA0_RD\|TCC_EA0_WR\|TCC_REQ\|TCC_HIT\|TCC_MISS\|FETCH_SIZE\|WRITE_SIZE\|TCC_BUBBLE\|MALL" | head -60 > $OUT/counters.txt
cd $REPO
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f | cut -c1-200; done
for st in 1 0; do
  PALACE_AMD_STREAM=$st timeout 600 python bench.py --no-cpu --no-tets > $OUT/bench_s$st.json 2> $OUT/bench_s$st.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_s$st.json").read().strip().splitlines()[-1])
    print("STREAM=$st", d["value"] / 1e9, "Gdof/s", d["ms_per_step"], "ms", {k: (v["iters_per_s"], v["iterations_to_1e-8"]) for k, v in d["pcg"].items() if isinstance(v, dict)})
except Exception as e:
    print("bench parse failed", e)
PY
done
