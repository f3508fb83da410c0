#!/bin/bash
# Round 2, GPU call 5: counted waits / unconditional stores / non-temporal streams; occupancy variants; timeline; bench.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02_5
rm -rf $OUT && mkdir -p $OUT
export PYTHONPATH=$REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
L=$REPO/palace_amd/lib
for cfg in "libpalace_amd.so 1 0" "libpalace_amd.so 2 0" "libpalace_amd.so 0 0" "libpalace_amd_minw2.so 1 0" "libpalace_amd_minw2early.so 1 0" "libpalace_amd_minw2early.so 0 0" "libpalace_amd.so 1 5" "libpalace_amd.so 1 4"; do
  set -- $cfg
  echo "== $1 GPOS=$2 WG=$3" | tee -a $OUT/time_apply.log
  PALACE_AMD_LIB=$L/$1 PALACE_AMD_STREAM_GPOS=$2 PALACE_AMD_STREAM_WG=$3 timeout 300 python scripts/time_apply.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/time_apply.log
done
PALACE_AMD_LIB=$L/libpalace_amd_trace.so timeout 300 python scripts/trace_stream.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $OUT/trace.log
OP=curlmass PALACE_AMD_LIB=$L/libpalace_amd_trace.so timeout 300 python scripts/trace_stream.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/trace.log
cd /tmp && export TMPDIR=/tmp
for op in curl curlmass; do
  OP=$op REPS=10 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$op -- python $REPO/scripts/profile_apply.py > $OUT/prof_$op.log 2>&1
done
cd $REPO
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -4 $f | cut -c1-140; done
timeout 600 python bench.py --no-cpu --no-tets > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print(d["value"] / 1e9, "Gdof/s", d["ms_per_step"], "ms", d["roofline"]["kernel_ms"], {k: (v["iters_per_s"], v["iterations_to_1e-8"]) for k, v in d["pcg"].items() if isinstance(v, dict)})
except Exception as e:
    print("bench parse failed", e)
PY
