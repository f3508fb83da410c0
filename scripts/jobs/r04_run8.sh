#!/bin/bash
# round 4, GPU call 8 (final evidence on the final library): the whole -m gpu suite, the default bench line, rocprofv3 kernel tables
# (bench, K + M apply, order 4, tetrahedra), PMC passes of the headline pair and of the tetrahedral kernels
cd "$GRAFT_REPO_ROOT"
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
( time timeout 2400 python -m pytest -q -m gpu tests --durations=8 ) > $O/r8_tests.log 2>&1
echo "tests exit $?" >> $O/r8_tests.log; grep -E "passed|failed" $O/r8_tests.log | tail -2
( time timeout 1500 python bench.py ) > $O/r8_bench.json 2> $O/r8_bench.err
echo "bench exit $?"
rm -rf $O/prof_bench $O/prof_curlmass $O/prof_pmc* $O/prof_p4* $O/prof_tet*
cd /tmp
export PYTHONPATH=$REPO
timeout 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $REPO/$O/prof_bench -- python $REPO/bench.py --no-cpu --no-tets --no-p4 --no-traffic > $REPO/$O/prof_bench.log 2>&1
OP=curlmass REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof_curlmass -- python $REPO/scripts/profile_apply.py > $REPO/$O/prof_curlmass.log 2>&1
ORDER=4 OP=curl REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof_p4_curl -- python $REPO/scripts/profile_apply.py > $REPO/$O/prof_p4_curl.log 2>&1
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  OP=curl REPS=10 CAL8=1 timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $REPO/$O/prof_pmc$i -- python $REPO/scripts/profile_apply.py > $REPO/$O/prof_pmc$i.log 2>&1
done
grep -h '^done' $REPO/$O/prof_pmc1.log | awk '{print $2}' > $REPO/$O/prof_cal_n.txt
cd $REPO
N=36 REPS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tet -- python scripts/time_tet.py > $O/prof_tet.log 2>&1
N=36 REPS=5 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $O/prof_tet_pmc1 -- python scripts/time_tet.py > $O/prof_tet_pmc1.log 2>&1
find $O/prof_bench $O/prof_curlmass $O/prof_p4_curl $O/prof_tet $O/prof_tet_pmc1 $O/prof_pmc1 $O/prof_pmc2 $O/prof_pmc3 $O/prof_pmc4 $O/prof_pmc5 -type f ! -name '*stats*' ! -name '*counter_collection*' ! -name '*marker*' -delete 2>/dev/null
du -sh $O; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r8_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step")}, r["frac"], r["kernel_ms"], r["traffic"])
print({k: (round(v["iters_per_s"], 1), v.get("iterations_to_1e-8")) for k, v in d["pcg"].items() if isinstance(v, dict) and "iters_per_s" in v})
PY
