#!/bin/bash
# rocprofv3 evidence for the dense MFMA (tetrahedra) path: kernel stats + PMC passes.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; rm -rf $OUT/prof_tet $OUT/prof_tet_pmc*; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
cd $REPO
run() { (cd /tmp && N=${N:-36} REPS=5 timeout 300 rocprofv3 "$@" -- python $REPO/scripts/time_tet.py) ; }
( cd $REPO; N=${N:-36} REPS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tet -- python scripts/time_tet.py > $OUT/prof_tet.log 2>&1 )
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd $REPO; N=${N:-36} REPS=5 timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/prof_tet_pmc$i -- python scripts/time_tet.py > $OUT/prof_tet_pmc$i.log 2>&1 )
done
python - <<'PY'
import csv, glob, json, os
OUT = os.path.join(os.getcwd(), "gpurun_out")
f = glob.glob(os.path.join(OUT, "prof_tet", "**", "*kernel_stats.csv"), recursive=True)
if f:
    for l in open(f[0]).read().splitlines()[:6]: print(l[:200])
pmc = {}
for d in sorted(glob.glob(os.path.join(OUT, "prof_tet_pmc*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "dense_apply" in k:
                import re
                k = re.search(r"dense_apply\w*<[^>]*>", k).group(0)
            elif "et_gather" in k: k = "et_gather"
            else: continue
            key = (k, row["Counter_Name"]); s, n = acc.get(key, (0.0, set())); n.add(row["Dispatch_Id"]); acc[key] = (s + float(row["Counter_Value"]), n)
        for (k, c), (s, n) in acc.items(): pmc.setdefault(k, {})[c] = s / max(1, len(n))
json.dump(pmc, open(os.path.join(OUT, "prof_tet_pmc.json"), "w"), indent=1)
for k, v in pmc.items(): print(k, {a: round(b) for a, b in v.items()})
PY
