#!/bin/bash
# PMC passes over the coarsened curl-curl + mass kernels (separate runs, counters only with --kernel-trace;
# FETCH_SIZE and WRITE_SIZE cannot share a pass: rocprofv3 aborts).  Each pass sets the 10M-dof problem up with all
# levels (~70 s of box time).
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; rm -rf $OUT/prof_coarse*; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  ( cd $REPO; REPS=5 timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/prof_coarse$i -- python scripts/profile_coarse.py > $OUT/prof_coarse$i.log 2>&1 )
done
cd $REPO
python - <<'PY'
import csv, glob, json, os, re
OUT = os.path.join(os.getcwd(), "gpurun_out")
pmc = {}
for d in sorted(glob.glob(os.path.join(OUT, "prof_coarse*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            m = re.search(r"nd_hex_apply_kernel<(\d), (\d)", k)
            if m: k = f"nd_hex_apply<{m.group(1)},{m.group(2)}>"
            elif "et_gather" in k: k = "et_gather(" + row.get("Grid_Size", "?") + ")"
            else: continue
            key = (k, row["Counter_Name"]); s, n = acc.get(key, (0.0, set())); n.add(row["Dispatch_Id"]); acc[key] = (s + float(row["Counter_Value"]), n)
        for (k, c), (s, n) in acc.items(): pmc.setdefault(k, {})[c] = s / max(1, len(n))
json.dump(pmc, open(os.path.join(OUT, "prof_coarse_pmc.json"), "w"), indent=1)
for k, v in pmc.items(): print(k, {a: round(b) for a, b in v.items()})
PY
grep -h "level p" $OUT/prof_coarse1.log
