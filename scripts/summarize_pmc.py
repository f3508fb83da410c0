#!/usr/bin/env python3
"""Per-kernel averages of every counter collected under <dir>/prof_pmc*/ (rocprofv3 --pmc, one group per run)."""
import csv, glob, json, os, sys

out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "prof_pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        for tag in ("nd_hex_stream_kernel", "et_run_gather_kernel", "nd_hex_apply_kernel", "et_gather_kernel", "OpAxpby"):
            if tag in k:
                if tag == "OpAxpby":
                    tag = "axpby_16B" if "k_ew<2" in k else "axpby_8B"
                key = (tag, row["Counter_Name"])
                s, n = acc.get(key, (0.0, set()))
                n.add(row["Dispatch_Id"])
                acc[key] = (s + float(row["Counter_Value"]), n)
res = {}
for (k, c), (s, n) in sorted(acc.items()):
    res.setdefault(k, {})[c] = s / max(1, len(n))
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for k, v in res.items():
    print(k, {c: round(x, 1) for c, x in v.items()})
