"""Reduce rocprofv3 --pmc counter_collection.csv files to one line per (kernel, grid size): dispatches, mean counter value (FETCH_SIZE /
WRITE_SIZE are KiB per dispatch as rocprofv3 reports them; on gfx950 FETCH_SIZE reports 0.5 of a 16-byte-lane stream: profiles/r06_apply_pmc.json)
and mean duration.  usage: pmc_by_kernel.py DIR_FETCH DIR_WRITE [min_calls] -> CSV on stdout"""
import csv, glob, os, sys, collections

def load(d):
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("palace::", "").replace("pa::", "").replace("void ", "").split("(")[0][:64]
            a = acc[(name, int(r["Grid_Size"]))]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return acc

fe, wr = load(sys.argv[1]), load(sys.argv[2])
min_calls = int(sys.argv[3]) if len(sys.argv) > 3 else 50
print("kernel,grid,dispatches,avg_us,fetch_MB_raw,write_MB,GBps_raw")
rows = []
for k, a in fe.items():
    if a[0] < min_calls:
        continue
    w = wr.get(k, [1, 0.0, 0.0])
    f_mb, w_mb, us = a[1] / a[0] * 1024 / 1e6, w[1] / max(w[0], 1) * 1024 / 1e6, a[2] / a[0]
    rows.append((a[2], f'"{k[0]}",{k[1]},{a[0]},{us:.1f},{f_mb:.2f},{w_mb:.2f},{(f_mb + w_mb) / us * 1e3:.0f}'))
for _, line in sorted(rows, reverse=True):
    print(line)
