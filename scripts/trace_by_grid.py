"""Group a rocprofv3 kernel trace (…_kernel_trace.csv) by kernel name AND grid size: calls, average and total duration -- tells the levels
of a multigrid cycle apart (the same kernel runs on every level).  usage: trace_by_grid.py trace.csv [min_total_ms]"""
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
acc = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("palace::", "").replace("pa::", "").replace("void ", "").split("(")[0][:70], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))
    a = acc[k]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in acc.values())
print(f"total {tot / 1e6:.1f} ms")
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if a[1] / 1e6 >= min_ms:
        print(f"{k[0]:70s} grid {k[1]:9d} calls {a[0]:7d} avg {a[1] / a[0] / 1e3:8.1f} us total {a[1] / 1e6:8.1f} ms {100 * a[1] / tot:5.1f} %")
