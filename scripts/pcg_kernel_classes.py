#!/usr/bin/env python3
"""Per-kernel-class table of ONE PCG + p-multigrid iteration at the bench size ("PCG bytes / iter ... achieved GB/s
per kernel class from rocprofv3").  Runs scripts/profile_pcg.py under rocprofv3 four times:
  --kernel-trace --stats            with ITS = a and ITS = b iterations   -> per-kernel calls and time per iteration = (b - a) difference
  --kernel-trace --pmc FETCH_SIZE   with ITS = a                          -> HBM-side bytes fetched per launch of every kernel
  --kernel-trace --pmc WRITE_SIZE   with ITS = a                          -> ... written
(counters in their own passes, as MI355X_MICROARCH.md prescribes; FETCH_SIZE x 2 for the 16-byte-lane streams is NOT applied here:
the column is the raw counter in bytes with its known gfx950 caveat, the calibrated figure for the headline pair is in bench.py).
usage: pcg_kernel_classes.py OUTDIR [HIP=0|1]   -> OUTDIR/pcg_kernel_classes.csv (+ the raw stats CSVs)"""
import csv, glob, os, re, shutil, subprocess, sys

out = os.path.abspath(sys.argv[1])
hip = sys.argv[2] if len(sys.argv) > 2 else "0"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
os.makedirs(out, exist_ok=True)
A, B = 10, 30


def run(tag, its, extra):
    d = os.path.join(out, tag)
    shutil.rmtree(d, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=root, ITS=str(its), HIP=hip)
    subprocess.run([exe, "--kernel-trace", *extra, "--output-format", "csv", "-d", d, "--", sys.executable,
                    os.path.join(root, "scripts", "profile_pcg.py")], cwd="/tmp", env=env, check=True, capture_output=True, timeout=1500)
    return d


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<.*?>)?)\(", name)
    n = m.group(1) if m else name.split("(")[0]
    return n.replace("palace::", "").replace("pa::", "")


def stats(d):
    res = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Name"])
            c, t = res.get(k, (0, 0.0))
            res[k] = (c + int(row["Calls"]), t + float(row["TotalDurationNs"]))
    return res


def pmc(d, ctr):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != ctr:
                continue
            k = short(row["Kernel_Name"])
            s, ids = acc.get(k, (0.0, set()))
            ids.add(row["Dispatch_Id"])
            acc[k] = (s + float(row["Counter_Value"]), ids)
    return {k: s / max(1, len(ids)) * 1024.0 for k, (s, ids) in acc.items()}  # KiB per dispatch -> bytes


sa, sb = stats(run("stats_a", A, ["--stats"])), stats(run("stats_b", B, ["--stats"]))
fetch = pmc(run("pmc_fetch", A, ["--pmc", "FETCH_SIZE"]), "FETCH_SIZE")
write = pmc(run("pmc_write", A, ["--pmc", "WRITE_SIZE"]), "WRITE_SIZE")
rows = []
for k, (cb, tb) in sb.items():
    ca, ta = sa.get(k, (0, 0.0))
    if cb - ca <= 0:
        continue
    calls = (cb - ca) / (B - A)
    us = (tb - ta) / (cb - ca) * 1e-3
    f, w = fetch.get(k), write.get(k)
    byt = (f or 0.0) + (w or 0.0)
    rows.append(dict(kernel=k, calls_per_iteration=round(calls, 2), avg_us=round(us, 2), us_per_iteration=round(calls * us, 1),
                     fetch_MB_per_launch=None if f is None else round(f / 1e6, 2), write_MB_per_launch=None if w is None else round(w / 1e6, 2),
                     GBps_raw_counters=None if not byt else round(byt / (us * 1e-6) / 1e9, 0)))
rows.sort(key=lambda r: -r["us_per_iteration"])
tot = sum(r["us_per_iteration"] for r in rows)
for r in rows:
    r["share"] = round(r["us_per_iteration"] / tot, 4)
with open(os.path.join(out, "pcg_kernel_classes.csv"), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
print(f"one PCG iteration (HIP={hip}): {tot:.0f} us of kernel time in {sum(r['calls_per_iteration'] for r in rows):.0f} launches")
for r in rows[:40]:
    print(r)
for t in ("stats_a", "stats_b", "pmc_fetch", "pmc_write"):  # keep the summaries, drop the raw traces
    for f in glob.glob(os.path.join(out, t, "**", "*"), recursive=True):
        if os.path.isfile(f) and not f.endswith("kernel_stats.csv"):
            os.remove(f)
