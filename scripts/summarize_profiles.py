"""Summarise gpurun_out/prof_* (written by scripts/profile_round.sh) into profiles/r01_*."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
TAG = os.environ.get("ROUND_TAG", "r01")
PROF = os.path.join(OUT, "profiles_new")
os.makedirs(PROF, exist_ok=True)

def find(d, suffix):
    # gpurun merges every call's output into the same local directory: take the newest file
    f = glob.glob(os.path.join(OUT, d, "**", "*" + suffix), recursive=True)
    return max(f, key=os.path.getmtime) if f else None

for d, name in (("prof_bench", "bench_kernel_stats.csv"), ("prof_curlmass", "apply_curlmass_kernel_stats.csv")):
    f = find(d, "kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(PROF, f"{TAG}_{name}"))
log = os.path.join(OUT, "prof_bench.log")
if os.path.exists(log):
    for line in open(log):
        if line.startswith('{"metric"'):
            open(os.path.join(PROF, f"{TAG}_bench_under_rocprof.json"), "w").write(line)

pmc = {}
for d in sorted(glob.glob(os.path.join(OUT, "prof_pmc*"))):
    f = find(os.path.basename(d), "counter_collection.csv")
    if not f or not os.path.isdir(d):
        continue
    acc = {}
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "nd_hex_apply" in k:
            k = "nd_hex_apply_kernel"
        elif "et_gather" in k:
            k = "et_gather_kernel"
        elif "k_axpby" in k or "OpAxpby" in k:
            k = "calibration_axpby"
        else:
            continue
        key = (k, row["Counter_Name"])
        s, n = acc.get(key, (0.0, set()))
        n.add(row["Dispatch_Id"])
        acc[key] = (s + float(row["Counter_Value"]), n)
    for (k, c), (s, n) in acc.items():
        pmc.setdefault(k, {})[c] = s / max(1, len(n))
if pmc:
    note = ("rocprofv3 --kernel-trace --pmc, one counter group per run (scripts/profile_round.sh; 10M-dof ND p=3 curl-curl "
            "apply); averages per dispatch. FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; on gfx950 FETCH_SIZE "
            "is exact only after the x2 correction for 16-B/lane streaming loads and uncalibrated for the 8-B/lane and "
            "gather loads of these kernels (MI355X_MICROARCH.md, HBM section), so both raw and x2 are given.")
    apply_k = {k: v for k, v in pmc.items() if k != "calibration_axpby"}
    tot_f = sum(v.get("FETCH_SIZE", 0.0) for v in apply_k.values()) * 1024
    tot_w = sum(v.get("WRITE_SIZE", 0.0) for v in apply_k.values()) * 1024
    calib = "no calibration stream in this run"
    cal = pmc.get("calibration_axpby")
    n_cal = int(os.environ.get("CAL_N", "0"))
    rf = rw = None
    if cal and n_cal:
        rf = cal.get("FETCH_SIZE", 0) * 1024 / (16.0 * n_cal)
        rw = cal.get("WRITE_SIZE", 0) * 1024 / (8.0 * n_cal)
        calib = (f"calibration on y = a x + b y over {n_cal} doubles (16 B/lane loads, known 16 B read + 8 B written per "
                 f"entry): FETCH_SIZE reports {cal.get('FETCH_SIZE', 0) * 1024 / (16.0 * n_cal):.3f} of the read bytes, "
                 f"WRITE_SIZE {cal.get('WRITE_SIZE', 0) * 1024 / (8.0 * n_cal):.3f} of the written bytes")
    json.dump({"note": note, "calibration": calib, "kernels": pmc,
               "per_apply_bytes": {"fetch_raw": tot_f, "fetch_x2": 2 * tot_f, "write_raw": tot_w,
                                   "traffic_raw": tot_f + tot_w,
                                   "traffic_corrected": (tot_f / rf + tot_w / rw) if rf and rw else None}},
              open(os.path.join(PROF, f"{TAG}_apply_pmc.json"), "w"), indent=1)
print(sorted(os.listdir(PROF)))
