"""Summarise gpurun_out/prof_* (written by scripts/profile_round.sh on the GPU box) into gpurun_out/profiles_new/<tag>_*
(run here, after the gpurun call: the summary is stamped with the commit the profile was taken at)."""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
TAG = os.environ.get("ROUND_TAG", "r02")
try:
    COMMIT = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    COMMIT = None
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
        if "nd_hex_stream" in k:
            k = "nd_hex_stream_kernel"
        elif "et_run_gather" in k:
            k = "et_run_gather_kernel"
        elif "nd_hex_apply" in k:
            k = "nd_hex_apply_kernel"
        elif "et_gather" in k:
            k = "et_gather_kernel"
        elif "OpAxpby" in k:
            k = "calibration_axpby" if "k_ew<2" in k or "k_ew<(int)2" in k else "calibration_axpby_8B"
        else:
            continue
        key = (k, row["Counter_Name"])
        s, n = acc.get(key, (0.0, set()))
        n.add(row["Dispatch_Id"])
        acc[key] = (s + float(row["Counter_Value"]), n)
    for (k, c), (s, n) in acc.items():
        pmc.setdefault(k, {})[c] = s / max(1, len(n))
if pmc:
    note = ("rocprofv3 --kernel-trace --pmc, one counter group per run (scripts/profile_round.sh; the bench mesh, ND p=3 curl-curl "
            "apply); averages per dispatch. FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; on gfx950 FETCH_SIZE "
            "is exact only after the x2 correction for 16-B/lane streaming loads and uncalibrated for the 8-B/lane and "
            "gather loads of these kernels (MI355X_MICROARCH.md, HBM section), so both raw and x2 are given.")
    apply_k = {k: v for k, v in pmc.items() if not k.startswith("calibration")}
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
    cal8 = pmc.get("calibration_axpby_8B")
    if cal8 and n_cal:
        calib += (f"; the same stream with 8 B/lane accesses: FETCH_SIZE {cal8.get('FETCH_SIZE', 0) * 1024 / (16.0 * (n_cal - 1)):.3f}, "
                  f"WRITE_SIZE {cal8.get('WRITE_SIZE', 0) * 1024 / (8.0 * (n_cal - 1)):.3f} (the element kernel streams its q-data "
                  "and index words with 16 B/lane loads, its x gathers and y stores are 8 B/lane: the 16 B factor is applied)")
    json.dump({"note": note, "commit": COMMIT, "calibration": calib, "kernels": pmc,
               "per_apply_bytes": {"fetch_raw": tot_f, "fetch_x2": 2 * tot_f, "write_raw": tot_w,
                                   "traffic_raw": tot_f + tot_w,
                                   "traffic_corrected": (tot_f / rf + tot_w / rw) if rf and rw else None}},
              open(os.path.join(PROF, f"{TAG}_apply_pmc.json"), "w"), indent=1)
print(sorted(os.listdir(PROF)))
