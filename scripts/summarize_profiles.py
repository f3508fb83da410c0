"""Summarise gpurun_out/prof_* (written by scripts/profile_round.sh on the GPU box) into gpurun_out/profiles_new/<tag>_*
(run here, after the gpurun call: the summary is stamped with the commit the profile was taken at)."""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
TAG = os.environ.get("ROUND_TAG", "r03")
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

for d, name in (("prof_bench", "bench_kernel_stats.csv"), ("prof_curlmass", "apply_curlmass_kernel_stats.csv"),
                ("prof_p4_curl", "p4_kernel_stats.csv"), ("prof_p4_curlmass", "p4_curlmass_kernel_stats.csv")):
    f = find(d, "kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(PROF, f"{TAG}_{name}"))
f = find("prof_bench", "marker_api_trace.csv") or find("prof_bench", "marker_trace.csv")
if f:  # the roctx phase ranges of the bench run: name -> count, total ms
    acc = {}
    for row in csv.DictReader(open(f)):
        name = row.get("Function") or row.get("Name") or "?"
        try:
            dt = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
        except Exception:
            continue
        c, t = acc.get(name, (0, 0.0))
        acc[name] = (c + 1, t + dt)
    with open(os.path.join(PROF, f"{TAG}_bench_phase_ranges.csv"), "w") as g:
        g.write("range,count,total_ms\n")
        for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            g.write(f"{k},{c},{t:.3f}\n")
# p = 4: FETCH_SIZE / WRITE_SIZE of the five-point streaming kernel and its run gather, per dispatch
p4 = {}
for d in sorted(glob.glob(os.path.join(OUT, "prof_p4pmc_*"))):
    f = find(os.path.basename(d), "counter_collection.csv")
    if not f:
        continue
    op = "curlcurl_mass" if "curlmass" in os.path.basename(d) else "curlcurl"
    acc = {}
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        k = "nd_hex_stream5_kernel" if "nd_hex_stream5" in k else ("et_run_gather_kernel" if "et_run_gather" in k else None)
        if not k:
            continue
        key = (k, row["Counter_Name"])
        s_, n_ = acc.get(key, (0.0, set()))
        n_.add(row["Dispatch_Id"])
        acc[key] = (s_ + float(row["Counter_Value"]), n_)
    for (k, c), (s_, n_) in acc.items():
        p4.setdefault(op, {}).setdefault(k, {})[c] = s_ / max(1, len(n_))
if p4:
    for op, ks in p4.items():
        f_ = sum(v.get("FETCH_SIZE", 0.0) for v in ks.values()) * 1024
        w_ = sum(v.get("WRITE_SIZE", 0.0) for v in ks.values()) * 1024
        ks["per_apply_bytes"] = {"fetch_raw": f_, "fetch_x2": 2 * f_, "write_raw": w_, "traffic_x2": 2 * f_ + w_}
    json.dump({"note": "p = 4 (Q1 = 5) streaming kernel, bench.py's p4 mesh (52 920 elements, 10 263 720 dofs): rocprofv3 --pmc FETCH_SIZE / "
                       "WRITE_SIZE (KiB, one counter per run), averages per dispatch; fetch_x2 applies the gfx950 correction for 16-B/lane "
                       "streaming loads (MI355X_MICROARCH.md); algorithmic bytes (SURVEY 8d, G = 11): 52 920 x 12 500 + 16 x 10 263 720 = 825.7 MB",
               "commit": COMMIT, "kernels": p4}, open(os.path.join(PROF, f"{TAG}_p4_pmc.json"), "w"), indent=1)
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
