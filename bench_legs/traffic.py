"""HBM-side traffic of one apply measured in the bench run (rocprofv3 --pmc child processes).  Part of bench.py (split in round 6; `python bench.py` is the entry point)."""
import json  # noqa: F401
import os  # noqa: F401
import sys  # noqa: F401
import time  # noqa: F401

import numpy as np  # noqa: F401

from .common import HBM_PEAK_GBS, ROOT, _rel, host_cores, oracle_hex_data  # noqa: F401


def measure_traffic(dofs, timeout=240):
    """HBM-side traffic of ONE curl-curl apply (element kernel + run gather), measured in this run: two `rocprofv3 --pmc` child
    processes (FETCH_SIZE, WRITE_SIZE: separate passes, counters only with --kernel-trace, as MI355X_MICROARCH.md prescribes)
    over scripts/profile_apply.py -- the bench mesh, 10 applies, then a calibration stream y = a x + b y with known bytes in the
    same process -- each counter divided by the fraction it reports of that known stream (the gfx950 FETCH_SIZE halving of
    16-byte-lane loads included).  Returns (bytes per apply or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals, n_cal = {}, None
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pa_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT, OP="curl", REPS="10", DOFS=str(dofs))
        try:
            p = subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.join(ROOT, "scripts", "profile_apply.py")], cwd="/tmp", env=env, capture_output=True, timeout=timeout)
            for ln in p.stdout.decode(errors="replace").splitlines():
                if ln.startswith("done"):
                    n_cal = int(ln.split()[1])
            acc = {}
            for fcsv in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(fcsv)):
                    k = row["Kernel_Name"]
                    tag = ("elem" if "nd_hex_stream_kernel" in k else "gather" if "et_run_gather_kernel" in k else
                           "cal" if ("OpAxpby" in k and ("k_ew<2" in k or "k_ew<(int)2" in k)) else None)
                    if tag and row["Counter_Name"] == ctr:
                        sm, ids = acc.get(tag, (0.0, set()))
                        ids.add(row["Dispatch_Id"])
                        acc[tag] = (sm + float(row["Counter_Value"]), ids)
            for tag, (sm, ids) in acc.items():
                vals[(tag, ctr)] = sm / max(1, len(ids)) * 1024.0  # (KiB per dispatch)
        except Exception as exc:  # noqa: BLE001
            return None, f"rocprofv3 --pmc {ctr} failed: {type(exc).__name__}: {exc}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    need = [("elem", "FETCH_SIZE"), ("gather", "FETCH_SIZE"), ("cal", "FETCH_SIZE"), ("elem", "WRITE_SIZE"), ("gather", "WRITE_SIZE"),
            ("cal", "WRITE_SIZE")]
    if n_cal is None or any(k not in vals for k in need):
        return None, "counter output incomplete: " + ", ".join(f"{a}.{b}" for a, b in need if (a, b) not in vals)
    rf = vals[("cal", "FETCH_SIZE")] / (16.0 * n_cal)
    rw = vals[("cal", "WRITE_SIZE")] / (8.0 * n_cal)
    traffic = (vals[("elem", "FETCH_SIZE")] + vals[("gather", "FETCH_SIZE")]) / rf + (vals[("elem", "WRITE_SIZE")] + vals[("gather", "WRITE_SIZE")]) / rw
    note = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes) over 10 applies on the bench mesh; per apply: "
            f"element kernel {vals[('elem', 'FETCH_SIZE')] / 1e6:.1f} MB fetched (raw) + {vals[('elem', 'WRITE_SIZE')] / 1e6:.1f} MB written, run gather "
            f"{vals[('gather', 'FETCH_SIZE')] / 1e6:.1f} + {vals[('gather', 'WRITE_SIZE')] / 1e6:.1f}; calibration on y = a x + b y over {n_cal} doubles "
            f"in the same process: FETCH_SIZE reports {rf:.3f} of the known read bytes, WRITE_SIZE {rw:.3f} of the written ones; raw counters divided by those")
    return traffic, note
