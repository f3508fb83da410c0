"""Counter target: five one-pass complex applies with anisotropic materials on the bench mesh (bench.py: complex_leg(aniso=True)'s
operator, no parity leg) and the calibration stream y = a x + b y (16 B read, 8 B written per entry), for
  rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python scripts/pmc_complex_aniso.py
(scripts/jobs/r05_run25.sh; the per-dispatch tables are reduced by scripts/pmc_complex_aniso.py --reduce <dir>)."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def reduce(root):
    out = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = {}
        for f in glob.glob(os.path.join(root, ctr, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                tag = ("elem" if "nd_hex_stream_kernel" in k else "gather" if "et_run_gather_kernel" in k else
                       "cal" if "OpAxpby" in k else None)
                if tag and row["Counter_Name"] == ctr:
                    sm, ids = acc.get(tag, (0.0, set()))
                    ids.add(row["Dispatch_Id"])
                    acc[tag] = (sm + float(row["Counter_Value"]), ids)
        for tag, (sm, ids) in acc.items():
            out[f"{tag}.{ctr}.KiB_per_dispatch"] = sm / max(1, len(ids))
            out[f"{tag}.{ctr}.dispatches"] = len(ids)
    return out


if len(sys.argv) > 2 and sys.argv[1] == "--reduce":
    print(json.dumps(reduce(sys.argv[2])))
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from palace_amd import ceed, linalg  # noqa: E402
from palace_amd.fem.partition import SlabProblem, strong_shape  # noqa: E402

ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, 10.0e6, levels=False, shape=strong_shape(10.0e6, 3))
nd = prob.spaces[-1]
c, s_ = np.cos(0.3), np.sin(0.3)
R = np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]]) @ np.array([[1.0, 0.0, 0.0], [0.0, c, -s_], [0.0, s_, c]])
eps = R @ np.diag([9.3, 9.3, 11.5]) @ R.T
loss = R @ np.diag([9.3 * 3.0e-5, 9.3 * 3.0e-5, 11.5 * 8.6e-5]) @ R.T
eps, loss = 0.5 * (eps + eps.T), 0.5 * (loss + loss.T)
if os.environ.get("ISO", "0") == "1":  # (round 6: the isotropic -- metric-form -- complex kernel, bench.py: complex_leg(aniso=False))
    eps, loss = np.array([2.08]), np.array([0.05 / 0.3])
Ar = ceed.curlcurlmass_operator(prob.geom, nd, ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[-0.3 * eps]), ceed.coefficient_context(3))
Ai = ceed.ndmass_operator(prob.geom, nd, ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[0.3 * loss]))
A = linalg.ComplexParOperator(ctx, Ar, Ai, prob.ess[-1], linalg.DIAG_ONE)
n = nd.ndofs
xr, xi = (torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(2))
yr, yi = torch.empty_like(xr), torch.empty_like(xr)
for _ in range(5):
    A.mult(xr, xi, yr, yi)
for _ in range(5):
    ctx.axpby(0.5, xr, 0.5, yr)
torch.cuda.synchronize()
print("done", n, int(ceed._lib.load().pa_op_complex_fused(Ar.handle, Ai.handle)))
