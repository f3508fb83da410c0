"""Config 2: the eigenmode leg.  Part of bench.py (split in round 6; `python bench.py` is the entry point)."""
import json  # noqa: F401
import os  # noqa: F401
import sys  # noqa: F401
import time  # noqa: F401

import numpy as np  # noqa: F401

from .common import HBM_PEAK_GBS, ROOT, _rel, host_cores, oracle_hex_data  # noqa: F401


def eigen_leg(order=3, dofs=1.0e6, steps=30):
    """BASELINE config 2's shape on the device: the cylinder cavity (radius 2.74 cm, height 5.48 cm, eps_r = 2.08, PEC) at ~1M dofs,
    p = 3, shift-and-invert about the reference's target 2.0 GHz: each outer step is (K - sigma^2 M)^-1 M x by FGMRES + Hiptmair
    p-multigrid + native AMS (positive-shift preconditioner), M-orthogonalisation on the device; the outer iteration is a plain
    Lanczos loop on the host (palace_amd/fem/eigen.py; ARPACK / SLEPc are out of scope).  Reported: inner iterations/s, seconds per
    outer step, the lowest distinct frequencies against the analytic values of docs/src/examples/cylinder.md:113-123."""
    from palace_amd import linalg
    from palace_amd.fem.eigen import HexEigenSystem
    from palace_amd.fem.mesh import cylinder_for_dofs

    t0 = time.perf_counter()
    mesh = cylinder_for_dofs(dofs, order)
    ctx = linalg.Context()
    es = HexEigenSystem(ctx, mesh, order, 2.0, eps_r=2.08, L0=1.0e-2, tol=1.0e-8, max_it=200)
    setup = time.perf_counter() - t0
    res = es.lanczos(steps, nev=4, res_tol=1.0e-8)
    f = [float(v) for v in res["frequencies_ghz"]]
    distinct = []
    for v in f:
        if not distinct or abs(v - distinct[-1]) > 1e-4 * v:
            distinct.append(v)
    analytic = {"TM010": 2.903605, "TE111": 2.922212, "TM011": 3.468149}
    out = {"workload": f"cylinder cavity, {mesh.ne} hex27 elements, ND p={order}, {es.n} dofs, target 2.0 GHz, inner FGMRES to 1e-8 "
                       "(Hiptmair p-multigrid 1..p + native AMS on K + sigma^2 M), divergence-free start vector",
           "dofs": es.n, "setup_s": setup, "outer_steps": res["steps"], "seconds": res["seconds"],
           "seconds_per_outer_step": res["seconds"] / max(1, res["steps"]),
           "inner_iterations": res["inner_iterations"], "inner_iterations_per_solve": res["inner_iterations"] / max(1, res["inner_solves"]),
           "inner_iters_per_s": res["inner_iterations"] / max(1e-9, res["inner_seconds"]),
           "divfree_pcg_iterations": res["divfree_pcg_iterations"],
           "frequencies_ghz": f[:8], "residual_estimates": [float(v) for v in res["residual_estimates"][:8]],
           "lowest_distinct_ghz": distinct[:3],
           "analytic_ghz": analytic,
           "rel_err_vs_analytic": [abs(a - b) / b for a, b in zip(distinct[:3], analytic.values())],
           "rayleigh_quotient_0_rel_diff": abs(res.get("rayleigh_quotient_0", float("nan")) - res["lambda"][0]) / res["lambda"][0]}
    return out
