"""CPU baseline for M2 ("the restated PCG + GMG on CPU"): PCG on K + M with the p-multigrid V-cycle
(levels p = 1, 2, 3; 4th-kind Chebyshev order 6) entirely through the oracle -- local applies by oracle/oracle_c.c
(dense tables, OpenMP), everything else numpy -- on a cylinder of about DOFS unknowns.  Prints one JSON line.
CPU only: python tests/tools/cpu_pcg_baseline.py [DOFS] [ITERS]
Lives under tests/: it runs the oracle (test infrastructure), which nothing outside tests/, smoke() and the cpu_baseline leg of bench.py may do."""
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

from oracle import palace_oracle as po
from palace_amd.fem.fespace import NDHexSpace
from palace_amd.fem.mesh import cylinder_for_dofs
from tests import util


def main():
    dofs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5e5
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    levels = [1, 2, 3]
    t0 = time.perf_counter()
    mesh = cylinder_for_dofs(dofs, 3)
    q1d = 4
    spaces = [NDHexSpace(mesh, p) for p in levels]
    geom = util.oracle_geom(mesh, q1d)
    cm, bm = util.make_ctx("scalar")
    cc, bc = util.make_ctx("identity")
    blob = np.concatenate([bm, bc])
    A = [util.FastParOperatorOracle(s, geom, "hdivmass", blob, s.ess_dofs(), q1d, cm, cc) for s in spaces]
    P = [po.InterpOracle(a.elem_dof_lex, a.elem_sign_lex, b.elem_dof_lex, b.elem_sign_lex, a.ndofs, b.ndofs,
                         po.nd_hex_interp_lex(a.p, b.p)) for a, b in zip(spaces[:-1], spaces[1:])]
    sm = [None] + [po.ChebyshevOracle(A[l], 6) for l in (1, 2)]
    d0 = 1.0 / A[0].diagonal()
    coarse = lambda r: po.pcg(A[0].mult, r, lambda v: d0 * v, rel_tol=1e-2, max_it=8)[0]  # noqa: E731
    B = po.GMGOracle(A, [(p.mult, p.mult_transpose) for p in P], sm, coarse, [s.ess_dofs() for s in spaces])
    n = spaces[-1].ndofs
    b = A[-1].mult(np.ones(n))
    b[spaces[-1].ess_dofs()] = 0.0
    t_setup = time.perf_counter() - t0
    x = np.random.default_rng(1).uniform(0, 1, n)
    A[-1].mult(x)
    t1 = time.perf_counter()
    napply = 5
    for _ in range(napply):
        A[-1].mult(x)
    t_apply = (time.perf_counter() - t1) / napply
    t2 = time.perf_counter()
    _, it, hist = po.pcg(A[-1].mult, b, B.mult, rel_tol=0.0, max_it=iters)
    t_pcg = time.perf_counter() - t2
    print(json.dumps({"what": "oracle PCG + p-multigrid (p=1,2,3; Chebyshev order 6; Jacobi-PCG(8) coarse) on CPU",
                      "dofs": n, "elements": mesh.ne, "threads": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count())),
                      "iterations": it, "seconds": t_pcg, "iters_per_s": it / t_pcg,
                      "curlcurlmass_apply_dof_per_s": n / t_apply, "rel_res": hist[-1] / hist[0] if hist else None,
                      "setup_s": t_setup}))


if __name__ == "__main__":
    main()
