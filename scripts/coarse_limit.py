"""What could an exact coarse solve buy?  PCG + p-multigrid iterations to 1e-8 with the level-0 problem solved (almost)
exactly by a long Jacobi-PCG, against the stand-ins of the bench (the upper bound for any AMS-class coarse solver).
  python scripts/coarse_limit.py [dofs]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem

dofs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0e6
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, dofs, levels=True)
print(f"{prob.mesh.ne} elements, {prob.n_true[-1]} dofs, level sizes {prob.n_true}", flush=True)
for hip in (False, True):
    for name, kw in (("bench stand-in", dict(coarse="cg" if hip else "chebyshev")),
                     ("cg 1e-2 / 8", dict(coarse="cg", coarse_tol=1e-2, coarse_max_it=8)),
                     ("cg 1e-2 / 50", dict(coarse="cg", coarse_tol=1e-2, coarse_max_it=50)),
                     ("cg 1e-4 / 500", dict(coarse="cg", coarse_tol=1e-4, coarse_max_it=500)),
                     ("cg 1e-10 / 3000", dict(coarse="cg", coarse_tol=1e-10, coarse_max_it=3000))):
        solver, b, x = prob.pcg_gmg_solver(max_it=500, rel_tol=1e-8, hiptmair=hip, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.mult(b, x)
        torch.cuda.synchronize()
        st = solver.stats()
        print(f"hiptmair={hip!s:5} coarse {name:16s}: {st['iterations']:4d} iterations, converged {st['converged']}, "
              f"{time.perf_counter() - t0:.2f} s", flush=True)
        prob._keep.clear()
