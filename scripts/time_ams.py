"""PCG + p-multigrid with the native AMS on level 0 against the stand-ins: iterations to 1e-8 and seconds.
  python scripts/time_ams.py [dofs]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem

dofs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0e6
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, dofs, levels=True)
print(f"{prob.mesh.ne} elements, level sizes {prob.n_true}", flush=True)
for hip in (True, False):
    for coarse in (("cg" if hip else "chebyshev"), "ams"):
        t0 = time.perf_counter()
        solver, b, x = prob.pcg_gmg_solver(max_it=400, rel_tol=1e-8, hiptmair=hip, coarse=coarse)
        torch.cuda.synchronize()
        t_setup = time.perf_counter() - t0
        solver.mult(b, x)  # warm-up (graph capture, work vectors)
        x.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.mult(b, x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = solver.stats()
        print(f"hiptmair={hip!s:5} coarse={coarse:9s}: {st['iterations']:4d} iterations to 1e-8, {dt:.3f} s "
              f"({st['iterations'] / dt:.1f} it/s), set-up {t_setup:.1f} s, converged {st['converged']}", flush=True)
        prob._keep.clear()
