import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import ceed, linalg
from palace_amd.fem.partition import SlabProblem
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, float(os.environ.get("DOFS", "10e6")))
for cmax, ctol, hip in ((8, 1e-2, False), (8, 1e-2, True), (30, 1e-2, True)):
    K, b, x = prob.pcg_gmg_solver(max_it=300, rel_tol=1e-8, coarse_tol=ctol, coarse_max_it=cmax, hiptmair=hip)
    K.mult(b, x); torch.cuda.synchronize()
    t0 = time.perf_counter(); K.mult(b, x); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = K.stats()
    print(f"hiptmair={hip} coarse max_it {cmax:3d} tol {ctol:g}: {st['iterations']} its to 1e-8 in {dt:.3f} s  ({st['iterations']/dt:.1f} it/s) conv={st['converged']}", flush=True)
    prob._keep.clear()
