"""Profile target: the p-coarsened curl-curl + mass operators (levels p = 1, 2 on the p = 3 quadrature data) of the
bench problem, timed with events; run under rocprofv3 --pmc for counters."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import ceed, linalg
from palace_amd.fem.partition import SlabProblem
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, float(os.environ.get("DOFS", "10e6")), levels=True)
mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
fine = ceed.curlcurlmass_operator(prob.geom, prob.spaces[-1], mass, ceed.coefficient_context(3))
for lvl in (1, 0):
    op = fine.coarsen(prob.geom, prob.spaces[lvl])
    n = prob.n_local[lvl]
    x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.zeros(n, dtype=torch.float64, device="cuda")
    for _ in range(3): op.mult(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    reps = int(os.environ.get("REPS", "10"))
    for _ in range(reps): op.mult(x, y)
    e1.record(); torch.cuda.synchronize()
    print(f"level p={prob.orders[lvl]} n={n} mult {e0.elapsed_time(e1)/reps:.4f} ms")
