"""Event timings (solver's stream, 50 warm-up + 300 timed applies) of the p = 3 operators at the bench size: ParOperator K (the
headline), local K, K + M, M and the p-coarsened K + M.  PALACE_AMD_LIB selects an alternative build of the library.
  python scripts/time_k.py [dofs]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from palace_amd import ceed, linalg
from palace_amd.fem.partition import SlabProblem, strong_shape

dofs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0e6
ctx = linalg.Context()
n_cross, nz = strong_shape(dofs, 3)
prob = SlabProblem(ctx, 0, 1, 3, dofs, levels=True, shape=(n_cross, nz))
nd = prob.spaces[-1]
mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
KM = ceed.curlcurlmass_operator(prob.geom, nd, mass, ceed.coefficient_context(3))
M = ceed.ndmass_operator(prob.geom, nd, mass)
KM2 = KM.coarsen(prob.geom, prob.spaces[-2])
K = prob.curlcurl_par_operator()
tag = os.environ.get("TAG", os.environ.get("PALACE_AMD_LIB", "default"))
line = [f"[{tag}]"]
for name, op, n in (("ParOp K", K, nd.ndofs), ("K", prob.local_curlcurl, nd.ndofs), ("K+M", KM, nd.ndofs), ("M", M, nd.ndofs),
                    ("K+M p2", KM2, prob.spaces[-2].ndofs)):
    x = torch.rand(n, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    best = 1e9
    for rep in range(2):
        with torch.cuda.stream(ctx.torch_stream):
            for _ in range(50):
                op.mult(x, y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                op.mult(x, y)
            e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 300)
    line.append(f"{name} {best * 1e3:.1f}")
print("  ".join(line) + "  (us)", flush=True)
