"""Coarse-level solver alternatives for the p-multigrid PCG at the bench size: Jacobi-PCG (host syncs for the
dots) vs Chebyshev-Jacobi (no reductions)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import ceed, linalg
from palace_amd.fem.partition import SlabProblem
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, 10e6, levels=True)
mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
curl = ceed.coefficient_context(3)
fine = ceed.curlcurlmass_operator(prob.geom, prob.spaces[-1], mass, curl)
local = [fine.coarsen(prob.geom, s) for s in prob.spaces[:-1]] + [fine]
A = [linalg.ParOperator(ctx, op, e, linalg.DIAG_ONE) for op, e in zip(local, prob.ess)]
P = [linalg.Interp(ctx, prob.spaces[l], prob.spaces[l + 1]) for l in range(len(A) - 1)]
n = prob.n_true[-1]
ones = torch.ones(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(ones)
A[-1].mult(ones, b); b[torch.from_numpy(prob.ess[-1].astype(np.int64)).cuda()] = 0.0
for name, mk in (("cg8", lambda: linalg.cg(ctx, A[0], linalg.jacobi(ctx, A[0]), rel_tol=1e-2, max_it=8)),
                 ("cheb4", lambda: linalg.chebyshev(ctx, A[0], 4)), ("cheb8", lambda: linalg.chebyshev(ctx, A[0], 8)),
                 ("cheb12", lambda: linalg.chebyshev(ctx, A[0], 12)), ("cheb16", lambda: linalg.chebyshev(ctx, A[0], 16)),
                 ("cheb8x2", lambda: linalg.chebyshev(ctx, A[0], 8, smooth_it=2))):
    B = linalg.gmg(ctx, A, P, mk(), cheby_order=6)
    K = linalg.cg(ctx, A[-1], B, rel_tol=1e-8, max_it=600)
    x = torch.zeros_like(b); K.mult(b, x); x.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter(); K.mult(b, x); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = K.stats()
    print(f"{name:8s} iterations {st['iterations']:4d} time {dt:.3f} s  it/s {st['iterations']/dt:.1f} converged {st['converged']}", flush=True)
