"""The p-coarsened levels on the fine level's quadrature data (the reference: CeedOperatorCoarsen, operator.cpp:528-546) against
every level on the rule of its own order (an option of SlabProblem.pcg_gmg_solver, NOT the reference's behaviour): PCG + p-multigrid
iterations/s over a fixed count, iterations to 1e-8, and the time of one K + M apply per level."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem, strong_shape
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, 10.0e6, levels=True, shape=strong_shape(10.0e6, 3))
def tm(A, n, reps=50):
    x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    for _ in range(5): A.mult(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ctx.torch_stream):
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): A.mult(x, y)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rule in ("fine", "own", "fine", "own"):
    for coarse in ("chebyshev", "ams"):
        K, b, x = prob.pcg_gmg_solver(max_it=50, hiptmair=False, coarse=coarse, level_rule=rule)
        K.mult(b, x); torch.cuda.synchronize()
        t0 = time.perf_counter(); K.mult(b, x); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        its = K.stats()["iterations"]
        apply_us = [round(tm(A, s.ndofs), 1) for A, s in zip(prob.last_A[1:], prob.spaces[1:])]
        prob._keep.clear()
        K, b, x = prob.pcg_gmg_solver(max_it=400, rel_tol=1e-8, hiptmair=False, coarse=coarse, level_rule=rule)
        K.mult(b, x); torch.cuda.synchronize()
        t0 = time.perf_counter(); K.mult(b, x); torch.cuda.synchronize(); dt8 = time.perf_counter() - t0
        st = K.stats()
        print(f"level_rule={rule:4s} coarse={coarse:9s}: {its / dt:6.1f} it/s; to 1e-8: {st['iterations']} iterations in {dt8:.3f} s; K+M ParOperator::Mult p2, p3: {apply_us} us", flush=True)
        prob._keep.clear()
