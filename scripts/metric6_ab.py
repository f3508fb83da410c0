"""A / B of the experiment build -DPA_METRIC6 (the metric form's |detJ| / w recomputed as w^2 / det(H): 6 instead of 7 doubles per point
streamed by the K + M kernels): run once per library (PALACE_AMD_LIB), prints K + M apply times per level, the complex apply, PCG it/s and
the difference of the K + M result from the oracle-checked default library's (saved by the first run)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import ceed, linalg
from palace_amd.fem.partition import SlabProblem, strong_shape
import bench
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, 10.0e6, levels=True, shape=strong_shape(10.0e6, 3))
tag = os.environ.get("TAG", "default")
def tm(A, n, reps=50):
    x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    for _ in range(5): A.mult(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ctx.torch_stream):
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): A.mult(x, y)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
K, b, x = prob.pcg_gmg_solver(max_it=50, hiptmair=False, coarse="chebyshev")
K.mult(b, x); torch.cuda.synchronize(); t0 = time.perf_counter(); K.mult(b, x); torch.cuda.synchronize(); dt = time.perf_counter() - t0
us = [round(tm(A, s.ndofs), 1) for A, s in zip(prob.last_A[1:], prob.spaces[1:])]
g = torch.Generator(device="cuda"); g.manual_seed(5)
v = torch.rand(prob.spaces[-1].ndofs, dtype=torch.float64, device="cuda", generator=g); w = torch.empty_like(v)
prob.last_A[-1].mult(v, w)
ref = "/tmp/metric6_ref.pt"
diff = None
if tag == "default": torch.save(w.cpu(), ref)
elif os.path.exists(ref):
    r = torch.load(ref).cuda(); diff = float((w - r).norm() / r.norm())
cl = bench.complex_leg(ctx, prob, reps=50, parity=False)
print(f"[{tag}] PCG {K.stats()['iterations'] / dt:.1f} it/s; K+M ParOperator::Mult p2, p3: {us} us; complex apply {cl['ms']:.4f} ms; rel diff of K+M x from the default library: {diff}", flush=True)
