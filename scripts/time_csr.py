"""Level-0 (order 1) assembled operator of the bench hierarchy: CSR SpMV time (ParOperator::Mult on the assembled level)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem, strong_shape
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, 10.0e6, levels=True, shape=strong_shape(10.0e6, 3))
K, b, x = prob.pcg_gmg_solver(max_it=50, hiptmair=False, coarse="chebyshev")
A0 = prob.last_A[0]
n = prob.spaces[0].ndofs
u, v = torch.rand(n, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda")
for _ in range(10): A0.mult(u, v)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(ctx.torch_stream):
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): A0.mult(u, v)
    e1.record(); torch.cuda.synchronize()
print(f"level-0 CSR ParOperator::Mult: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us ({n} rows)")
import time
K.mult(b, x); torch.cuda.synchronize(); t0 = time.perf_counter(); K.mult(b, x); torch.cuda.synchronize()
print(f"PCG chebyshev: {K.stats()['iterations'] / (time.perf_counter() - t0):.1f} it/s")
