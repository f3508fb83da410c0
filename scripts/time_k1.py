import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import ceed, linalg
from palace_amd.fem.partition import SlabProblem
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, 10e6, levels=False)
op = prob.local_curlcurl
n = prob.n_local[-1]
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.zeros(n, dtype=torch.float64, device="cuda")
for _ in range(3): op.add_mult(x, y)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): op.mult(x, y)
e1.record(); torch.cuda.synchronize()
print(f"PA_DBG={os.environ.get('PA_DBG','0'):>3} curl mult (K1+K2) {e0.elapsed_time(e1)/20:.4f} ms")
