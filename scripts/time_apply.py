import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import ceed, linalg
from palace_amd.fem.partition import SlabProblem
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, float(os.environ.get("DOFS", "10e6")), levels=False)
ops = {"curl": prob.local_curlcurl}
mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
ops["curlmass"] = ceed.curlcurlmass_operator(prob.geom, prob.spaces[-1], mass, ceed.coefficient_context(3))
ops["mass"] = ceed.ndmass_operator(prob.geom, prob.spaces[-1], mass)
n = prob.n_local[-1]
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.zeros(n, dtype=torch.float64, device="cuda")
for name, op in ops.items():
    for _ in range(3): op.add_mult(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): op.add_mult(x, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    for _ in range(3): op.mult(x, y)  # (the first launch of a kernel pays its one-time set-up)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): op.mult(x, y)
    e1.record(); torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1)/20
    print(f"{name:9s} add_mult {ms:.4f} ms  mult {ms2:.4f} ms  {op.algorithmic_bytes()/ms2/1e6:.0f} GB/s alg  {prob.n_true[-1]/ms2/1e6:.1f} Gdof/s (mult)")
