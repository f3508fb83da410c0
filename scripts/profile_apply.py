"""Profile target: a few launches of the fused ND apply kernels at the bench size (ORDER=3 default, ORDER=4: the five-point
streaming kernel on the bench's p = 4 mesh)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import ceed, linalg
from palace_amd.fem.partition import SlabProblem, strong_shape
ctx = linalg.Context()
dofs = float(os.environ.get("DOFS", "10e6"))
order = int(os.environ.get("ORDER", "3"))
if order == 3:
    prob = SlabProblem(ctx, 0, 1, 3, dofs, levels=False, shape=strong_shape(dofs, 3))  # the bench mesh (bench.py)
else:
    prob = SlabProblem(ctx, 0, 1, order, dofs, levels=False)  # bench.py's p4 leg
which = os.environ.get("OP", "curl")
if which == "curl":
    op = prob.local_curlcurl
else:
    mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
    op = ceed.curlcurlmass_operator(prob.geom, prob.spaces[-1], mass, ceed.coefficient_context(3))
n = prob.n_local[-1]
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.zeros(n, dtype=torch.float64, device="cuda")
for _ in range(int(os.environ.get("REPS", "10"))):
    op.mult(x, y)
# calibration stream for FETCH_SIZE / WRITE_SIZE: y = a x + b y reads 16 B and writes 8 B per entry
# with the same 8-byte-per-lane accesses as the apply kernels
for _ in range(5):
    ctx.axpby(0.5, x, 0.5, y)
if os.environ.get("CAL8"):  # the same stream with 8-byte lanes (views shifted by one entry: not 16-byte aligned)
    for _ in range(5):
        ctx.axpby(0.5, x[1:], 0.5, y[1:])
torch.cuda.synchronize()
print("done", n)
