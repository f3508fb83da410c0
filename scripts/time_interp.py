"""p-prolongations, discrete gradients and their transposes at the bench size (HIP events): the transfer kernels of one V-cycle."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem, SlabH1Space
ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, float(os.environ.get("DOFS", "10e6")), levels=True)
h1s = [SlabH1Space(prob.mesh, q, 0, 1, 0.0, prob.height, prob.radius) for q in prob.orders]
ops = {}
for l in range(len(prob.spaces) - 1):
    ops[f"P  nd p{prob.orders[l]}->p{prob.orders[l + 1]}"] = (linalg.Interp(ctx, prob.spaces[l], prob.spaces[l + 1]), prob.spaces[l].ndofs, prob.spaces[l + 1].ndofs)
for h, n in zip(h1s[1:], prob.spaces[1:]):
    ops[f"G  h1->nd p{n.p}"] = (linalg.Gradient(ctx, h, n), h.ndofs, n.ndofs)
def tm(f, reps=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ctx.torch_stream):
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, (op, nc, nf) in ops.items():
    xc, xf = torch.rand(nc, dtype=torch.float64, device="cuda"), torch.rand(nf, dtype=torch.float64, device="cuda")
    yc, yf = torch.empty_like(xc), torch.empty_like(xf)
    print(f"{name}: forward {tm(lambda: op.mult(xc, yf)):7.1f} us   transpose (kernel + gather) {tm(lambda: op.mult_transpose(xf, yc)):7.1f} us   ({nc} -> {nf} dofs)", flush=True)
