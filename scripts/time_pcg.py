"""PCG + p-multigrid iterations/s on one GPU at a given size (DOFS, default 1.25e6 = the per-rank workload of the
8-GPU strong-scaling case): fixed iteration count, both smoother configurations of bench.py."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem, strong_shape
ctx = linalg.Context()
dofs = float(os.environ.get("DOFS", "1.25e6"))
its = int(os.environ.get("ITS", "50"))
if os.environ.get("SLAB"):  # rank 0's slab of the 10M-dof bench mesh cut into SLAB pieces
    n, nz = strong_shape(10e6, 3)
    prob = SlabProblem(ctx, 0, 1, 3, dofs, shape=(n, nz // int(os.environ["SLAB"])))
else:
    prob = SlabProblem(ctx, 0, 1, 3, dofs)
print("dofs", prob.n_true[-1], "elements", prob.mesh.ne)
for name, hip in (("chebyshev", False), ("hiptmair", True)):
    K, b, x = prob.pcg_gmg_solver(max_it=its, hiptmair=hip, coarse="cg" if hip else "chebyshev")
    K.mult(b, x)
    best = 0.0
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K.mult(b, x)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = max(best, K.stats()["iterations"] / dt)
    st = K.stats()
    print(f"{name}: {best:.1f} it/s ({st['iterations']} iterations, rel res {st['final_res'] / st['initial_res']:.3e})")
    prob._keep.clear()
