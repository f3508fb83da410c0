"""Complex operator apply on tetrahedra (BASELINE config 3's shape: A = (K - w^2 eps M) + i w sigma M, ND p=3) through
ComplexParOperator::Mult: the one-pass dense form (pa_op_mult_complex, kind 2) vs PALACE_AMD_COMPLEX_FUSED=0 (four applies)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import ceed, linalg
from palace_amd.fem import tet
n = int(os.environ.get("N", "36")); p = int(os.environ.get("P", "3")); reps = int(os.environ.get("REPS", "100"))
ctx = linalg.Context()
mesh = tet.cube_tet_mesh(n)
if os.environ.get("CURVED", "0") == "1":  # quadratic geometry, every element warped: the general (per-point D) kernels
    mesh = tet.to_quadratic(mesh, lambda X: np.stack([X[:, 0] + 0.01 * np.sin(2 * X[:, 1] + X[:, 2]), X[:, 1] + 0.012 * X[:, 0] * X[:, 2],
                                                     X[:, 2] - 0.008 * np.cos(3 * X[:, 0]) * X[:, 1]], axis=1))
nd = tet.NDTetSpace(mesh, p)
pts, wts = tet.default_tet_rule(p)
interp, curl = nd.elem.tables(pts)
geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, **kw)
ident = ceed.coefficient_context(3)
mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([-2.08 * 0.3])])
cond = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([0.05])])
N = nd.ndofs
Ar = ceed.Operator(N, N).add_dense_integrator(geom, block, ceed.QF_HDIVMASS_33, np.concatenate([mass, ident]), ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize()
Ai = ceed.Operator(N, N).add_dense_integrator(geom, block, ceed.QF_HCURL_33, cond, ceed.EVAL_INTERP).finalize()
A = linalg.ComplexParOperator(ctx, Ar, Ai)
xr, xi = (torch.rand(N, dtype=torch.float64, device="cuda") for _ in range(2))
yr, yi = torch.empty_like(xr), torch.empty_like(xr)
for _ in range(10): A.mult(xr, xi, yr, yi)
with torch.cuda.stream(ctx.torch_stream):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): A.mult(xr, xi, yr, yi)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{mesh.ne} tets p={p}, {N} complex dofs, one-pass form {ceed._lib.load().pa_op_complex_fused(Ar.handle, Ai.handle)}: "
      f"ComplexParOperator::Mult {ms:.4f} ms  {N/ms/1e6:.2f} G complex dof/s")
