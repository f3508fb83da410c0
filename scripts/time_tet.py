"""Dense MFMA path at scale: ND tets on a Kuhn-split cube, timing of curl-curl and curl-curl+mass applies."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import ceed
from palace_amd.fem import tet
n = int(os.environ.get("N", "36")); p = int(os.environ.get("P", "3")); reps = int(os.environ.get("REPS", "20"))
t0 = time.time()
mesh = tet.cube_tet_mesh(n)
if os.environ.get("WARP"):  # tet10 geometry, curved for x > WARP only (a mesh with a curved region)
    x0 = float(os.environ["WARP"])
    def _warp(X):
        w = np.clip(X[:, 0] - x0, 0.0, None) ** 2
        return X + np.stack([0.3 * w * np.sin(3 * X[:, 1]), 0.4 * w * X[:, 2], -0.35 * w * np.cos(2 * X[:, 1])], axis=1)
    mesh = tet.to_quadratic(mesh, _warp)
nd = tet.NDTetSpace(mesh, p)
pts, wts = tet.tet_quadrature(p + 1) if os.environ.get("RULE") == "conical" else tet.default_tet_rule(p)
interp, curl = nd.elem.tables(pts)
geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, **kw)
ident = ceed.coefficient_context(3)
mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
ops = {"curl": ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, ceed.QF_HDIV_33, ident, ceed.EVAL_CURL).finalize(),
       "curlmass": ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, ceed.QF_HDIVMASS_33, np.concatenate([mass, ident]), ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize()}
print("affine sub-operators:", {k: o.dense_affine() for k, o in ops.items()})
print(f"setup {time.time()-t0:.1f}s: {mesh.ne} tets, p={p}, P={nd.P}, Q={len(wts)}, {nd.ndofs} dofs", flush=True)
x = torch.rand(nd.ndofs, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
for name, op in ops.items():
    for _ in range(3): op.mult(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): op.mult(x, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nct = 3 if name == "curl" else 6
    flops = mesh.ne * (2 * 2 * nct * len(wts) * nd.P)
    print(f"{name:9s} mult {ms:.4f} ms  {op.algorithmic_bytes()/ms/1e6:.0f} GB/s alg  {nd.ndofs/ms/1e6:.2f} Gdof/s  {mesh.ne/ms/1e6:.3f} Gelem/s  {flops/ms/1e9:.2f} TFLOP/s (table contractions)")
