"""Two-space operators (pa_mixed.hip) at scale: ND <-> RT mixed mass, H1 -> ND mixed gradient and the element error integrator on a
Kuhn-split cube of tetrahedra.  Prints ms per apply and the rate of the bytes the kernel has to move per element (geometry
data 11 Q doubles + the element dofs in and out + the E-vector), one wave per element."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import ceed
from palace_amd.fem import rt, tet
n = int(os.environ.get("N", "24")); p = int(os.environ.get("P", "2")); reps = int(os.environ.get("REPS", "20"))
t0 = time.time()
mesh = tet.cube_tet_mesh(n)
nd, sp, h1 = tet.NDTetSpace(mesh, p), rt.RTTetSpace(mesh, p), tet.H1TetSpace(mesh, p)
pts, wts = tet.tet_quadrature(p + 1)
nint, ncurl = nd.elem.tables(pts)
rint, _ = sp.elem.tables(pts)
hint, hgrad = h1.elem.tables(pts)
geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
ndb = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, nint, ncurl, **kw)
rtb = ceed.DenseBlock(ceed.FE_HDIV, sp.ndofs, sp.offsets, rint, None, orients=sp.orients)
h1b = ceed.DenseBlock(ceed.FE_H1, h1.ndofs, h1.offsets, hint, hgrad)
ident = ceed.coefficient_context(3)
ops = {"mass ND->RT": (ceed.Operator(rtb.lsize, ndb.lsize).add_dense_mixed_integrator(geom, ndb, rtb, ceed.QF_HCURLHDIV_33, ident).finalize(), ndb, rtb),
       "mass RT->ND": (ceed.Operator(ndb.lsize, rtb.lsize).add_dense_mixed_integrator(geom, rtb, ndb, ceed.QF_HDIVHCURL_33, ident).finalize(), rtb, ndb),
       "grad H1->ND": (ceed.Operator(ndb.lsize, h1b.lsize).add_dense_mixed_integrator(geom, h1b, ndb, ceed.QF_HCURL_33, ident).finalize(), h1b, ndb)}
Q = len(wts)
print(f"setup {time.time()-t0:.1f}s: {mesh.ne} tets, p={p}, Q={Q}, ND {nd.ndofs} (P={nd.P}), RT {sp.ndofs} (P={sp.P}), H1 {h1.ndofs} (P={h1.P})", flush=True)
def timed(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, (op, tb, sb) in ops.items():
    x = torch.rand(tb.lsize, dtype=torch.float64, device="cuda"); y = torch.zeros(sb.lsize, dtype=torch.float64, device="cuda")
    ms = timed(lambda: op.mult(x, y))
    byt = mesh.ne * 8 * (11 * Q + tb.P + 2 * sb.P) + 8 * sb.lsize + mesh.ne * 4 * (tb.P + sb.P)
    print(f"{name:12s} {ms:.4f} ms  {byt/ms/1e6:.0f} GB/s  {mesh.ne/ms/1e6:.3f} Gelem/s")
integ = ceed.ElementErrorIntegrator(geom, ndb, rtb, ceed.QF_HCURLHDIV_ERROR_33, np.concatenate([ident, ident]))
u1 = torch.rand(ndb.lsize, dtype=torch.float64, device="cuda"); u2 = torch.rand(rtb.lsize, dtype=torch.float64, device="cuda")
est = torch.zeros(integ.ne, dtype=torch.float64, device="cuda")
ms = timed(lambda: integ.apply_add(u1, u2, est))
byt = mesh.ne * 8 * (11 * Q + ndb.P + rtb.P + 2) + mesh.ne * 4 * (ndb.P + rtb.P)
print(f"{'error ND,RT':12s} {ms:.4f} ms  {byt/ms/1e6:.0f} GB/s  {mesh.ne/ms/1e6:.3f} Gelem/s")
