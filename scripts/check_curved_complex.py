import sys
import numpy as np, torch
sys.path.insert(0, ".")
from palace_amd import ceed, linalg
from palace_amd.fem import tet
from tests import util
ctx = linalg.Context()
mesh = tet.cube_tet_mesh(4)
mesh.attr[:] = 1 + (np.arange(mesh.ne) % 2)
warp = lambda X: np.stack([X[:, 0] + 0.04 * np.sin(2 * X[:, 1] + X[:, 2]), X[:, 1] + 0.05 * X[:, 0] * X[:, 2], X[:, 2] - 0.03 * np.cos(3 * X[:, 0]) * X[:, 1]], axis=1)
m2 = tet.to_quadratic(mesh, warp); m2.attr[:] = mesh.attr; mesh = m2
worst = 0.0
for p in (1, 2, 3):
    nd = tet.NDTetSpace(mesh, p)
    pts, wts = tet.default_tet_rule(p)
    interp, curl = nd.elem.tables(pts)
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
    kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
    block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, **kw)
    _, b3 = util.make_ctx("aniso", 2); _, bm = util.make_ctx("scalar", 2); _, bi = util.make_ctx("aniso", 2)
    n = nd.ndofs
    Ar = ceed.Operator(n, n).add_dense_integrator(geom, block, ceed.QF_HDIVMASS_33, np.concatenate([bm, b3]), ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize()
    for imode in ("mass", "curl", "both"):
        if imode == "mass":
            Ai = ceed.Operator(n, n).add_dense_integrator(geom, block, ceed.QF_HCURL_33, bi, ceed.EVAL_INTERP).finalize()
        elif imode == "curl":
            Ai = ceed.Operator(n, n).add_dense_integrator(geom, block, ceed.QF_HDIV_33, bi, ceed.EVAL_CURL).finalize()
        else:
            Ai = ceed.Operator(n, n).add_dense_integrator(geom, block, ceed.QF_HDIVMASS_33, np.concatenate([bi, bm]), ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize()
        rng = np.random.default_rng(3)
        xr, xi = (torch.from_numpy(rng.uniform(-1, 1, n)).cuda() for _ in range(2))
        A = linalg.ComplexParOperator(ctx, Ar, Ai, np.zeros(0, np.int32), linalg.DIAG_ONE)
        yr, yi = torch.empty_like(xr), torch.empty_like(xr)
        A.mult(xr, xi, yr, yi)
        t = [torch.empty_like(xr) for _ in range(4)]
        Ar.mult(xr, t[0]); Ai.mult(xi, t[1]); Ai.mult(xr, t[2]); Ar.mult(xi, t[3])
        rr, ri = t[0] - t[1], t[2] + t[3]
        e = max(float((yr - rr).abs().max() / rr.abs().max()), float((yi - ri).abs().max() / ri.abs().max()))
        worst = max(worst, e)
        print("p=%d %-5s fused=%d  rel diff %.2e" % (p, imode, ceed._lib.load().pa_op_complex_fused(Ar.handle, Ai.handle), e))
print("WORST %.2e" % worst)
