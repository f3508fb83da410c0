"""Weighted sums of H(curl) integrators fused into one sub-operator (pa_op_add_sub_sum / pa_op_add_sub_dense_sum;
SURVEY.md 8f-1: sum_k a_k {K, C, M} behind BuildParSumOperator, linalg/rap.cpp:843-919) against the sum of the
separately assembled operators and against the oracle, with different attribute maps per term."""
import numpy as np
import pytest

from oracle import palace_oracle as po
from tests import util

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("which", ["K+M+C", "M+C", "K+K", "KM+C"])
@pytest.mark.parametrize("p", [1, 2, 3])
def test_fused_sum_hex(cylinder_mesh, p, which):
    import torch

    from palace_amd import ceed
    from palace_amd.fem.fespace import NDHexSpace

    mesh = cylinder_mesh
    na = int(mesh.attr.max())
    nd = NDHexSpace(mesh, p)
    q1d = p + 1
    geom = ceed.GeomFactorData(mesh, q1d)
    ogeom = util.oracle_geom(mesh, q1d)
    _, b_an = util.make_ctx("aniso", na)
    _, b_sc = util.make_ctx("scalar", na)
    _, b_id = util.make_ctx("identity")
    K = (ceed.QF_HDIV_33, b_an, "hdiv")
    K2 = (ceed.QF_HDIV_33, b_id, "hdiv")
    M = (ceed.QF_HCURL_33, b_sc, "hcurl")
    Cc = (ceed.QF_HCURL_33, b_an, "hcurl")
    KM = (ceed.QF_HDIVMASS_33, np.concatenate([b_sc, b_id]), "hdivmass")
    terms = {"K+M+C": [(1.0, K), (-0.37, M), (0.21, Cc)], "M+C": [(2.0, M), (-1.5, Cc)],
             "K+K": [(0.5, K), (0.25, K2)], "KM+C": [(1.0, KM), (0.3, Cc)]}[which]
    fused = ceed.Operator(nd.ndofs, nd.ndofs).add_integrator_sum(
        geom, nd, [(a, t[0], t[1]) for a, t in terms]).finalize()
    x = np.random.default_rng(p).uniform(-1, 1, nd.ndofs)
    y = torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")
    fused.mult(_dev(x), y)
    ref = sum(a * util.oracle_apply_c(nd, ogeom, t[2], t[1], x, q1d) for a, t in terms)
    assert np.abs(y.cpu().numpy() - ref).max() < 1e-12 * np.abs(ref).max()
    d = torch.empty_like(y)
    fused.assemble_diagonal(d)
    dsum = torch.zeros_like(y)
    for a, t in terms:
        ops = {"hdiv": ceed.EVAL_CURL, "hcurl": ceed.EVAL_INTERP, "hdivmass": ceed.EVAL_CURL | ceed.EVAL_INTERP}[t[2]]
        op = ceed.Operator(nd.ndofs, nd.ndofs).add_integrator(geom, nd, t[0], t[1], ops).finalize()
        dk = torch.empty_like(y)
        op.assemble_diagonal(dk)
        dsum += a * dk
    assert float((d - dsum).abs().max()) < 1e-12 * float(dsum.abs().max())


@pytest.mark.parametrize("p", [1, 2])
def test_fused_sum_tets(p):
    import torch

    from palace_amd import ceed
    from palace_amd.fem import tet

    def warp(X):
        return X + 0.03 * np.stack([np.sin(2 * X[:, 1]), X[:, 0] * X[:, 2], np.cos(3 * X[:, 0])], axis=1)

    mesh = tet.to_quadratic(tet.cube_tet_mesh(3), warp)
    mesh.attr[:] = 1 + (np.arange(mesh.ne) % 2)
    nd = tet.NDTetSpace(mesh, p)
    pts, wts = tet.tet_quadrature(p + 1)
    interp, curl = nd.elem.tables(pts)
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
    J = mesh.jacobians(pts)
    ogeom = po.build_geom_factor_33(mesh.attr.astype(np.float64), wts,
                                    np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9))
    kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
    block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, **kw)
    c_an, b_an = util.make_ctx("aniso", 2)
    c_sc, b_sc = util.make_ctx("scalar", 2)
    c_id, b_id = util.make_ctx("identity")
    terms = [(1.0, ceed.QF_HDIV_33, b_id, po.QF_HDIV, c_id), (-0.8, ceed.QF_HCURL_33, b_sc, po.QF_HCURL, c_sc),
             (0.4, ceed.QF_HCURL_33, b_an, po.QF_HCURL, c_an)]
    fused = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator_sum(geom, block, [t[:3] for t in terms]).finalize()
    x = np.random.default_rng(p).uniform(-1, 1, nd.ndofs)
    y = torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")
    fused.mult(_dev(x), y)
    ref = np.zeros(nd.ndofs)
    for a, _, _, qf_o, c in terms:
        orc = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients if nd.diagonal_transform else None, interp, curl,
                                    ogeom, qf_o, c, curl_orients=None if nd.diagonal_transform else nd.curl_orients)
        ref += a * orc.apply_add(x, np.zeros(nd.ndofs))
    assert np.abs(y.cpu().numpy() - ref).max() < 1e-12 * np.abs(ref).max()


def test_fused_sum_rejects_other_integrators(cylinder_mesh):
    from palace_amd import ceed
    from palace_amd.fem.fespace import NDHexSpace

    nd = NDHexSpace(cylinder_mesh, 1)
    geom = ceed.GeomFactorData(cylinder_mesh, 2)
    _, b = util.make_ctx("identity")
    with pytest.raises(RuntimeError, match="fused"):
        ceed.Operator(nd.ndofs, nd.ndofs).add_integrator_sum(geom, nd, [(1.0, ceed.QF_H1_1, b)])
