"""The operators of the flux error estimators (linalg/errorestimator.cpp) on the device vs the oracle:

* the mixed mass (v, C u) between a Nedelec and a Raviart-Thomas space, either way round -- FluxProjector's `Flux`
  (errorestimator.cpp:164-176, f_apply_hcurlhdiv_33 / f_apply_hdivhcurl_33 chosen by fem/integ/vecfemass.cpp:88-101);
* the element error integrator (fem/libceed/integrator.cpp:550-626 with hcurlhdiv_error_33_qf.h) of
  GradFluxErrorEstimator / CurlFluxErrorEstimator;
* both together as ComputeErrorEstimates does (errorestimator.cpp:189-268): smooth flux by PCG + Jacobi on the mass matrix
  of the smooth space, then the element-wise error -- the estimate of a field that already lies in the smooth space vanishes.

Straight and curved tetrahedra (curl-oriented ND restriction for p >= 2) and the hexahedra of the O-grid cylinder."""
import numpy as np
import pytest

from oracle import palace_oracle as po
from tests import util

pytestmark = pytest.mark.gpu
REL = 1e-12


def _warp(X):
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    return np.stack([x + 0.04 * np.sin(2 * y + z), y + 0.05 * x * z, z - 0.03 * np.cos(3 * x) * y], axis=1)


def _tet_setup(kind, p):
    from palace_amd import ceed
    from palace_amd.fem import rt, tet

    m = tet.cube_tet_mesh(3)
    m.attr[:] = 1 + (np.arange(m.ne) % 2)
    if kind == "tet10":
        m2 = tet.to_quadratic(m, _warp)
        m2.attr[:] = m.attr
        m = m2
    nd, sp = tet.NDTetSpace(m, p), rt.RTTetSpace(m, p)
    pts, wts = tet.tet_quadrature(p + 1)
    nint, ncurl = nd.elem.tables(pts)
    rint, _ = sp.elem.tables(pts)
    geom = ceed.DenseGeomFactorData(m.elem_nodes, m.nodes, m.attr, m.geometry_grad_table(pts), wts)
    J = m.jacobians(pts)
    ogeom = po.build_geom_factor_33(m.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(m.ne, -1, 9))
    if nd.diagonal_transform:
        ndb = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, nint, ncurl, orients=nd.orients)
        ndo = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, nint, ncurl, ogeom, po.QF_HCURL, None)
    else:
        ndb = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, nint, ncurl, curl_orients=nd.curl_orients)
        ndo = po.CeedOperatorOracle(nd.ndofs, nd.offsets, None, nint, ncurl, ogeom, po.QF_HCURL, None,
                                    curl_orients=nd.curl_orients)
    rtb = ceed.DenseBlock(ceed.FE_HDIV, sp.ndofs, sp.offsets, rint, None, orients=sp.orients)
    rto = po.CeedOperatorOracle(sp.ndofs, sp.offsets, sp.orients, rint, rint, ogeom, po.QF_HDIV, None)
    return geom, ogeom, ndb, ndo, rtb, rto, 2


def _hex_setup(mesh, p):
    from palace_amd import ceed
    from palace_amd.fem import rthex
    from palace_amd.fem.basis1d import gauss_legendre
    from palace_amd.fem.fespace import NDHexSpace

    q1d = p + 1
    nd, sp = NDHexSpace(mesh, p), rthex.RTHexSpace(mesh, p)
    _, wts = po.hex_quadrature(q1d)
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.x, mesh.attr, po.mesh_q2_grad_table(q1d), wts)
    ogeom = util.oracle_geom(mesh, q1d)
    off, ori = nd.native_restriction()
    nint, ncurl = util.dense_tables(nd, q1d)
    nint, ncurl = np.asarray(nint).reshape(3, -1, nd.P), np.asarray(ncurl).reshape(3, -1, nd.P)
    rint, _ = rthex.rt_hex_tables(p, gauss_legendre(q1d)[0])
    ndb = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, off, nint, ncurl, orients=ori)
    ndo = po.CeedOperatorOracle(nd.ndofs, off, ori, nint, ncurl, ogeom, po.QF_HCURL, None)
    rtb = ceed.DenseBlock(ceed.FE_HDIV, sp.ndofs, sp.elem_dof_lex, rint, None, orients=sp.elem_sign_lex < 0)
    rto = po.CeedOperatorOracle(sp.ndofs, sp.elem_dof_lex, sp.elem_sign_lex < 0, rint, rint, ogeom, po.QF_HDIV, None)
    return geom, ogeom, ndb, ndo, rtb, rto, int(mesh.attr.max())


def _check(setup, seed):
    import torch

    from palace_amd import ceed

    geom, ogeom, ndb, ndo, rtb, rto, nattr = setup
    rng = np.random.default_rng(seed)
    c_ns, b_ns = util.make_ctx("nonsym", nattr)  # a general 3x3 material exposes a swapped factor order
    c_an, b_an = util.make_ctx("aniso", nattr)
    for qf, qfo, (tb, to), (sb, so) in ((ceed.QF_HCURLHDIV_33, po.QF_HCURLHDIV, (ndb, ndo), (rtb, rto)),
                                        (ceed.QF_HDIVHCURL_33, po.QF_HDIVHCURL, (rtb, rto), (ndb, ndo))):
        op = ceed.Operator(sb.lsize, tb.lsize).add_dense_mixed_integrator(geom, tb, sb, qf, b_ns).finalize()
        x = rng.uniform(-1, 1, tb.lsize)
        ref = po.MixedSpaceOracle(to, so, ogeom, qfo, c_ns).apply_add(x, np.zeros(sb.lsize))
        y = torch.empty(sb.lsize, dtype=torch.float64, device="cuda")
        op.mult(torch.from_numpy(x).cuda(), y)
        assert np.abs(y.cpu().numpy() - ref).max() < REL * np.abs(ref).max(), (qfo, "mult")
        y2 = torch.from_numpy(ref.copy()).cuda()
        op.add_mult(torch.from_numpy(x).cuda(), y2)
        assert np.abs(y2.cpu().numpy() - 2 * ref).max() < REL * np.abs(ref).max(), (qfo, "add_mult")
        assert not op.is_symmetric()
    pair = np.concatenate([b_an, b_ns])
    for qf, qfo, (b1, o1), (b2, o2) in ((ceed.QF_HCURLHDIV_ERROR_33, po.QF_HCURLHDIV_ERROR, (ndb, ndo), (rtb, rto)),
                                        (ceed.QF_HDIVHCURL_ERROR_33, po.QF_HDIVHCURL_ERROR, (rtb, rto), (ndb, ndo))):
        integ = ceed.ElementErrorIntegrator(geom, b1, b2, qf, pair)
        u1, u2 = rng.uniform(-1, 1, b1.lsize), rng.uniform(-1, 1, b2.lsize)
        e0 = rng.uniform(0, 1, integ.ne)  # ApplyAdd: accumulates
        ref = po.MixedSpaceOracle(o1, o2, ogeom, qfo, c_an, c_ns).error_add(u1, u2, e0.copy())
        est = torch.from_numpy(e0.copy()).cuda()
        integ.apply_add(torch.from_numpy(u1).cuda(), torch.from_numpy(u2).cuda(), est)
        assert np.abs(est.cpu().numpy() - ref).max() < REL * np.abs(ref).max(), qfo
        assert (ref - e0).min() > 0


@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("kind", ["tet4", "tet10"])
def test_mixed_mass_and_error_integrators_tets(kind, p):
    _check(_tet_setup(kind, p), p)


@pytest.mark.parametrize("p", [1, 2])
def test_mixed_mass_and_error_integrators_hexes(cylinder_mesh, p):
    _check(_hex_setup(cylinder_mesh, p), 10 + p)


def test_mixed_operator_argument_checks():
    from palace_amd import ceed
    from palace_amd.lib import PalaceAmdError

    geom, ogeom, ndb, ndo, rtb, rto, nattr = _tet_setup("tet4", 1)
    _, blob = util.make_ctx("aniso", nattr)
    with pytest.raises(PalaceAmdError, match="element types"):  # H(curl) trial needs the hcurlhdiv QFunction
        ceed.Operator(rtb.lsize, ndb.lsize).add_dense_mixed_integrator(geom, ndb, rtb, ceed.QF_HDIVHCURL_33, blob)
    with pytest.raises(PalaceAmdError, match="dimensions"):
        ceed.Operator(ndb.lsize, rtb.lsize).add_dense_mixed_integrator(geom, ndb, rtb, ceed.QF_HCURLHDIV_33, blob)
    with pytest.raises(PalaceAmdError, match="mixed-space"):  # (a pair QFunction: no two-space form)
        ceed.Operator(rtb.lsize, ndb.lsize).add_dense_mixed_integrator(geom, ndb, rtb, ceed.QF_HDIVMASS_33, blob)
    op = ceed.Operator(rtb.lsize, ndb.lsize).add_dense_mixed_integrator(geom, ndb, rtb, ceed.QF_HCURLHDIV_33, blob).finalize()
    import torch

    with pytest.raises(PalaceAmdError, match="diagonal"):
        op.assemble_diagonal(torch.empty(rtb.lsize, dtype=torch.float64, device="cuda"))


@pytest.mark.parametrize("kind", ["tet4", "tet10"])
def test_grad_flux_error_estimate(kind):
    """ComputeErrorEstimates for GradFluxErrorEstimator (errorestimator.cpp:189-268, :271-360) with the library's pieces:
    D = M_RT^-1 Flux(eps) E by PCG + Jacobi (ConfigureLinearSolver, :66-107, use_mg = false), then eta_e^2 =
    int_e |eps^-1/2 D - eps^1/2 E|^2.  Against the same procedure through the oracle operators (dense solve), and: the
    estimate of a field whose flux already lies in the smooth space is zero to solver tolerance."""
    import torch

    from palace_amd import ceed, linalg

    p = 2
    geom, ogeom, ndb, ndo, rtb, rto, nattr = _tet_setup(kind, p)
    eps = np.array([[2.0, 0.3, 0.0], [0.3, 1.5, 0.1], [0.0, 0.1, 1.2]])
    w, V = np.linalg.eigh(eps)
    sq, isq = (V * np.sqrt(w)) @ V.T, (V / np.sqrt(w)) @ V.T
    c_eps = po.CoeffCtx(attr_mat=[0] * nattr, mat_coeff=[eps])
    c_sq = po.CoeffCtx(attr_mat=[0] * nattr, mat_coeff=[sq])
    c_isq = po.CoeffCtx(attr_mat=[0] * nattr, mat_coeff=[isq])
    c_id = po.CoeffCtx()
    ctx = linalg.Context()
    flux = ceed.Operator(rtb.lsize, ndb.lsize).add_dense_mixed_integrator(geom, ndb, rtb, ceed.QF_HCURLHDIV_33,
                                                                          c_eps.pack()).finalize()
    mass = ceed.Operator(rtb.lsize, rtb.lsize).add_dense_integrator(geom, rtb, ceed.QF_HDIV_33, c_id.pack(),
                                                                    ceed.EVAL_INTERP).finalize()
    M = linalg.ParOperator(ctx, mass, np.zeros(0, np.int32), linalg.DIAG_ONE)
    cg = linalg.cg(ctx, M, linalg.jacobi(ctx, M), rel_tol=1e-13, max_it=500)
    integ = ceed.ElementErrorIntegrator(geom, ndb, rtb, ceed.QF_HCURLHDIV_ERROR_33, np.concatenate([c_sq.pack(), c_isq.pack()]))
    rng = np.random.default_rng(3)
    E = rng.uniform(-1, 1, ndb.lsize)
    Ed = torch.from_numpy(E).cuda()
    rhs = torch.empty(rtb.lsize, dtype=torch.float64, device="cuda")
    flux.mult(Ed, rhs)
    D = torch.zeros_like(rhs)
    cg.mult(rhs, D)
    est = torch.zeros(integ.ne, dtype=torch.float64, device="cuda")
    integ.apply_add(Ed, D, est)
    # oracle: the same with a dense solve
    Mo = po.CeedOperatorOracle(rto.lsize, rto.off, None if rto.sgn is None else rto.sgn < 0, rto.interp, rto.interp, ogeom,
                               po.QF_HDIV, c_id)
    Md = np.stack([Mo.apply_add(e, np.zeros(rto.lsize)) for e in np.eye(rto.lsize)], axis=1)
    rhs_o = po.MixedSpaceOracle(ndo, rto, ogeom, po.QF_HCURLHDIV, c_eps).apply_add(E, np.zeros(rto.lsize))
    D_o = np.linalg.solve(Md, rhs_o)
    est_o = po.MixedSpaceOracle(ndo, rto, ogeom, po.QF_HCURLHDIV_ERROR, c_sq, c_isq).error_add(E, D_o, np.zeros(integ.ne))
    assert np.abs(D.cpu().numpy() - D_o).max() < 1e-9 * np.abs(D_o).max()
    assert np.abs(est.cpu().numpy() - est_o).max() < 1e-9 * est_o.max()
    assert est_o.min() > 0
    if kind != "tet4":
        return
    # a smooth flux is reproduced (straight elements): the L2 projection of a constant vector c onto ND_p is c itself, eps c is
    # constant and lies in RT_p, so D = eps E exactly and every element estimate vanishes
    Nd = po.CeedOperatorOracle(ndo.lsize, ndo.off, None if ndo.sgn is None else ndo.sgn < 0, ndo.interp, ndo.deriv, ogeom,
                               po.QF_HCURL, c_id, curl_orients=None if ndo.cor is None else ndo.cor.astype(np.int8))
    Mn = np.stack([Nd.apply_add(e, np.zeros(ndo.lsize)) for e in np.eye(ndo.lsize)], axis=1)
    cvec = np.array([0.7, -0.4, 1.1])
    # load (v_j, c) = sum_q w detJ vhat_j . (A^T c) with A = adj(J)^T / detJ stored column-major in rows 2..10 of the geometry data
    A = ogeom[:, 2:, :].reshape(ogeom.shape[0], 3, 3, -1)  # [e, column, row, q]
    atc = np.einsum("ejiq,i->eqj", A, cvec)
    le = np.einsum("dqj,eq,eqd->ej", ndo.interp, ogeom[:, 1, :], atc)
    load = np.zeros(ndo.lsize)
    np.add.at(load, ndo.off.ravel(), ndo._restrict_t(le, slice(None)).ravel())
    Ecd = torch.from_numpy(np.linalg.solve(Mn, load)).cuda()
    flux.mult(Ecd, rhs)
    D.zero_()
    cg.mult(rhs, D)
    est.zero_()
    integ.apply_add(Ecd, D, est)
    scale = float(torch.dot(D, D)) / D.numel()
    assert float(est.max()) < 1e-18 * scale, (float(est.max()), scale)
