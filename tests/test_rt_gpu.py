"""H(div) mass operator on Raviart-Thomas tetrahedra and the discrete curl ND -> RT on the device vs the oracle
(the flux B = curl A after a solve, drivers/eigensolver.cpp:469-477, and the RT mass of the
flux error estimator, linalg/errorestimator.cpp).  The RT space itself is pinned by tests/test_rt_space.py."""
import numpy as np
import pytest

from oracle import palace_oracle as po
from tests import util

pytestmark = pytest.mark.gpu
REL = 1e-12


def _warp(X):
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    return np.stack([x + 0.04 * np.sin(2 * y + z), y + 0.05 * x * z, z - 0.03 * np.cos(3 * x) * y], axis=1)


def _mesh(kind):
    from palace_amd.fem import tet

    m = tet.cube_tet_mesh(3)
    m.attr[:] = 1 + (np.arange(m.ne) % 2)
    if kind == "tet10":
        m2 = tet.to_quadratic(m, _warp)
        m2.attr[:] = m.attr
        return m2
    return m


def _geom(mesh, pts, wts):
    from palace_amd import ceed

    g = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
    J = mesh.jacobians(pts)
    Jcm = np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9)
    return g, po.build_geom_factor_33(mesh.attr.astype(np.float64), wts, Jcm)


@pytest.mark.parametrize("coeff", ["scalar", "aniso"])
@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("kind", ["tet4", "tet10"])
def test_rt_mass_apply(kind, p, coeff):
    import torch

    from palace_amd import ceed
    from palace_amd.fem import rt, tet

    mesh = _mesh(kind)
    sp = rt.RTTetSpace(mesh, p)
    pts, wts = tet.tet_quadrature(p + 1)
    interp, _ = sp.elem.tables(pts)
    geom, ogeom = _geom(mesh, pts, wts)
    c, blob = util.make_ctx(coeff, 2)
    block = ceed.DenseBlock(ceed.FE_HDIV, sp.ndofs, sp.offsets, interp, None, orients=sp.orients)
    op = ceed.Operator(sp.ndofs, sp.ndofs).add_dense_integrator(geom, block, ceed.QF_HDIV_33, blob,
                                                                ceed.EVAL_INTERP).finalize()
    orc = po.CeedOperatorOracle(sp.ndofs, sp.offsets, sp.orients, interp, interp, ogeom, po.QF_HDIV, c)
    x = np.random.default_rng(p).uniform(-1, 1, sp.ndofs)
    ref = orc.apply_add(x, np.zeros(sp.ndofs))
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    op.mult(xd, yd)
    assert np.abs(yd.cpu().numpy() - ref).max() < REL * np.abs(ref).max()
    dd = torch.empty_like(xd)
    op.assemble_diagonal(dd)
    dref = orc.diagonal()
    assert np.abs(dd.cpu().numpy() - dref).max() < REL * np.abs(dref).max()
    assert float(dd.min()) > 0.0


@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("kind", ["tet4", "tet10"])
def test_rt_divdiv_apply(kind, p):
    """DivDivIntegrator on a Raviart-Thomas space (fem/integ/divdiv.cpp:31-57: Div | Weight, f_apply_l2_1 with a scalar
    coefficient): (c div u, div v) vs the oracle; the divergence of a discrete curl vanishes."""
    import torch

    from palace_amd import ceed, linalg
    from palace_amd.fem import rt, tet

    mesh = _mesh(kind)
    sp = rt.RTTetSpace(mesh, p)
    pts, wts = tet.tet_quadrature(p + 1)
    interp, div = sp.elem.tables(pts)
    geom, ogeom = _geom(mesh, pts, wts)
    c1 = po.CoeffCtx(attr_mat=[1, 0], mat_coeff=[np.array([1.9]), np.array([0.4])], dim=1)
    block = ceed.DenseBlock(ceed.FE_HDIV, sp.ndofs, sp.offsets, interp, div[None], orients=sp.orients)
    op = ceed.Operator(sp.ndofs, sp.ndofs).add_dense_integrator(geom, block, ceed.QF_L2_1, c1.pack(),
                                                                ceed.EVAL_DIV | ceed.EVAL_WEIGHT).finalize()
    orc = po.CeedOperatorOracle(sp.ndofs, sp.offsets, sp.orients, interp, div, ogeom, po.QF_L2_1, c1, qw=wts, deriv_comps=1)
    x = np.random.default_rng(p).uniform(-1, 1, sp.ndofs)
    ref = orc.apply_add(x, np.zeros(sp.ndofs))
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    op.mult(xd, yd)
    assert np.abs(yd.cpu().numpy() - ref).max() < REL * np.abs(ref).max()
    dd = torch.empty_like(xd)
    op.assemble_diagonal(dd)
    dref = orc.diagonal()
    assert np.abs(dd.cpu().numpy() - dref).max() < REL * np.abs(dref).max()
    # div curl = 0: the flux of any Nedelec potential is in the kernel
    nd = tet.NDTetSpace(mesh, p)
    ctx = linalg.Context()
    C = linalg.DenseInterp(ctx, nd.restriction(), sp.restriction(interp_range=True), rt.tet_curl_matrix(p))
    a = torch.from_numpy(np.random.default_rng(p + 7).uniform(-1, 1, nd.ndofs)).cuda()
    b = torch.empty_like(xd)
    C.mult(a, b)
    op.mult(b, yd)
    assert float(yd.abs().max()) < 1e-11 * float(b.abs().max()) * float(dd.abs().max())


@pytest.mark.parametrize("p", [1, 2, 3])
def test_discrete_curl_and_flux_energy(p):
    """B = C A on the device = the oracle's interpolator; (K A, A) = (M_RT B, B) with all three operators on the GPU."""
    import torch

    from palace_amd import ceed, linalg
    from palace_amd.fem import rt, tet

    ctx = linalg.Context()
    mesh = _mesh("tet10")
    nd, sp = tet.NDTetSpace(mesh, p), rt.RTTetSpace(mesh, p)
    Cm = rt.tet_curl_matrix(p)
    C = linalg.DenseInterp(ctx, nd.restriction(), sp.restriction(interp_range=True), Cm)
    orc = po.DenseInterpOracle(nd.restriction(), sp.restriction(interp_range=True), Cm)
    rng = np.random.default_rng(p)
    a = rng.uniform(-1, 1, nd.ndofs)
    ad = torch.from_numpy(a).cuda()
    b = torch.empty(sp.ndofs, dtype=torch.float64, device="cuda")
    C.mult(ad, b)
    ref = orc.mult(a)
    assert np.abs(b.cpu().numpy() - ref).max() < REL * np.abs(ref).max()
    z = rng.uniform(-1, 1, sp.ndofs)
    at = torch.empty_like(ad)
    C.mult_transpose(torch.from_numpy(z).cuda(), at)
    assert abs(z @ b.cpu().numpy() - at.cpu().numpy() @ a) < REL * np.abs(z).sum() * float(b.abs().max())
    # energy identity with the matrix-free operators
    pts, wts = tet.tet_quadrature(p + 1)
    geom, _ = _geom(mesh, pts, wts)
    interp, curl = nd.elem.tables(pts)
    kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
    K = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(
        geom, ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, **kw), ceed.QF_HDIV_33,
        ceed.coefficient_context(3), ceed.EVAL_CURL).finalize()
    rint, _ = sp.elem.tables(pts)
    M = ceed.Operator(sp.ndofs, sp.ndofs).add_dense_integrator(
        geom, ceed.DenseBlock(ceed.FE_HDIV, sp.ndofs, sp.offsets, rint, None, orients=sp.orients), ceed.QF_HDIV_33,
        ceed.coefficient_context(3), ceed.EVAL_INTERP).finalize()
    ka, mb = torch.empty_like(ad), torch.empty_like(b)
    K.mult(ad, ka)
    M.mult(b, mb)
    e_k, e_m = float(ad @ ka), float(b @ mb)
    assert abs(e_k - e_m) < 1e-11 * abs(e_k)


def test_rt_rejects_other_qfunctions():
    from palace_amd import ceed
    from palace_amd.fem import rt, tet

    mesh = _mesh("tet4")
    sp = rt.RTTetSpace(mesh, 1)
    pts, wts = tet.tet_quadrature(2)
    interp, _ = sp.elem.tables(pts)
    geom, _ = _geom(mesh, pts, wts)
    block = ceed.DenseBlock(ceed.FE_HDIV, sp.ndofs, sp.offsets, interp, None, orients=sp.orients)
    with pytest.raises(RuntimeError, match="H\\(div\\)"):
        ceed.Operator(sp.ndofs, sp.ndofs).add_dense_integrator(geom, block, ceed.QF_HCURL_33,
                                                               ceed.coefficient_context(3), ceed.EVAL_INTERP)


@pytest.mark.parametrize("p", [1, 2, 3])
def test_rt_hex_mass_and_discrete_curl(cylinder_mesh, p):
    """Hexahedra (the O-grid cylinder): RT mass through the dense MFMA path, B = C A through the dense interpolator, and
    the energy identity against the sum-factorised curl-curl operator."""
    import torch

    from palace_amd import ceed, linalg
    from palace_amd.fem import rthex
    from palace_amd.fem.basis1d import gauss_legendre
    from palace_amd.fem.fespace import NDHexSpace

    mesh = cylinder_mesh
    q1d = p + 1
    nd, sp = NDHexSpace(mesh, p), rthex.RTHexSpace(mesh, p)
    _, wts = po.hex_quadrature(q1d)
    dgeom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.x, mesh.attr, po.mesh_q2_grad_table(q1d), wts)
    ogeom = util.oracle_geom(mesh, q1d)
    rint, _ = rthex.rt_hex_tables(p, gauss_legendre(q1d)[0])
    c, blob = util.make_ctx("aniso", int(mesh.attr.max()))
    block = ceed.DenseBlock(ceed.FE_HDIV, sp.ndofs, sp.elem_dof_lex, rint, None, orients=sp.elem_sign_lex < 0)
    M = ceed.Operator(sp.ndofs, sp.ndofs).add_dense_integrator(dgeom, block, ceed.QF_HDIV_33, blob,
                                                               ceed.EVAL_INTERP).finalize()
    orc = po.CeedOperatorOracle(sp.ndofs, sp.elem_dof_lex, sp.elem_sign_lex < 0, rint, rint, ogeom, po.QF_HDIV, c)
    rng = np.random.default_rng(p)
    x = rng.uniform(-1, 1, sp.ndofs)
    y = torch.empty(sp.ndofs, dtype=torch.float64, device="cuda")
    M.mult(torch.from_numpy(x).cuda(), y)
    ref = orc.apply_add(x, np.zeros(sp.ndofs))
    assert np.abs(y.cpu().numpy() - ref).max() < REL * np.abs(ref).max()
    # flux of a random potential and its energy
    ctx = linalg.Context()
    Cm = rthex.hex_curl_matrix(p)
    dom = dict(offsets=nd.elem_dof_lex, lsize=nd.ndofs, orients=nd.elem_sign_lex < 0)
    C = linalg.DenseInterp(ctx, dom, sp.restriction(interp_range=True), Cm)
    a = rng.uniform(-1, 1, nd.ndofs)
    ad = torch.from_numpy(a).cuda()
    b = torch.empty(sp.ndofs, dtype=torch.float64, device="cuda")
    C.mult(ad, b)
    bref = po.DenseInterpOracle(dom, sp.restriction(interp_range=True), Cm).mult(a)
    assert np.abs(b.cpu().numpy() - bref).max() < REL * np.abs(bref).max()
    K = ceed.curlcurl_operator(ceed.GeomFactorData(mesh, q1d), nd, ceed.coefficient_context(3))
    M1 = ceed.Operator(sp.ndofs, sp.ndofs).add_dense_integrator(dgeom, block, ceed.QF_HDIV_33,
                                                                ceed.coefficient_context(3), ceed.EVAL_INTERP).finalize()
    ka, mb = torch.empty_like(ad), torch.empty_like(b)
    K.mult(ad, ka)
    M1.mult(b, mb)
    e_k, e_m = float(ad @ ka), float(b @ mb)
    assert abs(e_k - e_m) < 1e-11 * abs(e_k)
