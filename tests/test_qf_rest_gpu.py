"""The QFunction families no integrator of the hot path's callers uses (VERDICT r3 "Missing 7"), through
the C ABI against the oracle (pinned on the reference headers: tests/test_oracle_rest.py):

* H(div) mass on boundary and line elements (f_apply_hdiv_32 | _31 | _21), div-div on plane / boundary elements, and div-div +
  mass in one pass (DivDivMassIntegrator, f_apply_l2mass_22 | _32 | _21 | _31 | _33) -- pa_op_add_sub_dense with FE_HDIV blocks;
* the two-space members on boundary and line elements (f_apply_hcurlhdiv_32 | _31 | _21, f_apply_hdivhcurl_*, f_apply_hcurl_*
  between two spaces: VectorFEMassIntegrator / MixedVectorGradientIntegrator there) -- pa_op_add_sub_dense_mixed, with the
  transposed apply;
* GradientIntegrator (f_apply_hcurlh1d_* on all five geometries) -- pa_op_add_sub_dense_gradient.

The H(div) tables on triangles are the rotated Nedelec ones (tests/test_mixed_grad_gpu.py::_tri_blocks; the divergence of the
rotated field is the scalar curl of the Nedelec one), on segments any one-component table does: the operators are algebraic in
the tables, what is under test is E, B, D, B^T, E^T with the QFunction's D."""
import os

import numpy as np
import pytest

from oracle import palace_oracle as po

pytestmark = pytest.mark.gpu
REL = 1e-12


def _mult(op, x, n):
    import torch

    y = torch.empty(n, dtype=torch.float64, device="cuda")
    op.mult(torch.from_numpy(np.ascontiguousarray(x)).cuda(), y)
    return y.cpu().numpy()


def _mult_t(op, x, n):
    import torch

    y = torch.empty(n, dtype=torch.float64, device="cuda")
    op.mult_transpose(torch.from_numpy(np.ascontiguousarray(x)).cuda(), y)
    return y.cpu().numpy()


def _diag(op, n):
    import torch

    d = torch.empty(n, dtype=torch.float64, device="cuda")
    op.assemble_diagonal(d)
    return d.cpu().numpy()


def _tri(p, surface):
    """Triangles of the reference's cavity2d mesh in the plane or lifted to a curved surface: geometry data (device, oracle),
    quadrature weights, the blocks {H1, ND, rotated-ND as H(div) with its divergence} and their oracle sides."""
    from palace_amd import ceed
    from palace_amd.fem import tri

    M_ = np.load(os.path.join(os.path.dirname(__file__), "golden", "cavity2d_mesh.npz"))
    en = M_["elem_nodes"].astype(np.int64)
    used, inv = np.unique(en[:, :3], return_inverse=True)
    attr = 1 + (np.arange(en.shape[0]) % 2)
    mesh = tri.TriMesh(M_["nodes"][used], inv.reshape(-1, 3), attr, elem_nodes=en, nodes=M_["nodes"])
    nd, h1 = tri.NDTriSpace(mesh, p), tri.H1TriSpace(mesh, p)
    pts, wts = tri.tri_quadrature(p + 1)
    nint, ncurl = nd.elem.tables(pts)
    hint, hgrad = h1.elem.tables(pts)
    rint = np.stack([nint[1], -nint[0]])
    G = mesh.geometry_grad_table(pts)
    if surface:
        xy = mesh.nodes
        L = np.ptp(xy[:, 0])
        nodes = np.column_stack([xy, 0.15 * L * np.sin(3.0 * xy[:, 0] / L) + 0.3 * xy[:, 0] * xy[:, 1] / L])
        J = np.einsum("dqn,eni->eqid", G, nodes[mesh.elem_nodes])  # [e, q, 3, 2]
        ogeom = po.build_geom_factor_32(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 6))
    else:
        nodes = mesh.nodes
        J = mesh.jacobians(pts)
        ogeom = po.build_geom_factor_22(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 4))
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, nodes, mesh.attr, G, wts)
    blocks = dict(
        h1=(ceed.DenseBlock(ceed.FE_H1, h1.ndofs, h1.offsets, hint, hgrad),
            po.CeedOperatorOracle(h1.ndofs, h1.offsets, None, hint, hgrad, ogeom, None, None, vector_fe=False)),
        nd=(ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, nint, ncurl, orients=nd.orients),
            po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, nint, ncurl, ogeom, None, None, qw=wts)),
        rt=(ceed.DenseBlock(ceed.FE_HDIV, nd.ndofs, nd.offsets, rint, ncurl, orients=nd.orients),
            po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, rint, ncurl, ogeom, None, None, qw=wts)))
    return geom, ogeom, wts, blocks, hgrad


def _ctx(rng, sdim, sym=True):
    A = rng.uniform(-1, 1, (sdim, sdim))
    return po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[(A @ A.T if sym else A) + 2 * np.eye(sdim), np.array([0.6])], a=1.2, dim=sdim)


C1 = po.CoeffCtx(attr_mat=[1, 0], mat_coeff=[np.array([1.9]), np.array([0.4])], dim=1)


def _check_square(op, orc, n, rng, tag):
    x = rng.uniform(-1, 1, n)
    ref = orc.apply_add(x, np.zeros(n))
    assert np.abs(_mult(op, x, n) - ref).max() < REL * np.abs(ref).max(), tag
    dref = orc.diagonal()
    assert np.abs(_diag(op, n) - dref).max() < REL * np.abs(dref).max(), tag
    return x, ref


@pytest.mark.parametrize("surface", [False, True])
@pytest.mark.parametrize("p", [1, 2])
def test_hdiv_forms_on_triangles(p, surface):
    from palace_amd import ceed

    geom, ogeom, wts, blocks, _ = _tri(p, surface)
    rtb, rto = blocks["rt"]
    n = rtb.lsize
    sdim = 3 if surface else 2
    rng = np.random.default_rng(100 + 10 * p + surface)
    cm = _ctx(rng, sdim)
    flips = rto.sgn < 0
    q_mass, oq_mass = (ceed.QF_HDIV_32, po.QF_HDIV_32) if surface else (ceed.QF_HDIV_22, po.QF_HDIV_22)
    q_pair = ceed.QF_L2MASS_32 if surface else ceed.QF_L2MASS_22
    mass = ceed.Operator(n, n).add_dense_integrator(geom, rtb, q_mass, cm.pack(), ceed.EVAL_INTERP).finalize()
    x, ym = _check_square(mass, po.CeedOperatorOracle(n, rto.off, flips, rto.interp, rto.deriv, ogeom, oq_mass, cm, qw=wts), n, rng, "mass")
    dd = ceed.Operator(n, n).add_dense_integrator(geom, rtb, ceed.QF_L2_1, C1.pack(), ceed.EVAL_DIV | ceed.EVAL_WEIGHT).finalize()
    _check_square(dd, po.CeedOperatorOracle(n, rto.off, flips, rto.interp, rto.deriv, ogeom, po.QF_L2_1, C1, qw=wts), n, rng, "div-div")
    pair = ceed.Operator(n, n).add_dense_integrator(geom, rtb, q_pair, np.concatenate([cm.pack(), C1.pack()]),
                                                    ceed.EVAL_INTERP | ceed.EVAL_DIV | ceed.EVAL_WEIGHT).finalize()
    orc = po.CeedOperatorOracle(n, rto.off, flips, rto.interp, rto.deriv, ogeom, po.QF_L2MASS, cm, C1, qw=wts)
    x, yp = _check_square(pair, orc, n, rng, "div-div + mass")
    # ... and it is the sum of its two halves on the device
    ys = _mult(mass, x, n) + _mult(dd, x, n)
    assert np.abs(ys - yp).max() < REL * np.abs(yp).max()
    if surface:  # a flat piece of surface gives the plane operator: the contravariant map must not be the covariant one
        nd_mass = ceed.Operator(n, n).add_dense_integrator(geom, blocks["nd"][0], ceed.QF_HCURL_32, cm.pack(), ceed.EVAL_INTERP).finalize()
        assert np.abs(_mult(nd_mass, x, n) - _mult(mass, x, n)).max() > 1e-3 * np.abs(ym).max()


@pytest.mark.parametrize("surface", [False, True])
@pytest.mark.parametrize("p", [1, 2])
def test_two_space_and_gradient_forms_on_triangles(p, surface):
    from palace_amd import ceed

    geom, ogeom, wts, blocks, hgrad = _tri(p, surface)
    (h1b, h1o), (ndb, ndo), (rtb, rto) = blocks["h1"], blocks["nd"], blocks["rt"]
    sdim = 3 if surface else 2
    rng = np.random.default_rng(200 + 10 * p + surface)
    c_ns = _ctx(rng, sdim, sym=False)
    if surface:
        cases = ((ceed.QF_HCURLHDIV_32, po.QF_HCURLHDIV_32, (ndb, ndo, None), (rtb, rto)),
                 (ceed.QF_HDIVHCURL_32, po.QF_HDIVHCURL_32, (rtb, rto, None), (ndb, ndo)),
                 (ceed.QF_HCURL_32, po.QF_HCURL_32, (h1b, h1o, hgrad), (ndb, ndo)),       # MixedVectorGradient, mixedvecgrad.cpp:113-120
                 (ceed.QF_HCURLHDIV_32, po.QF_HCURLHDIV_32, (h1b, h1o, hgrad), (rtb, rto)))
        for qf, qfo, (tb, to, tt), (sb, so) in cases:
            op = ceed.Operator(sb.lsize, tb.lsize).add_dense_mixed_integrator(geom, tb, sb, qf, c_ns.pack()).finalize()
            x, y = rng.uniform(-1, 1, tb.lsize), rng.uniform(-1, 1, sb.lsize)
            ref = po.MixedSpaceOracle(to, so, ogeom, qfo, c_ns, first_tab=tt).apply_add(x, np.zeros(sb.lsize))
            ax = _mult(op, x, sb.lsize)
            assert np.abs(ax - ref).max() < REL * np.abs(ref).max(), (qfo, tb.fe_type)
            aty = _mult_t(op, y, tb.lsize)  # adjointness of the transposed apply
            assert abs(y @ ax - aty @ x) < 1e-11 * np.abs(y).sum() * np.abs(ax).max(), qfo
    # GradientIntegrator: H1 trial (Grad), vector H1 test with sdim components, byNODES
    nh = h1b.lsize
    vtest = ceed.DenseBlock(ceed.FE_H1, sdim * nh, h1b.offsets, h1b.interp, None)
    qf = ceed.QF_HCURLH1D_32 if surface else ceed.QF_HCURLH1D_22
    op = ceed.Operator(sdim * nh, nh).add_dense_gradient_integrator(geom, h1b, vtest, nh, qf, c_ns.pack()).finalize()
    x = rng.uniform(-1, 1, nh)
    ref = po.MixedSpaceOracle(h1o, h1o, ogeom, po.QF_HCURLH1D, c_ns, first_tab=hgrad).gradient_add(x, np.zeros(sdim * nh), nh)
    assert np.abs(_mult(op, x, sdim * nh) - ref).max() < REL * np.abs(ref).max()
    # with the identity coefficient, (grad u, e_c) summed over the test functions of component c is the integral of d_c u:
    # zero for a constant u
    ident = po.CoeffCtx(dim=sdim)
    opi = ceed.Operator(sdim * nh, nh).add_dense_gradient_integrator(geom, h1b, vtest, nh, qf, ident.pack()).finalize()
    assert np.abs(_mult(opi, np.ones(nh), sdim * nh)).max() < 1e-12 * np.abs(ref).max()


def _curve(sdim, p, ne=37):
    from palace_amd import ceed
    from palace_amd.fem.basis1d import gauss_legendre, gauss_lobatto, lagrange_eval

    t = np.linspace(0.0, 1.0, 2 * ne + 1) ** 1.3
    X = np.stack([np.cos(2.1 * t) + 0.2 * t, np.sin(1.7 * t), 0.4 * t * t + 0.1 * np.sin(5 * t)][:sdim], axis=1)
    en = np.stack([2 * np.arange(ne), 2 * np.arange(ne) + 2, 2 * np.arange(ne) + 1], axis=1).astype(np.int32)
    attr = (1 + (np.arange(ne) % 2)).astype(np.int32)
    qx, qw = gauss_legendre(p + 2)
    _, Gm = lagrange_eval(np.array([0.0, 1.0, 0.5]), qx)
    geom = ceed.DenseGeomFactorData(en, X, attr, Gm[None], qw)
    ogeom = po.build_geom_factor_line(attr.astype(np.float64), qw, np.einsum("qn,eni->eqi", Gm, X[en]))
    B, G = lagrange_eval(gauss_lobatto(p + 1), qx)
    h1_off = np.zeros((ne, p + 1), dtype=np.int32)
    h1_off[:, 0], h1_off[:, p] = np.arange(ne), np.arange(ne) + 1
    for k in range(1, p):
        h1_off[:, k] = ne + 1 + (p - 1) * np.arange(ne) + (k - 1)
    n_h1 = ne + 1 + (p - 1) * ne
    Bo, Go = lagrange_eval(gauss_legendre(p + 1)[0], qx)  # p + 1 interior dofs per segment: values and a "divergence" table
    v_off = ((p + 1) * np.arange(ne)[:, None] + np.arange(p + 1)[None, :]).astype(np.int32)
    v_ori = np.zeros((ne, p + 1), dtype=bool)
    v_ori[1::2] = True
    return dict(geom=geom, ogeom=ogeom, qw=qw, ne=ne, h1=(n_h1, h1_off, B[None], G[None]), vec=((p + 1) * ne, v_off, v_ori, Bo[None], Go[None]))


@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("sdim", [2, 3])
def test_hdiv_and_two_space_forms_on_lines(sdim, p):
    from palace_amd import ceed

    S = _curve(sdim, p)
    geom, ogeom, qw = S["geom"], S["ogeom"], S["qw"]
    n, off, ori, Bo, Go = S["vec"]
    n_h1, h1_off, B, G = S["h1"]
    rng = np.random.default_rng(300 + 10 * p + sdim)
    cm, c_ns = _ctx(rng, sdim), _ctx(rng, sdim, sym=False)
    tag = "31" if sdim == 3 else "21"
    q = lambda name: getattr(ceed, "QF_%s_%s" % (name, tag))  # noqa: E731
    rtb = ceed.DenseBlock(ceed.FE_HDIV, n, off, Bo, Go, orients=ori)
    ndb = ceed.DenseBlock(ceed.FE_HCURL, n, off, Bo, None, orients=ori)
    h1b = ceed.DenseBlock(ceed.FE_H1, n_h1, h1_off, B, G)
    side = po.CeedOperatorOracle(n, off, ori, Bo, Go, ogeom, None, None, qw=qw)
    h1o = po.CeedOperatorOracle(n_h1, h1_off, None, B, G, ogeom, None, None, vector_fe=False)
    mass = ceed.Operator(n, n).add_dense_integrator(geom, rtb, q("HDIV"), cm.pack(), ceed.EVAL_INTERP).finalize()
    x, ym = _check_square(mass, po.CeedOperatorOracle(n, off, ori, Bo, Go, ogeom, po.QF_HDIV_LINE, cm, qw=qw), n, rng, "mass")
    pair = ceed.Operator(n, n).add_dense_integrator(geom, rtb, q("L2MASS"), np.concatenate([cm.pack(), C1.pack()]),
                                                    ceed.EVAL_INTERP | ceed.EVAL_DIV | ceed.EVAL_WEIGHT).finalize()
    _check_square(pair, po.CeedOperatorOracle(n, off, ori, Bo, Go, ogeom, po.QF_L2MASS, cm, C1, qw=qw), n, rng, "div-div + mass")
    # on a line the contravariant map is the unit tangent, the covariant one the tangent over its squared length: different D
    nd_mass = ceed.Operator(n, n).add_dense_integrator(geom, ndb, q("HCURL"), cm.pack(), ceed.EVAL_INTERP).finalize()
    assert np.abs(_mult(nd_mass, x, n) - ym).max() > 1e-3 * np.abs(ym).max()
    for qf, qfo, (tb, to, tt), (sb, so) in (
            (q("HCURLHDIV"), po.QF_HCURLHDIV_LINE, (ndb, side, None), (rtb, side)),
            (q("HDIVHCURL"), po.QF_HDIVHCURL_LINE, (rtb, side, None), (ndb, side)),
            (q("HCURL"), po.QF_HCURL_LINE, (h1b, h1o, G), (ndb, side)),            # MixedVectorGradient on a line (mixedvecgrad.cpp:78-112)
            (q("HCURLHDIV"), po.QF_HCURLHDIV_LINE, (h1b, h1o, G), (rtb, side))):
        op = ceed.Operator(sb.lsize, tb.lsize).add_dense_mixed_integrator(geom, tb, sb, qf, c_ns.pack()).finalize()
        xx, yy = rng.uniform(-1, 1, tb.lsize), rng.uniform(-1, 1, sb.lsize)
        ref = po.MixedSpaceOracle(to, so, ogeom, qfo, c_ns, first_tab=tt).apply_add(xx, np.zeros(sb.lsize))
        ax = _mult(op, xx, sb.lsize)
        assert np.abs(ax - ref).max() < REL * np.abs(ref).max(), (qfo, tb.fe_type)
        aty = _mult_t(op, yy, tb.lsize)
        assert abs(yy @ ax - aty @ xx) < 1e-11 * np.abs(yy).sum() * np.abs(ax).max(), qfo
    # GradientIntegrator on the curve: byVDIM ordering (component stride 1, offsets times the vector dimension)
    vtest = ceed.DenseBlock(ceed.FE_H1, sdim * n_h1, sdim * h1_off, B, None)
    op = ceed.Operator(sdim * n_h1, n_h1).add_dense_gradient_integrator(geom, h1b, vtest, 1, q("HCURLH1D"), c_ns.pack()).finalize()
    xx = rng.uniform(-1, 1, n_h1)
    ref = po.MixedSpaceOracle(h1o, h1o, ogeom, po.QF_HCURLH1D, c_ns, first_tab=G).gradient_add(xx, np.zeros(sdim * n_h1), n_h1)
    ref = ref.reshape(sdim, n_h1).T.ravel()  # the oracle numbers by nodes, the operator above by vector dimension
    assert np.abs(_mult(op, xx, sdim * n_h1) - ref).max() < REL * np.abs(ref).max()


@pytest.mark.parametrize("p", [1, 2])
@pytest.mark.parametrize("kind", ["tet4", "tet10"])
def test_divdiv_mass_and_gradient_on_tetrahedra(kind, p):
    """f_apply_l2mass_33 on Raviart-Thomas tetrahedra = H(div) mass + div-div (both checked on their own in tests/test_rt_gpu.py),
    and f_apply_hcurlh1d_33 on nodal tetrahedra."""
    from palace_amd import ceed
    from palace_amd.fem import rt, tet
    from tests.test_rt_gpu import _geom, _mesh

    mesh = _mesh(kind)
    sp = rt.RTTetSpace(mesh, p)
    pts, wts = tet.tet_quadrature(p + 1)
    interp, div = sp.elem.tables(pts)
    geom, ogeom = _geom(mesh, pts, wts)
    rng = np.random.default_rng(400 + p)
    cm = _ctx(rng, 3)
    n = sp.ndofs
    block = ceed.DenseBlock(ceed.FE_HDIV, n, sp.offsets, interp, div[None], orients=sp.orients)
    pair = ceed.Operator(n, n).add_dense_integrator(geom, block, ceed.QF_L2MASS_33, np.concatenate([cm.pack(), C1.pack()]),
                                                    ceed.EVAL_INTERP | ceed.EVAL_DIV | ceed.EVAL_WEIGHT).finalize()
    orc = po.CeedOperatorOracle(n, sp.offsets, sp.orients, interp, div, ogeom, po.QF_L2MASS, cm, C1, qw=wts, deriv_comps=1)
    x, yp = _check_square(pair, orc, n, rng, "div-div + mass")
    mass = ceed.Operator(n, n).add_dense_integrator(geom, block, ceed.QF_HDIV_33, cm.pack(), ceed.EVAL_INTERP).finalize()
    dd = ceed.Operator(n, n).add_dense_integrator(geom, block, ceed.QF_L2_1, C1.pack(), ceed.EVAL_DIV | ceed.EVAL_WEIGHT).finalize()
    assert np.abs(_mult(mass, x, n) + _mult(dd, x, n) - yp).max() < REL * np.abs(yp).max()
    # gradient form, byNODES
    h1 = tet.H1TetSpace(mesh, p)
    hint, hgrad = h1.elem.tables(pts)
    nh = h1.ndofs
    c_ns = _ctx(rng, 3, sym=False)
    h1b = ceed.DenseBlock(ceed.FE_H1, nh, h1.offsets, hint, hgrad)
    vtest = ceed.DenseBlock(ceed.FE_H1, 3 * nh, h1.offsets, hint, None)
    h1o = po.CeedOperatorOracle(nh, h1.offsets, None, hint, hgrad, ogeom, None, None, vector_fe=False)
    op = ceed.Operator(3 * nh, nh).add_dense_gradient_integrator(geom, h1b, vtest, nh, ceed.QF_HCURLH1D_33, c_ns.pack()).finalize()
    xx = rng.uniform(-1, 1, nh)
    ref = po.MixedSpaceOracle(h1o, h1o, ogeom, po.QF_HCURLH1D, c_ns, first_tab=hgrad).gradient_add(xx, np.zeros(3 * nh), nh)
    assert np.abs(_mult(op, xx, 3 * nh) - ref).max() < REL * np.abs(ref).max()
    # MixedVectorCurlIntegrator with an H(div) test space: (C curl u, v), f_apply_hdiv_33 between the Nedelec space (curl table) and
    # the Raviart-Thomas one; its transpose is the weak curl with an H(div) trial space
    nd = tet.NDTetSpace(mesh, p)
    nint, ncurl = nd.elem.tables(pts)
    kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
    ndb = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, nint, ncurl, **kw)
    rtb = ceed.DenseBlock(ceed.FE_HDIV, n, sp.offsets, interp, None, orients=sp.orients)
    okw = dict(curl_orients=nd.curl_orients) if not nd.diagonal_transform else {}
    ndo = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients if nd.diagonal_transform else None, nint, ncurl, ogeom, None, None, **okw)
    rto = po.CeedOperatorOracle(n, sp.offsets, sp.orients, interp, interp, ogeom, None, None)
    mc = ceed.Operator(n, nd.ndofs).add_dense_mixed_integrator(geom, ndb, rtb, ceed.QF_HDIV_33, c_ns.pack()).finalize()
    xa, yb = rng.uniform(-1, 1, nd.ndofs), rng.uniform(-1, 1, n)
    ref_c = po.MixedSpaceOracle(ndo, rto, ogeom, po.QF_HDIV, c_ns, first_tab=ndo.deriv).apply_add(xa, np.zeros(n))
    ax = _mult(mc, xa, n)
    assert np.abs(ax - ref_c).max() < REL * np.abs(ref_c).max()
    assert abs(yb @ ax - _mult_t(mc, yb, nd.ndofs) @ xa) < 1e-11 * np.abs(yb).sum() * np.abs(ax).max()
    # MassIntegrator on the vector space (f_apply_h1_3, non-symmetric 3 x 3 coefficient), byNODES, and its transpose
    vm = ceed.Operator(3 * nh, 3 * nh).add_dense_vector_mass_integrator(geom, vtest, 3, nh, c_ns.pack()).finalize()
    xv, yv = rng.uniform(-1, 1, 3 * nh), rng.uniform(-1, 1, 3 * nh)
    uq = np.einsum("qj,cej->ecq", hint.reshape(-1, hint.shape[-1]), xv.reshape(3, nh)[:, h1.offsets])
    vq = po.apply_h1_vec(c_ns, ogeom, uq)
    ref_m = np.zeros(3 * nh)
    for c in range(3):
        np.add.at(ref_m, (c * nh + h1.offsets).ravel(), np.einsum("qj,eq->ej", hint.reshape(-1, hint.shape[-1]), vq[:, c, :]).ravel())
    mv = _mult(vm, xv, 3 * nh)
    assert np.abs(mv - ref_m).max() < REL * np.abs(ref_m).max()
    assert abs(yv @ mv - _mult_t(vm, yv, 3 * nh) @ xv) < 1e-11 * np.abs(yv).sum() * np.abs(mv).max()
    # ... and its assembled form (pa_op_full_assemble: rectangular CSR, rows = the dofs of the vector test space)
    A = op.full_assemble(skip_zeros=True)
    assert A.shape == (3 * nh, nh) and np.abs(A @ xx - ref).max() < REL * np.abs(ref).max()
