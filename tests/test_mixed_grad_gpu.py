"""Two-space operators beyond the Nedelec <-> Raviart-Thomas mass of the 3-D flux estimators (tests/test_estimator_gpu.py):

* MixedVectorGradientIntegrator (C grad phi, v), H1 trial and H(curl) or H(div) test space (fem/integ/mixedvecgrad.cpp:43-76:
  f_apply_hcurl_33 | _22, f_apply_hcurlhdiv_33 | _22 on gradients) -- the `Atn` block of the boundary-mode eigenproblem
  (models/modeeigensolver.cpp:52);
* the plane members of the two-space QFunctions (qfunctions/22/hcurlhdiv_22_qf.h, hcurlhdiv_error_22_qf.h: the 2-D branch of
  the flux estimators, linalg/errorestimator.cpp:345-349) on the reference's cavity2d triangulation.

Against the oracle (pinned on the reference headers, tests/test_oracle_ref.py, tests/test_oracle_2d.py), and against an
identity that needs no oracle: grad(H1_p) lies in ND_p, so (C grad phi, v) = M_ND(C) G phi with the discrete gradient G."""
import os

import numpy as np
import pytest

from oracle import palace_oracle as po
from tests import util

pytestmark = pytest.mark.gpu
REL = 1e-12


def _warp(X):
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    return np.stack([x + 0.04 * np.sin(2 * y + z), y + 0.05 * x * z, z - 0.03 * np.cos(3 * x) * y], axis=1)


def _tet_blocks(kind, p):
    from palace_amd import ceed
    from palace_amd.fem import rt, tet

    m = tet.cube_tet_mesh(3)
    m.attr[:] = 1 + (np.arange(m.ne) % 2)
    if kind == "tet10":
        m2 = tet.to_quadratic(m, _warp)
        m2.attr[:] = m.attr
        m = m2
    nd, sp, h1 = tet.NDTetSpace(m, p), rt.RTTetSpace(m, p), tet.H1TetSpace(m, p)
    pts, wts = tet.tet_quadrature(p + 1)
    nint, ncurl = nd.elem.tables(pts)
    rint, _ = sp.elem.tables(pts)
    hint, hgrad = h1.elem.tables(pts)
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
    h1b = ceed.DenseBlock(ceed.FE_H1, h1.ndofs, h1.offsets, hint, hgrad)
    h1o = po.CeedOperatorOracle(h1.ndofs, h1.offsets, None, hint, hgrad, ogeom, po.QF_HCURL, None, vector_fe=False)
    return geom, ogeom, (h1b, h1o, hgrad), (ndb, ndo, nd), (rtb, rto)


def _mult(op, x, n):
    import torch

    y = torch.empty(n, dtype=torch.float64, device="cuda")
    op.mult(torch.from_numpy(np.ascontiguousarray(x)).cuda(), y)
    return y.cpu().numpy()


@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("kind", ["tet4", "tet10"])
def test_mixed_vector_gradient_tets(kind, p):
    from palace_amd import ceed

    geom, ogeom, (h1b, h1o, hgrad), (ndb, ndo, _), (rtb, rto) = _tet_blocks(kind, p)
    c_ns, b_ns = util.make_ctx("nonsym", 2)
    rng = np.random.default_rng(p)
    phi = rng.uniform(-1, 1, h1b.lsize)
    for qf, qfo, (tb, to) in ((ceed.QF_HCURL_33, po.QF_HCURL, (ndb, ndo)), (ceed.QF_HCURLHDIV_33, po.QF_HCURLHDIV, (rtb, rto))):
        op = ceed.Operator(tb.lsize, h1b.lsize).add_dense_mixed_integrator(geom, h1b, tb, qf, b_ns).finalize()
        ref = po.MixedSpaceOracle(h1o, to, ogeom, qfo, c_ns, first_tab=hgrad).apply_add(phi, np.zeros(tb.lsize))
        y = _mult(op, phi, tb.lsize)
        assert np.abs(y - ref).max() < REL * np.abs(ref).max(), qfo
        # constants have no gradient
        assert np.abs(_mult(op, np.ones(h1b.lsize), tb.lsize)).max() < 1e-12 * np.abs(ref).max()
    # the transposed pairing with an H1 test space: (C u, grad psi) for u in ND (f_apply_hcurl_33, Interp -> Grad)
    op = ceed.Operator(h1b.lsize, ndb.lsize).add_dense_mixed_integrator(geom, ndb, h1b, ceed.QF_HCURL_33, b_ns).finalize()
    u = rng.uniform(-1, 1, ndb.lsize)
    ref = po.MixedSpaceOracle(ndo, h1o, ogeom, po.QF_HCURL, c_ns, second_tab=hgrad).apply_add(u, np.zeros(h1b.lsize))
    assert np.abs(_mult(op, u, h1b.lsize) - ref).max() < REL * np.abs(ref).max()


@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("kind", ["tet4", "tet10"])
def test_mixed_vector_gradient_equals_mass_times_discrete_gradient(kind, p):
    """grad(H1_p) is a subspace of ND_p: (C grad phi, v_i) = sum_j M_ij (G phi)_j with the ND mass matrix of the same
    coefficient and the discrete gradient (the element matrix of basis.cpp:139-143) -- no oracle involved."""
    import torch

    from palace_amd import ceed, linalg
    from palace_amd.fem import tet

    geom, ogeom, (h1b, h1o, hgrad), (ndb, ndo, nd), _ = _tet_blocks(kind, p)
    _, blob = util.make_ctx("nonsym", 2)
    ctx = linalg.Context()
    mixed = ceed.Operator(ndb.lsize, h1b.lsize).add_dense_mixed_integrator(geom, h1b, ndb, ceed.QF_HCURL_33, blob).finalize()
    mass = ceed.Operator(ndb.lsize, ndb.lsize).add_dense_integrator(geom, ndb, ceed.QF_HCURL_33, blob, ceed.EVAL_INTERP).finalize()
    # the range side of an interpolator takes the dual-inverse dof transformation (restriction.cpp:318-336)
    G = linalg.DenseInterp(ctx, dict(offsets=h1b.offsets, lsize=h1b.lsize), nd.restriction(interp_range=True),
                           tet.tet_gradient_matrix(p))
    phi = torch.from_numpy(np.random.default_rng(7 + p).uniform(-1, 1, h1b.lsize)).cuda()
    gphi = torch.empty(ndb.lsize, dtype=torch.float64, device="cuda")
    G.mult(phi, gphi)
    lhs = torch.empty_like(gphi)
    rhs = torch.empty_like(gphi)
    mixed.mult(phi, lhs)
    mass.mult(gphi, rhs)
    assert float((lhs - rhs).abs().max()) < 1e-11 * float(rhs.abs().max())


def _mult_t(op, x, n):
    import torch

    y = torch.empty(n, dtype=torch.float64, device="cuda")
    op.mult_transpose(torch.from_numpy(np.ascontiguousarray(x)).cuda(), y)
    return y.cpu().numpy()


@pytest.mark.parametrize("p", [1, 2])
@pytest.mark.parametrize("kind", ["tet4", "tet10"])
def test_two_space_transposes(kind, p):
    """pa_op_mult_transpose of two-space operators (Btn = -Atn^T of models/modeeigensolver.cpp:410-418 without assembling):
    against the oracle operator with trial and test exchanged, the paired QFunction and the transposed coefficient, and
    <A x, y> = <x, A^T y> on the device."""
    from palace_amd import ceed

    geom, ogeom, (h1b, h1o, hgrad), (ndb, ndo, _), (rtb, rto) = _tet_blocks(kind, p)
    c_ns, b_ns = util.make_ctx("nonsym", 2)
    import copy

    c_t = copy.deepcopy(c_ns)
    c_t.mat = c_ns.mat.reshape(-1, 3, 3).transpose(0, 2, 1).reshape(-1, 9).copy()
    rng = np.random.default_rng(80 + p)
    cases = ((ceed.QF_HCURL_33, (h1b, h1o, hgrad), (ndb, ndo, None), po.QF_HCURL),          # Atn and its transpose
             (ceed.QF_HCURLHDIV_33, (ndb, ndo, None), (rtb, rto, None), po.QF_HDIVHCURL),    # Flux of the Grad estimator
             (ceed.QF_HDIVHCURL_33, (rtb, rto, None), (ndb, ndo, None), po.QF_HCURLHDIV))
    for qf, (tb, to, tt), (sb, so, st), qfo_t in cases:
        op = ceed.Operator(sb.lsize, tb.lsize).add_dense_mixed_integrator(geom, tb, sb, qf, b_ns).finalize()
        x, y = rng.uniform(-1, 1, tb.lsize), rng.uniform(-1, 1, sb.lsize)
        ref = po.MixedSpaceOracle(so, to, ogeom, qfo_t, c_t, first_tab=st, second_tab=tt).apply_add(y, np.zeros(tb.lsize))
        got = _mult_t(op, y, tb.lsize)
        assert np.abs(got - ref).max() < REL * np.abs(ref).max(), qfo_t
        lhs, rhs = _mult(op, x, sb.lsize) @ y, x @ got
        assert abs(lhs - rhs) < 1e-12 * (np.abs(ref).max() * np.abs(x).sum())


@pytest.mark.parametrize("p", [1, 2])
def test_two_space_full_assembly(p):
    """BilinearForm(h1, nd)::FullAssemble of Atn (models/modeeigensolver.cpp:45-56): the rectangular CSR matrix of a two-space
    operator by coloured probing -- every entry is what the apply produces; against the apply, its transpose and the oracle."""
    from palace_amd import ceed

    geom, ogeom, (h1b, h1o, hgrad), (ndb, ndo, _), (rtb, rto) = _tet_blocks("tet10", p)
    c_ns, b_ns = util.make_ctx("nonsym", 2)
    rng = np.random.default_rng(90 + p)
    for qf, qfo, (tb, to, tt), (sb, so) in ((ceed.QF_HCURL_33, po.QF_HCURL, (h1b, h1o, hgrad), (ndb, ndo)),
                                            (ceed.QF_HCURLHDIV_33, po.QF_HCURLHDIV, (ndb, ndo, None), (rtb, rto))):
        op = ceed.Operator(sb.lsize, tb.lsize).add_dense_mixed_integrator(geom, tb, sb, qf, b_ns).finalize()
        A = op.full_assemble()
        assert A.shape == (sb.lsize, tb.lsize)
        x, y = rng.uniform(-1, 1, tb.lsize), rng.uniform(-1, 1, sb.lsize)
        ref = po.MixedSpaceOracle(to, so, ogeom, qfo, c_ns, first_tab=tt).apply_add(x, np.zeros(sb.lsize))
        assert np.abs(A @ x - ref).max() < REL * np.abs(ref).max()
        assert np.abs(A @ x - _mult(op, x, sb.lsize)).max() < 1e-13 * np.abs(ref).max()
        aty = _mult_t(op, y, tb.lsize)
        assert np.abs(A.T @ y - aty).max() < 1e-12 * np.abs(aty).max()
        As = op.full_assemble(skip_zeros=True)
        assert As.nnz <= A.nnz and abs(As - A).max() == 0.0


# ---- plane elements -----------------------------------------------------------------------------------------------------


def _tri_blocks(p):
    """ND_p, H1_p and the rotated-Nedelec form of RT_{p-1} on triangles: u_RT = R u_ND with R = [[0, 1], [-1, 0]] in reference
    coordinates maps tangential to normal moments, and J R = det(J) R J^-T makes the contravariant image of R u the rotated
    covariant image of u -- a conforming H(div) space with the restriction of the Nedelec one."""
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
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
    J = mesh.jacobians(pts)
    ogeom = po.build_geom_factor_22(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 4))
    ndb = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, nint, ncurl, orients=nd.orients)
    ndo = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, nint, ncurl, ogeom, po.QF_HCURL_22, None, qw=wts)
    rtb = ceed.DenseBlock(ceed.FE_HDIV, nd.ndofs, nd.offsets, rint, None, orients=nd.orients)
    rto = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, rint, ncurl, ogeom, po.QF_HDIV_22, None, qw=wts)
    h1b = ceed.DenseBlock(ceed.FE_H1, h1.ndofs, h1.offsets, hint, hgrad)
    h1o = po.CeedOperatorOracle(h1.ndofs, h1.offsets, None, hint, hgrad, ogeom, po.QF_HCURL_22, None, vector_fe=False)
    return geom, ogeom, (h1b, h1o, hgrad), (ndb, ndo), (rtb, rto)


def _ctx22(rng, sym=False):
    A = rng.uniform(-1, 1, (2, 2))
    return po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[(A @ A.T if sym else A) + 2 * np.eye(2), np.array([0.6])], a=1.2, dim=2)


@pytest.mark.parametrize("p", [1, 2, 3])
def test_two_space_operators_on_triangles(p):
    import torch

    from palace_amd import ceed

    geom, ogeom, (h1b, h1o, hgrad), (ndb, ndo), (rtb, rto) = _tri_blocks(p)
    rng = np.random.default_rng(20 + p)
    c_ns, c_an = _ctx22(rng), _ctx22(rng, sym=True)
    for qf, qfo, (tb, to, tt), (sb, so, st) in (
            (ceed.QF_HCURLHDIV_22, po.QF_HCURLHDIV_22, (ndb, ndo, None), (rtb, rto, None)),
            (ceed.QF_HDIVHCURL_22, po.QF_HDIVHCURL_22, (rtb, rto, None), (ndb, ndo, None)),
            (ceed.QF_HCURL_22, po.QF_HCURL_22, (h1b, h1o, hgrad), (ndb, ndo, None)),        # MixedVectorGradient, modeeigensolver.cpp:52
            (ceed.QF_HCURLHDIV_22, po.QF_HCURLHDIV_22, (h1b, h1o, hgrad), (rtb, rto, None))):  # its H(div)-test form
        op = ceed.Operator(sb.lsize, tb.lsize).add_dense_mixed_integrator(geom, tb, sb, qf, c_ns.pack()).finalize()
        x = rng.uniform(-1, 1, tb.lsize)
        ref = po.MixedSpaceOracle(to, so, ogeom, qfo, c_ns, first_tab=tt, second_tab=st).apply_add(x, np.zeros(sb.lsize))
        assert np.abs(_mult(op, x, sb.lsize) - ref).max() < REL * np.abs(ref).max(), (qfo, tb.fe_type)
    pair = np.concatenate([c_an.pack(), c_ns.pack()])
    for qf, qfo, (b1, o1), (b2, o2) in ((ceed.QF_HCURLHDIV_ERROR_22, po.QF_HCURLHDIV_ERROR_22, (ndb, ndo), (rtb, rto)),
                                        (ceed.QF_HDIVHCURL_ERROR_22, po.QF_HDIVHCURL_ERROR_22, (rtb, rto), (ndb, ndo))):
        integ = ceed.ElementErrorIntegrator(geom, b1, b2, qf, pair)
        u1, u2 = rng.uniform(-1, 1, b1.lsize), rng.uniform(-1, 1, b2.lsize)
        e0 = rng.uniform(0, 1, integ.ne)
        ref = po.MixedSpaceOracle(o1, o2, ogeom, qfo, c_an, c_ns).error_add(u1, u2, e0.copy())
        est = torch.from_numpy(e0.copy()).cuda()
        integ.apply_add(torch.from_numpy(u1).cuda(), torch.from_numpy(u2).cuda(), est)
        assert np.abs(est.cpu().numpy() - ref).max() < REL * np.abs(ref).max(), qfo
        assert (ref - e0).min() > 0


@pytest.mark.parametrize("p", [1, 2, 3])
def test_plane_hdiv_mass(p):
    """f_apply_hdiv_22 through the dense path (FE_HDIV block, Interp): apply and diagonal against the oracle."""
    import torch

    from palace_amd import ceed

    geom, ogeom, _, _, (rtb, rto) = _tri_blocks(p)
    rng = np.random.default_rng(40 + p)
    c = _ctx22(rng, sym=True)
    op = ceed.Operator(rtb.lsize, rtb.lsize).add_dense_integrator(geom, rtb, ceed.QF_HDIV_22, c.pack(), ceed.EVAL_INTERP).finalize()
    orc = po.CeedOperatorOracle(rto.lsize, rto.off, rto.sgn < 0, rto.interp, rto.deriv, ogeom, po.QF_HDIV_22, c)
    x = rng.uniform(-1, 1, rtb.lsize)
    ref = orc.apply_add(x, np.zeros(rtb.lsize))
    assert np.abs(_mult(op, x, rtb.lsize) - ref).max() < REL * np.abs(ref).max()
    d = torch.empty(rtb.lsize, dtype=torch.float64, device="cuda")
    op.assemble_diagonal(d)
    dref = orc.diagonal()
    assert np.abs(d.cpu().numpy() - dref).max() < REL * np.abs(dref).max()


def test_plane_flux_estimate():
    """The 2-D branch of GradFluxErrorEstimator (errorestimator.cpp:345-349) with the library's pieces: D = M_RT^-1 Flux E by PCG
    + Jacobi, eta_e^2 = int_e |D - E|^2 (unit material).  Against the same through the oracle operators with a dense solve."""
    import torch

    from palace_amd import ceed, linalg

    p = 1
    geom, ogeom, _, (ndb, ndo), (rtb, rto) = _tri_blocks(p)
    c_id = po.CoeffCtx(dim=2)
    ctx = linalg.Context()
    flux = ceed.Operator(rtb.lsize, ndb.lsize).add_dense_mixed_integrator(geom, ndb, rtb, ceed.QF_HCURLHDIV_22, c_id.pack()).finalize()
    # the Raviart-Thomas mass matrix (f_apply_hdiv_22).  With the rotated tables and R^T (adjJt^T adjJt) R = J^T J / detJ^2 it
    # equals the Nedelec mass matrix of the plain tables: checked below
    mass = ceed.Operator(rtb.lsize, rtb.lsize).add_dense_integrator(geom, rtb, ceed.QF_HDIV_22, c_id.pack(), ceed.EVAL_INTERP).finalize()
    mass_nd = ceed.Operator(ndb.lsize, ndb.lsize).add_dense_integrator(geom, ndb, ceed.QF_HCURL_22, c_id.pack(), ceed.EVAL_INTERP).finalize()
    rng = np.random.default_rng(5)
    E = rng.uniform(-1, 1, ndb.lsize)
    # oracle side; M_RT = sum_q w detJ (J u / detJ) . (J v / detJ) column by column
    rhs_o = po.MixedSpaceOracle(ndo, rto, ogeom, po.QF_HCURLHDIV_22, c_id).apply_add(E, np.zeros(rto.lsize))
    A = ogeom[:, 2:6, :]
    Jl = np.stack([A[:, 3], -A[:, 2], -A[:, 1], A[:, 0]], axis=1)  # J / detJ, column-major
    Md = np.zeros((rto.lsize, rto.lsize))
    for j in range(rto.lsize):
        ej = np.zeros(rto.lsize)
        ej[j] = 1.0
        uq = np.einsum("dqj,ej->edq", rto.interp, rto._restrict(ej, slice(None)))
        pu = np.stack([Jl[:, 0] * uq[:, 0] + Jl[:, 2] * uq[:, 1], Jl[:, 1] * uq[:, 0] + Jl[:, 3] * uq[:, 1]], axis=1)
        vq = ogeom[:, 1][:, None, :] * np.stack([Jl[:, 0] * pu[:, 0] + Jl[:, 1] * pu[:, 1],
                                                  Jl[:, 2] * pu[:, 0] + Jl[:, 3] * pu[:, 1]], axis=1)
        ve = np.einsum("dqj,edq->ej", rto.interp, vq)
        np.add.at(Md[:, j], rto.off.ravel(), rto._restrict_t(ve, slice(None)).ravel())
    D_o = np.linalg.solve(Md, rhs_o)
    est_o = po.MixedSpaceOracle(ndo, rto, ogeom, po.QF_HCURLHDIV_ERROR_22, c_id, c_id).error_add(E, D_o, np.zeros(ndo.NE))
    x = rng.uniform(-1, 1, ndb.lsize)
    assert np.abs(_mult(mass, x, ndb.lsize) - Md @ x).max() < 1e-11 * np.abs(Md @ x).max()
    assert np.abs(_mult(mass_nd, x, ndb.lsize) - Md @ x).max() < 1e-11 * np.abs(Md @ x).max()
    M = linalg.ParOperator(ctx, mass, np.zeros(0, np.int32), linalg.DIAG_ONE)
    cg = linalg.cg(ctx, M, linalg.jacobi(ctx, M), rel_tol=1e-13, max_it=2000)
    Ed = torch.from_numpy(E).cuda()
    rhs = torch.empty(rtb.lsize, dtype=torch.float64, device="cuda")
    flux.mult(Ed, rhs)
    D = torch.zeros_like(rhs)
    cg.mult(rhs, D)
    integ = ceed.ElementErrorIntegrator(geom, ndb, rtb, ceed.QF_HCURLHDIV_ERROR_22, np.concatenate([c_id.pack(), c_id.pack()]))
    est = torch.zeros(integ.ne, dtype=torch.float64, device="cuda")
    integ.apply_add(Ed, D, est)
    assert np.abs(D.cpu().numpy() - D_o).max() < 1e-8 * np.abs(D_o).max()
    assert np.abs(est.cpu().numpy() - est_o).max() < 1e-8 * est_o.max()
    assert est_o.min() > 0


def _scalar_pair(kind, p):
    """A discontinuous scalar space (the L2 space the scalar curl of a plane Nedelec field lives in; here with the nodal
    basis of order p on every element and no sharing) and the continuous H1 space of order p, on triangles or tetrahedra."""
    from palace_amd import ceed

    if kind == "tri":
        geom, ogeom, (h1b, h1o, hgrad), _, _ = _tri_blocks(p)
    else:
        geom, ogeom, (h1b, h1o, hgrad), _, _ = _tet_blocks(kind, p)
    ne, P = h1b.offsets.shape
    off = np.arange(ne * P, dtype=np.int32).reshape(ne, P)
    l2b = ceed.DenseBlock(ceed.FE_H1, ne * P, off, h1b.interp, h1b.deriv)
    l2o = po.CeedOperatorOracle(ne * P, off, None, h1b.interp, h1b.deriv, ogeom, po.QF_H1MASS, None, vector_fe=False)
    return geom, ogeom, (l2b, l2o), (h1b, h1o)


@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("kind", ["tri", "tet10"])
def test_scalar_flux_pair(kind, p):
    """The scalar branch of the flux estimators (2-D curl: errorestimator.cpp:122-160, :446-474): MassIntegrator from a
    discontinuous scalar space into H1 (f_apply_h1_1 between two spaces) and the element error f_apply_l2h1_error."""
    import torch

    from palace_amd import ceed

    geom, ogeom, (l2b, l2o), (h1b, h1o) = _scalar_pair(kind, p)
    rng = np.random.default_rng(60 + p)
    c1 = po.CoeffCtx(attr_mat=[1, 0], mat_coeff=[np.array([1.9]), np.array([0.4])], dim=1)
    c2 = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[np.array([0.9]), np.array([1.3])], dim=1)
    op = ceed.Operator(h1b.lsize, l2b.lsize).add_dense_mixed_integrator(geom, l2b, h1b, ceed.QF_H1_1, c1.pack()).finalize()
    x = rng.uniform(-1, 1, l2b.lsize)
    ref = po.MixedSpaceOracle(l2o, h1o, ogeom, po.QF_H1MASS, c1).apply_add(x, np.zeros(h1b.lsize))
    assert np.abs(_mult(op, x, h1b.lsize) - ref).max() < REL * np.abs(ref).max()
    # the continuous field embedded in the discontinuous space gives the H1 mass operator
    m = ceed.Operator(h1b.lsize, h1b.lsize).add_dense_integrator(geom, h1b, ceed.QF_H1_1, c1.pack(), ceed.EVAL_INTERP).finalize()
    xc = rng.uniform(-1, 1, h1b.lsize)
    ym = _mult(m, xc, h1b.lsize)
    assert np.abs(_mult(op, xc[h1b.offsets].ravel(), h1b.lsize) - ym).max() < 1e-11 * np.abs(ym).max()
    integ = ceed.ElementErrorIntegrator(geom, l2b, h1b, ceed.QF_L2H1_ERROR, np.concatenate([c1.pack(), c2.pack()]))
    u2 = rng.uniform(-1, 1, h1b.lsize)
    e0 = rng.uniform(0, 1, integ.ne)
    ref = po.MixedSpaceOracle(l2o, h1o, ogeom, po.QF_L2H1_ERROR, c1, c2).error_add(x, u2, e0.copy())
    est = torch.from_numpy(e0.copy()).cuda()
    integ.apply_add(torch.from_numpy(x).cuda(), torch.from_numpy(u2).cuda(), est)
    assert np.abs(est.cpu().numpy() - ref).max() < REL * np.abs(ref).max()
    assert (ref - e0).min() > 0
    # equal fields with equal coefficients: no error
    est.zero_()
    same = ceed.ElementErrorIntegrator(geom, l2b, h1b, ceed.QF_L2H1_ERROR, np.concatenate([c1.pack(), c1.pack()]))
    same.apply_add(torch.from_numpy(xc[h1b.offsets].ravel().copy()).cuda(), torch.from_numpy(xc).cuda(), est)
    assert float(est.abs().max()) < 1e-24 + 1e-26 * float(np.abs(ref).max())


def test_two_space_argument_checks_2d():
    from palace_amd import ceed
    from palace_amd.lib import PalaceAmdError

    geom, ogeom, (h1b, h1o, hgrad), (ndb, ndo), (rtb, rto) = _tri_blocks(1)
    blob = po.CoeffCtx(dim=2).pack()
    with pytest.raises(PalaceAmdError, match="dimension"):  # a _33 QFunction on plane geometry data
        ceed.Operator(rtb.lsize, ndb.lsize).add_dense_mixed_integrator(geom, ndb, rtb, ceed.QF_HCURLHDIV_33, blob)
    with pytest.raises(PalaceAmdError, match="element types"):  # H1 gradients are covariant: not an H(div) side
        ceed.Operator(ndb.lsize, h1b.lsize).add_dense_mixed_integrator(geom, h1b, ndb, ceed.QF_HDIVHCURL_22, blob)
