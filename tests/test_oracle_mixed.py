"""CPU checks of the two-space oracle operators (oracle/palace_oracle.py: MixedSpaceOracle) through identities that involve no
device code: they guard the checker the `-m gpu` parity tests of tests/test_mixed_grad_gpu.py and tests/test_estimator_gpu.py
lean on (the pointwise QFunctions themselves are pinned on the reference headers in tests/test_oracle_ref.py and
tests/test_oracle_2d.py)."""
import copy

import numpy as np
import pytest

from oracle import palace_oracle as po
from palace_amd.fem import rt, tet
from tests import util


def _warp(X):
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    return np.stack([x + 0.04 * np.sin(2 * y + z), y + 0.05 * x * z, z - 0.03 * np.cos(3 * x) * y], axis=1)


def _spaces(p, curved):
    m = tet.cube_tet_mesh(2)
    m.attr[:] = 1 + (np.arange(m.ne) % 2)
    if curved:
        m2 = tet.to_quadratic(m, _warp)
        m2.attr[:] = m.attr
        m = m2
    nd, sp, h1 = tet.NDTetSpace(m, p), rt.RTTetSpace(m, p), tet.H1TetSpace(m, p)
    pts, wts = tet.tet_quadrature(p + 1)
    nint, ncurl = nd.elem.tables(pts)
    rint, _ = sp.elem.tables(pts)
    hint, hgrad = h1.elem.tables(pts)
    J = m.jacobians(pts)
    og = po.build_geom_factor_33(m.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(m.ne, -1, 9))
    kw = dict(curl_orients=nd.curl_orients) if not nd.diagonal_transform else {}
    ndo = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients if nd.diagonal_transform else None, nint, ncurl, og, po.QF_HCURL,
                                None, **kw)
    rto = po.CeedOperatorOracle(sp.ndofs, sp.offsets, sp.orients, rint, rint, og, po.QF_HDIV, None)
    h1o = po.CeedOperatorOracle(h1.ndofs, h1.offsets, None, hint, hgrad, og, po.QF_HCURL, None, vector_fe=False)
    return m, nd, og, ndo, rto, h1o, hgrad


@pytest.mark.parametrize("curved", [False, True])
@pytest.mark.parametrize("p", [1, 2])
def test_mixed_gradient_is_mass_times_discrete_gradient(p, curved):
    """grad(H1_p) lies in ND_p: (C grad phi, v_i) = sum_j M_ij(C) (G phi)_j with the element gradient matrix of
    basis.cpp:139-143 pushed through the restrictions (dual-inverse dof transformation on the range side)."""
    m, nd, og, ndo, rto, h1o, hgrad = _spaces(p, curved)
    c, _ = util.make_ctx("nonsym", 2)
    phi = np.random.default_rng(p).uniform(-1, 1, h1o.lsize)
    lhs = po.MixedSpaceOracle(h1o, ndo, og, po.QF_HCURL, c, first_tab=hgrad).apply_add(phi, np.zeros(ndo.lsize))
    # G phi: element values G_e phi_e are the ND dofs in the element's own orientation; undo the restriction transformation
    Ge = tet.tet_gradient_matrix(p)
    ue = phi[h1o.off] @ Ge.T  # [ne, P_nd] = T_e x_e
    x = np.zeros(ndo.lsize)
    if ndo.cor is None:
        x[ndo.off] = ue * (1.0 if ndo.sgn is None else ndo.sgn)
    else:  # x_e = T_e^-1 u_e, element by element (tridiagonal T_e as dense blocks)
        t = ndo.cor
        for e in range(ndo.NE):
            Te = np.diag(t[e, :, 1]) + np.diag(t[e, 1:, 0], -1) + np.diag(t[e, :-1, 2], 1)
            x[ndo.off[e]] = np.linalg.solve(Te, ue[e])
    M = po.CeedOperatorOracle(ndo.lsize, ndo.off, None if ndo.sgn is None else ndo.sgn < 0, ndo.interp, ndo.deriv, og, po.QF_HCURL, c,
                              curl_orients=None if ndo.cor is None else ndo.cor.astype(np.int8))
    rhs = M.apply_add(x, np.zeros(ndo.lsize))
    assert np.abs(lhs - rhs).max() < 1e-12 * np.abs(rhs).max()
    # constants have no gradient
    zero = po.MixedSpaceOracle(h1o, ndo, og, po.QF_HCURL, c, first_tab=hgrad).apply_add(np.ones(h1o.lsize), np.zeros(ndo.lsize))
    assert np.abs(zero).max() < 1e-13 * np.abs(lhs).max()


@pytest.mark.parametrize("p", [1, 2])
def test_two_space_pairs_are_transposes(p):
    """(hcurlhdiv, C) and (hdivhcurl, C^T) with trial and test exchanged are each other's transposes; hcurl is its own."""
    m, nd, og, ndo, rto, h1o, hgrad = _spaces(p, True)
    c, _ = util.make_ctx("nonsym", 2)
    ct = copy.deepcopy(c)
    ct.mat = c.mat.reshape(-1, 3, 3).transpose(0, 2, 1).reshape(-1, 9).copy()
    rng = np.random.default_rng(10 + p)
    for qf, (to, tt), (so, st), qft in ((po.QF_HCURL, (h1o, hgrad), (ndo, None), po.QF_HCURL),
                                        (po.QF_HCURLHDIV, (ndo, None), (rto, None), po.QF_HDIVHCURL),
                                        (po.QF_HDIVHCURL, (rto, None), (ndo, None), po.QF_HCURLHDIV)):
        x, y = rng.uniform(-1, 1, to.lsize), rng.uniform(-1, 1, so.lsize)
        Ax = po.MixedSpaceOracle(to, so, og, qf, c, first_tab=tt, second_tab=st).apply_add(x, np.zeros(so.lsize))
        Aty = po.MixedSpaceOracle(so, to, og, qft, ct, first_tab=st, second_tab=tt).apply_add(y, np.zeros(to.lsize))
        assert abs(Ax @ y - x @ Aty) < 1e-13 * abs(Ax @ y)


def test_error_integrand_vanishes_for_matching_fluxes():
    """eta_e = 0 when the second field is the exact flux: with C1 = C2 = I and u2 the RT interpolant of a constant vector field
    that also is u1 in ND (both spaces hold constants on straight elements), the integrand |J u2 / detJ - adjJt u1|^2 is zero."""
    m, nd, og, ndo, rto, h1o, hgrad = _spaces(1, False)
    cid = po.CoeffCtx()
    cvec = np.array([0.7, -0.4, 1.1])
    # L2 projections of the constant onto ND_1 and RT_1 reproduce it
    def project(o, tab, piola_rows):
        Mo = np.stack([o.apply_add(e, np.zeros(o.lsize)) for e in np.eye(o.lsize)], axis=1)
        le = np.einsum("dqj,eq,eqd->ej", tab, og[:, 1, :], piola_rows)
        load = np.zeros(o.lsize)
        np.add.at(load, o.off.ravel(), o._restrict_t(le, slice(None)).ravel())
        return np.linalg.solve(Mo, load)

    A = og[:, 2:, :].reshape(og.shape[0], 3, 3, -1)  # adjJt [e, column, row, q]
    Jl = po.adjJt33(np.transpose(og[:, 2:, :], (0, 2, 1)))[0].reshape(og.shape[0], -1, 3, 3)  # J / detJ [e, q, column, row]
    nd_mass = po.CeedOperatorOracle(ndo.lsize, ndo.off, ndo.sgn < 0, ndo.interp, ndo.deriv, og, po.QF_HCURL, cid)
    rt_mass = po.CeedOperatorOracle(rto.lsize, rto.off, rto.sgn < 0, rto.interp, rto.deriv, og, po.QF_HDIV, cid)
    u1 = project(nd_mass, ndo.interp, np.einsum("ejiq,i->eqj", A, cvec))
    u2 = project(rt_mass, rto.interp, np.einsum("eqji,i->eqj", Jl, cvec))
    est = po.MixedSpaceOracle(ndo, rto, og, po.QF_HCURLHDIV_ERROR, cid, cid).error_add(u1, u2, np.zeros(ndo.NE))
    assert est.max() < 1e-24
