"""2-D path (the reference's examples/cavity2d, order-2 Nedelec triangles): the oracle's restatement of
fem/qfunctions/22/*.h and fem/qfunctions/1/l2_1_qf.h against vectors produced by the reference's own
headers, and the whole chain against the reference's regression eigenfrequencies
(test/data/regression/ref/cavity2d/eigenmode/eig.csv; the reference gates them at rtol 1e-4)."""
import os

import numpy as np
import pytest

from oracle import palace_oracle as po

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "qf2d_golden.npz"))
TOL = 1e-13


def _ctx(blob, dim):
    iv = np.asarray(blob).view(np.int32).reshape(-1, 2)[:, 0]
    nattr = int(iv[0])
    nmat = int(iv[1 + nattr])
    c = po.CoeffCtx(dim=dim)
    c.attr_mat = iv[1 : 1 + nattr].astype(np.int32)
    c.mat = np.asarray(blob)[2 + nattr : 2 + nattr + nmat * dim * dim].reshape(nmat, dim * dim)
    return c, 2 + nattr + nmat * dim * dim


def test_geom_factor_22():
    geom = po.build_geom_factor_22(np.ones(1), G["qw"], G["J"].T[None])
    np.testing.assert_allclose(geom[0, 1:], G["geom"][1:], rtol=TOL, atol=TOL)


def test_qfunctions_22():
    geom = G["geom"][None]
    c2, n2 = _ctx(G["ctx2"], 2)
    c1, _ = _ctx(G["ctx1"], 1)
    np.testing.assert_allclose(po.apply_hcurl_22(c2, geom, G["u"][None])[0], G["hcurl_22"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(po.apply_l2_1(c1, geom, G["qw"], G["cu"][None])[0], G["l2_1"], rtol=TOL, atol=TOL)
    pm, n = _ctx(G["ctx_pair"], 2)
    pc, _ = _ctx(G["ctx_pair"][n:], 1)
    v, cv = po.apply_hdivmass_22(pm, pc, geom, G["qw"], G["u"][None], G["cu"][None])
    np.testing.assert_allclose(v[0], G["hdivmass_22_v"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(cv[0], G["hdivmass_22_cv"], rtol=TOL, atol=TOL)


def test_boundary_qfunctions_32():
    geom = po.build_geom_factor_32(np.ones(1), G["qw"], G["J32"].T[None])
    np.testing.assert_allclose(geom[0, 1:], G["geom32"][1:], rtol=1e-12, atol=1e-13)
    c3, _ = _ctx(G["ctx3"], 3)
    v = po.apply_hcurl_32(c3, G["geom32"][None], G["u"][None])[0]
    np.testing.assert_allclose(v, G["hcurl_32"], rtol=1e-12, atol=1e-13)


def test_pair_qfunctions_22_32():
    """f_apply_hcurlmass_22, f_apply_hdivmass_32, f_apply_hcurlmass_32 (2-D / boundary H1 diffusion + mass, boundary ND
    curl-curl + mass) against the reference headers: pair contexts of different dimensions, the scalar one first or second."""
    c1, _ = _ctx(G["ctx1"], 1)
    c2, _ = _ctx(G["ctx2"], 2)
    c3, _ = _ctx(G["ctx3"], 3)
    u, cu = G["u"][None], G["cu"][None]
    v, gv = po.apply_hcurlmass_22(c1, c2, G["geom"][None], cu, u)
    np.testing.assert_allclose(v[0], G["hcurlmass_22_v"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(gv[0], G["hcurlmass_22_gv"], rtol=TOL, atol=TOL)
    v, cv = po.apply_hdivmass_32(c3, c1, G["geom32"][None], G["qw"], u, cu)
    np.testing.assert_allclose(v[0], G["hdivmass_32_v"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(cv[0], G["hdivmass_32_cv"], rtol=1e-12, atol=1e-13)
    v, gv = po.apply_hcurlmass_32(c1, c3, G["geom32"][None], cu, u)
    np.testing.assert_allclose(v[0], G["hcurlmass_32_v"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(gv[0], G["hcurlmass_32_gv"], rtol=1e-12, atol=1e-13)


def test_two_space_qfunctions_22():
    """f_apply_hcurlhdiv_22 / f_apply_hdivhcurl_22 and the error integrands f_apply_hcurlhdiv_error_22 /
    f_apply_hdivhcurl_error_22 (2-D flux projection and estimators, linalg/errorestimator.cpp:345-349) against the reference
    headers, with a non-symmetric coefficient."""
    geom = G["geom"][None]
    cn, _ = _ctx(G["ctx2n"], 2)
    c2, _ = _ctx(G["ctx2"], 2)
    u, u2 = G["u"][None], G["u2"][None]
    np.testing.assert_allclose(po.apply_hcurlhdiv_22(cn, geom, u)[0], G["hcurlhdiv_22"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(po.apply_hdivhcurl_22(cn, geom, u)[0], G["hdivhcurl_22"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(po.apply_hdiv_22(cn, geom, u)[0], G["hdiv_22"], rtol=TOL, atol=TOL)
    assert not np.allclose(G["hcurlhdiv_22"], G["hdivhcurl_22"])
    np.testing.assert_allclose(po.apply_hcurlhdiv_error_22(cn, c2, geom, u, u2)[0], G["hcurlhdiv_error_22"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(po.apply_hdivhcurl_error_22(cn, c2, geom, u, u2)[0], G["hdivhcurl_error_22"], rtol=TOL, atol=TOL)
    assert G["hcurlhdiv_error_22"].min() > 0 and not np.allclose(G["hcurlhdiv_error_22"], G["hdivhcurl_error_22"])


def test_scalar_error_qfunction():
    """f_apply_l2h1_error (l2h1_error_qf.h: the 2-D curl flux estimator's integrand) against the reference header."""
    c1, _ = _ctx(G["ctx1"], 1)
    c1b, _ = _ctx(G["ctx1b"], 1)
    got = po.apply_l2h1_error(c1, c1b, G["geom"][None], G["cu"][None], G["cu2"][None])[0]
    np.testing.assert_allclose(got, G["l2h1_error"], rtol=TOL, atol=TOL)
    assert G["l2h1_error"].min() > 0


def test_line_element_qfunctions_21_31():
    """geom_21 / geom_31, f_apply_hcurl_21 / _31, f_apply_hcurlmass_21 / _31 (line elements: boundaries of plane problems, curves
    in space) against vectors produced by the reference headers (tests/golden/make_golden.py: fixtures_line)."""
    L = np.load(os.path.join(os.path.dirname(__file__), "golden", "qf1d_golden.npz"))
    c1, _ = _ctx(L["ctx1"], 1)
    for sdim in (2, 3):
        cm, _ = _ctx(L["ctx%d" % sdim], sdim)
        geom = po.build_geom_factor_line(np.ones(1), L["qw"], L["J%d1" % sdim].T[None])
        np.testing.assert_allclose(geom[0, 1:], L["geom%d1" % sdim][1:], rtol=1e-13, atol=1e-13)
        g = L["geom%d1" % sdim][None]
        np.testing.assert_allclose(po.apply_hcurl_line(cm, g, L["u"][None])[0], L["hcurl_%d1" % sdim], rtol=1e-13, atol=1e-13)
        mv, gv = po.apply_hcurlmass_line(c1, cm, g, L["u"][None], L["gu"][None])
        np.testing.assert_allclose(mv[0], L["hcurlmass_%d1_v" % sdim], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(gv[0], L["hcurlmass_%d1_gv" % sdim], rtol=1e-13, atol=1e-13)


def test_cavity2d_eigenfrequencies():
    """Order-2 Nedelec triangles on the reference's own mesh, eps_r = 2.08 with loss tangent 4e-4, PEC:
    K x = omega^2 eps M x  ->  f = sqrt(lambda / (eps_r (1 - i tan d))) c0 / 2 pi."""
    import scipy.sparse.linalg as spl

    from palace_amd.fem import tri

    M_ = np.load(os.path.join(os.path.dirname(__file__), "golden", "cavity2d_mesh.npz"))
    en = M_["elem_nodes"].astype(np.int64)
    used, inv = np.unique(en[:, :3], return_inverse=True)
    mesh = tri.TriMesh(M_["nodes"][used], inv.reshape(-1, 3), M_["attr"], elem_nodes=en, nodes=M_["nodes"])
    nd = tri.NDTriSpace(mesh, 2)
    pts, wts = tri.tri_quadrature(3)
    interp, curl = nd.elem.tables(pts)
    J = mesh.jacobians(pts)
    geom = po.build_geom_factor_22(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 4))
    K = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, interp, curl, geom, po.QF_L2_1, po.CoeffCtx(dim=1),
                              qw=wts).assemble_sparse()
    M = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, interp, curl, geom, po.QF_HCURL_22,
                              po.CoeffCtx(dim=2)).assemble_sparse()
    free = np.setdiff1d(np.arange(nd.ndofs), nd.ess_dofs())
    # shift-invert next to each expected eigenvalue (the huge gradient null space sits at zero)
    Kf, Mf = K[free][:, free].tocsc(), M[free][:, free].tocsc()
    lam_ref = (2 * np.pi * M_["eig_re_GHz"] * 1e9 / 299792458.0) ** 2 * 2.08
    found = []
    for target in np.unique(np.round(lam_ref, 1)):
        vals = spl.eigsh(Kf, k=3, M=Mf, sigma=1.001 * target, which="LM", return_eigenvectors=False)
        found += [v for v in vals if abs(v - target) < 0.02 * target]
    lam = np.sort(np.array(found))
    assert lam.size == 5
    f = np.sqrt(lam / (2.08 * (1.0 - 4e-4j))) * 299792458.0 / (2 * np.pi) / 1e9
    np.testing.assert_allclose(f.real, M_["eig_re_GHz"], rtol=1e-7)
    np.testing.assert_allclose(f.imag, M_["eig_im_GHz"], rtol=1e-5)


def test_h1_laplace_eigenvalues_on_the_cavity2d_mesh():
    """2-D H1 path of the oracle (f_apply_hcurl_22 on the gradient = DiffusionIntegrator in 2-D, f_apply_h1_1 mass) on
    the reference's cavity2d mesh (the 1 x 0.5 rectangle) with order-3 nodal triangles: Dirichlet Laplace eigenvalues
    pi^2 (m^2 + 4 n^2)."""
    import scipy.sparse.linalg as spl

    from palace_amd.fem import tri

    M_ = np.load(os.path.join(os.path.dirname(__file__), "golden", "cavity2d_mesh.npz"))
    en = M_["elem_nodes"].astype(np.int64)
    used, inv = np.unique(en[:, :3], return_inverse=True)
    mesh = tri.TriMesh(M_["nodes"][used], inv.reshape(-1, 3), M_["attr"], elem_nodes=en, nodes=M_["nodes"])
    h1 = tri.H1TriSpace(mesh, 3)
    assert h1.ndofs == 19288
    pts, wts = tri.tri_quadrature(4)
    interp, grad = h1.elem.tables(pts)
    assert np.abs(interp.sum(axis=2) - 1.0).max() < 1e-13  # partition of unity
    J = mesh.jacobians(pts)
    geom = po.build_geom_factor_22(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 4))
    assert abs(geom[:, 1, :].sum() - 0.5) < 1e-12  # area of the rectangle
    K = po.CeedOperatorOracle(h1.ndofs, h1.offsets, None, interp, grad, geom, po.QF_HCURL_22, po.CoeffCtx(dim=2),
                              vector_fe=False).assemble_sparse()
    M = po.CeedOperatorOracle(h1.ndofs, h1.offsets, None, interp, grad, geom, po.QF_H1MASS, po.CoeffCtx(dim=1),
                              vector_fe=False).assemble_sparse()
    assert abs(K - K.T).max() < 1e-12 and abs(K @ np.ones(h1.ndofs)).max() < 1e-10
    free = np.setdiff1d(np.arange(h1.ndofs), h1.ess_dofs())
    lam = spl.eigsh(K[free][:, free].tocsc(), k=5, M=M[free][:, free].tocsc(), sigma=0.0, which="LM",
                    return_eigenvectors=False)
    ref = np.sort([np.pi ** 2 * (m * m + 4 * n * n) for m in range(1, 6) for n in range(1, 4)])[:5]
    np.testing.assert_allclose(np.sort(lam), ref, rtol=2e-8)


def test_cavity2d_magnetostatic_inductance():
    """The reference's regression value M11 = 6.283185306350e-07 H (test/data/regression/ref/cavity2d/magnetostatic/
    terminal-M.csv; examples/cavity2d/cavity2d_magnetostatic.json: order 2, PEC on the walls, unit surface current along
    +x on the bottom edge) through the oracle's 2-D curl-curl path (f_apply_l2_1 with its q_w input).  The field is the
    uniform B_z = mu0 J_s, exactly representable, so M = 2 E / I^2 = mu0 * area / width^2 = mu0 / 2 to rounding."""
    import scipy.sparse.linalg as spl

    from palace_amd.fem import tri

    M_ = np.load(os.path.join(os.path.dirname(__file__), "golden", "cavity2d_mesh.npz"))
    en = M_["elem_nodes"].astype(np.int64)
    used, inv = np.unique(en[:, :3], return_inverse=True)
    mesh = tri.TriMesh(M_["nodes"][used], inv.reshape(-1, 3), M_["attr"], elem_nodes=en, nodes=M_["nodes"])
    p = 2
    nd = tri.NDTriSpace(mesh, p)
    pts, wts = tri.tri_quadrature(3)
    interp, curl = nd.elem.tables(pts)
    J = mesh.jacobians(pts)
    geom = po.build_geom_factor_22(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 4))
    K = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, interp, curl, geom, po.QF_L2_1, po.CoeffCtx(dim=1),
                              qw=wts).assemble_sparse().tocsr()
    Mm = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, interp, curl, geom, po.QF_HCURL_22,
                               po.CoeffCtx(dim=2)).assemble_sparse().tocsr()
    # boundary edges -> (element, local edge)
    bv = np.searchsorted(used, M_["bdr_edges"].astype(np.int64))
    ekey = {tuple(e): i for i, e in enumerate(map(tuple, mesh.edge_verts))}
    owner = {}
    for e in range(mesh.ne):
        for k in range(3):
            owner.setdefault(int(mesh.elem_edges[e, k]), (e, k))
    s, ws = np.polynomial.legendre.leggauss(4)
    s, ws = 0.5 * (s + 1), 0.5 * ws
    sgn = np.where(nd.orients, -1.0, 1.0)
    b = np.zeros(nd.ndofs)
    pec_mask = np.zeros(mesh.edge_verts.shape[0], dtype=bool)
    width = 0.0
    for (va, vb), attr in zip(bv, M_["bdr_attr"]):
        ge = ekey[(min(va, vb), max(va, vb))]
        if attr == 3:
            pec_mask[ge] = True
            continue
        e, k = owner[ge]
        la, lb = tri.LOCAL_EDGES[k]
        that = tri.REF_VERTS[lb] - tri.REF_VERTS[la]
        xs = tri.REF_VERTS[la][None, :] + s[:, None] * that[None, :]
        val, _ = nd.elem.tables(xs)                       # [2, nq, P]
        loc = np.einsum("dqj,d,q->j", val, that, ws)      # int phi_hat . t_hat ds_hat  (= int phi . t ds, Piola)
        tphys = mesh.verts[mesh.tris[e, lb]] - mesh.verts[mesh.tris[e, la]]
        assert abs(tphys[1]) < 1e-12
        width += abs(tphys[0])
        np.add.at(b, nd.offsets[e], np.sign(tphys[0]) * sgn[e] * loc)   # J_s = +x
    assert width == pytest.approx(1.0, abs=1e-12)
    free = np.setdiff1d(np.arange(nd.ndofs), nd.ess_dofs(pec_mask))
    Kf, Mf, bf = K[free][:, free], Mm[free][:, free], b[free]
    # K is singular on gradients; b is orthogonal to them (closed current path through the PEC walls is not needed: the
    # gradients vanish on the walls and J_s is constant, so int J_s . grad(phi) ds = phi(1,0) - phi(0,0) = 0)
    x = spl.spsolve((Kf + 1e-9 * Mf).tocsc(), bf)
    r = Kf @ x - bf
    assert np.linalg.norm(r) < 1e-6 * np.linalg.norm(bf)
    energy = x @ (Kf @ x)                                  # = b^T K^+ b = area / width^2
    assert energy == pytest.approx(0.5, rel=1e-7)
    mu0 = 1.25663706127e-6                                 # utils/constants.hpp:26
    assert mu0 * energy == pytest.approx(float(M_["M11_H"]), rel=1e-7)


def test_tri_transfer_and_gradient_commute_with_the_curl():
    """Host tables of the 2-D p-multigrid on triangles (tri.nd_tri_transfer_matrix, tri.lowest_order_gradient) against the
    oracle's curl-curl: the order-1 gradients, prolonged to order 2, stay in the null space of the order-2 curl-curl
    (curl P G = 0), and the Galerkin product P^T K_2 P is the order-1 curl-curl (nested spaces, same quadrature)."""
    import scipy.sparse as sp

    from palace_amd.fem import tri

    M_ = np.load(os.path.join(os.path.dirname(__file__), "golden", "cavity2d_mesh.npz"))
    en = M_["elem_nodes"].astype(np.int64)[:400]
    used, inv = np.unique(en[:, :3], return_inverse=True)
    allused, allinv = np.unique(en, return_inverse=True)
    mesh = tri.TriMesh(M_["nodes"][used], inv.reshape(-1, 3), None, elem_nodes=allinv.reshape(en.shape), nodes=M_["nodes"][allused])
    s1, s2, h1 = tri.NDTriSpace(mesh, 1), tri.NDTriSpace(mesh, 2), tri.H1TriSpace(mesh, 1)
    pts, wts = tri.tri_quadrature(3)
    J = mesh.jacobians(pts)
    geom = po.build_geom_factor_22(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 4))
    Ks = []
    for s in (s1, s2):
        interp, curl = s.elem.tables(pts)
        Ks.append(po.CeedOperatorOracle(s.ndofs, s.offsets, s.orients, interp, curl, geom, po.QF_L2_1, po.CoeffCtx(dim=1),
                                        qw=wts).assemble_sparse().tocsr())
    T = tri.nd_tri_transfer_matrix(1, 2)
    sg1, sg2 = np.where(s1.orients, -1.0, 1.0), np.where(s2.orients, -1.0, 1.0)
    rows = np.repeat(s2.offsets[:, :, None], 3, axis=2).ravel()
    cols = np.repeat(s1.offsets[:, None, :], 8, axis=1).ravel()
    vals = (sg2[:, :, None] * T[None] * sg1[:, None, :]).ravel()
    _, first = np.unique(rows.astype(np.int64) * s1.ndofs + cols, return_index=True)  # every element gives the same entry
    P = sp.csr_matrix((vals[first], (rows[first], cols[first])), shape=(s2.ndofs, s1.ndofs))
    G = tri.lowest_order_gradient(h1, s1)
    scale = abs(Ks[1]).max()
    assert abs(Ks[1] @ (P @ G)).max() < 1e-11 * scale
    assert abs(P.T @ Ks[1] @ P - Ks[0]).max() < 1e-11 * scale
    xy = tri.vertex_coordinates(h1)
    ev = mesh.verts[mesh.edge_verts[:, 1]] - mesh.verts[mesh.edge_verts[:, 0]]
    assert np.abs(np.abs(G @ xy) - np.abs(ev)).max() < 1e-13  # gradient of the coordinates = the edge vectors
