"""CPU pins of the tetrahedral spaces (palace_amd.fem.tet) through the oracle operator: the discrete
sequence property (gradients of degree-p polynomials are in the order-p Nedelec space and in the kernel
of curl-curl, their mass norm is exact), symmetry, and the analytic PEC cube-cavity eigenvalues
k^2 = pi^2 (l^2 + m^2 + n^2).  These pin the tables and the curl-oriented (tridiagonal) restriction
the GPU parity tests then take as inputs."""
import numpy as np
import pytest
import scipy.linalg as sl

from oracle import palace_oracle as po
from palace_amd.fem import tet


def _geom(mesh, pts, wts):
    J = mesh.jacobians(pts)
    return po.build_geom_factor_33(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9))


def _field(p):
    def F(X):
        x, y, z = X[..., 0], X[..., 1], X[..., 2]
        if p == 1:
            return np.stack([1 + 0 * x, 2 + 0 * y, -1 + 0 * z], -1)
        if p == 2:
            return np.stack([y + 2 * x, x - z, -y + 3 * z], -1)
        return np.stack([2 * x * y + z * z, x * x - 2 * y * z, 2 * x * z - y * y], -1)
    return F


@pytest.mark.parametrize("p", [1, 2, 3])
def test_nd_tet_sequence_and_eigenvalues(p):
    mesh = tet.cube_tet_mesh(2)
    nd = tet.NDTetSpace(mesh, p)
    assert nd.diagonal_transform == (p == 1)
    pts, wts = tet.tet_quadrature(p + 1)
    interp, curl = nd.elem.tables(pts)
    geom = _geom(mesh, pts, wts)
    kw = dict(curl_orients=nd.curl_orients)
    K = po.CeedOperatorOracle(nd.ndofs, nd.offsets, None, interp, curl, geom, po.QF_HDIV, po.CoeffCtx(), **kw).assemble_sparse()
    M = po.CeedOperatorOracle(nd.ndofs, nd.offsets, None, interp, curl, geom, po.QF_HCURL, po.CoeffCtx(), **kw).assemble_sparse()
    assert abs(K - K.T).max() < 1e-13 and abs(M - M.T).max() < 1e-15
    F = _field(p)
    x = nd.interpolate(F)
    assert np.abs(K @ x).max() < 1e-12
    g, w = np.polynomial.legendre.leggauss(6)
    g, w = (g + 1) / 2, w / 2
    XX = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    WW = np.einsum("i,j,k->ijk", w, w, w).ravel()
    exact = (WW * (F(XX) ** 2).sum(-1)).sum()
    assert abs(x @ (M @ x) - exact) < 1e-12 * exact
    ess = nd.ess_dofs()
    free = np.setdiff1d(np.arange(nd.ndofs), ess)
    ev = sl.eigh(K[free][:, free].toarray(), M[free][:, free].toarray(), eigvals_only=True)
    ev = ev[ev > 1e-6][:5] / np.pi**2
    tol = {1: 0.15, 2: 0.03, 3: 3e-3}[p]
    assert np.abs(ev - np.array([2, 2, 2, 3, 3])).max() < tol * 3


def test_h1_tet_partition_of_unity_and_laplace():
    mesh = tet.to_quadratic(tet.cube_tet_mesh(2), lambda X: X + 0.03 * np.sin(3 * X[:, [1, 2, 0]]))
    for p in (1, 2, 3, 4):
        h1 = tet.H1TetSpace(mesh, p)
        pts, wts = tet.tet_quadrature(p + 1)
        interp, grad = h1.elem.tables(pts)
        assert np.abs(interp.sum(axis=2) - 1).max() < 1e-11 and np.abs(grad.sum(axis=2)).max() < 1e-10
        geom = _geom(mesh, pts, wts)
        A = po.CeedOperatorOracle(h1.ndofs, h1.offsets, None, interp, grad, geom, po.QF_HCURL, po.CoeffCtx(),
                                  vector_fe=False).assemble_sparse()
        assert abs(A - A.T).max() < 1e-12
        assert np.abs(A @ np.ones(h1.ndofs)).max() < 1e-11


def test_tet_quadrature_exactness():
    from math import factorial

    for rule, deg, npts in ((lambda: tet.tet_quadrature_symmetric(2), 2, 4), (lambda: tet.tet_quadrature_symmetric(4), 5, 14),
                            (lambda: tet.tet_quadrature_symmetric(6), 6, 24), (lambda: tet.tet_quadrature(4), 7, 64)):
        x, w = rule()
        assert len(w) == npts and w.min() > 0 and x.min() > 0 and x.sum(axis=1).max() < 1
        for d in range(deg + 1):
            for a in range(d + 1):
                for b in range(d + 1 - a):
                    c = d - a - b
                    exact = factorial(a) * factorial(b) * factorial(c) / factorial(a + b + c + 3)
                    assert abs((w * x[:, 0] ** a * x[:, 1] ** b * x[:, 2] ** c).sum() - exact) < 2e-15


@pytest.mark.parametrize("pc,pf", [(1, 2), (2, 3), (1, 3)])
def test_tet_prolongation_and_gradient_oracle(pc, pf):
    """CPU: the dense interpolator oracle with the dual-inverse range restriction reproduces fields of the
    coarse space exactly, and the discrete gradient commutes with nodal interpolation."""
    mesh = tet.to_quadratic(tet.cube_tet_mesh(2), lambda X: X + 0.03 * np.sin(3 * X[:, [1, 2, 0]]))
    ndc, ndf = tet.NDTetSpace(mesh, pc), tet.NDTetSpace(mesh, pf)
    P = po.DenseInterpOracle(ndc.restriction(), ndf.restriction(interp_range=True), tet.nd_tet_transfer_matrix(pc, pf))
    F = _field(pc)   # gradient of a degree-pc polynomial: in ND(pc) on straight elements ...
    meshs = tet.cube_tet_mesh(2)
    ndc_s, ndf_s = tet.NDTetSpace(meshs, pc), tet.NDTetSpace(meshs, pf)
    Ps = po.DenseInterpOracle(ndc_s.restriction(), ndf_s.restriction(interp_range=True), tet.nd_tet_transfer_matrix(pc, pf))
    xc, xf = ndc_s.interpolate(F), ndf_s.interpolate(F)
    assert np.abs(Ps.mult(xc) - xf).max() < 1e-12 * np.abs(xf).max()
    # ... on the curved mesh: the prolongation of any coarse vector is the same FUNCTION, so the fine
    # mass form of P x equals the coarse mass form of x
    pts, wts = tet.tet_quadrature(pf + 1)
    geom = _geom(mesh, pts, wts)

    def mass(nd):
        interp, curl = nd.elem.tables(pts)
        kw = dict(curl_orients=nd.curl_orients)
        return po.CeedOperatorOracle(nd.ndofs, nd.offsets, None, interp, curl, geom, po.QF_HCURL, po.CoeffCtx(), **kw)

    x = np.random.default_rng(0).uniform(-1, 1, ndc.ndofs)
    Mc, Mf = mass(ndc), mass(ndf)
    y = P.mult(x)
    a = x @ Mc.apply_add(x, np.zeros(ndc.ndofs))
    b = y @ Mf.apply_add(y, np.zeros(ndf.ndofs))
    assert abs(a - b) < 1e-11 * abs(a)
    # transpose consistency
    z = np.random.default_rng(1).uniform(-1, 1, ndf.ndofs)
    assert abs(z @ P.mult(x) - P.mult_transpose(z) @ x) < 1e-12 * np.abs(z).sum()
    # gradient: G (nodal values of phi) = ND interpolant of grad phi, for a polynomial of degree pf
    h1 = tet.H1TetSpace(meshs, pf)
    G = po.DenseInterpOracle(h1.restriction(), ndf_s.restriction(interp_range=True), tet.tet_gradient_matrix(pf))
    nodes = np.zeros((h1.ndofs, 3))
    Xn = np.einsum("qn,eni->eqi", np.stack([1 - h1.elem.nodes.sum(axis=1), *h1.elem.nodes.T], axis=1), meshs.verts[meshs.tets])
    nodes[h1.offsets.ravel()] = Xn.reshape(-1, 3)
    x_, y_, z_ = nodes.T
    phi = {1: x_ + 2 * y_ - z_, 2: x_ * y_ + x_**2 - y_ * z_ + 1.5 * z_**2, 3: x_**2 * y_ + x_ * z_**2 - y_**2 * z_}[pf]
    g = G.mult(phi)
    ref = ndf_s.interpolate(_field(pf))
    assert np.abs(g - ref).max() < 1e-11 * np.abs(ref).max()


def test_uniform_refinement_and_lowest_order_gradient():
    """tet.refine_uniform (every tetrahedron into 8, every boundary triangle into 4: volume, attributes and boundary tags kept,
    conforming) and tet.lowest_order_gradient / vertex_coordinates (the inputs of the native AMS solver on tetrahedra): the
    incidence matrix is in the kernel of the oracle's curl-curl operator, and G applied to a coordinate gives the edge vectors --
    the Nedelec interpolant of a constant field, whose mass energy is the volume."""
    from oracle import palace_oracle as po
    from palace_amd.fem import tet

    m = tet.cube_tet_mesh(2)
    m.attr[:] = 1 + (np.arange(m.ne) % 2)
    bf = m.face_verts[m.boundary_face_mask]
    m.bdr_tris, m.bdr_attr = bf, 1 + (np.arange(bf.shape[0]) % 3)
    r = tet.refine_uniform(m)
    assert r.ne == 8 * m.ne and len(r.bdr_tris) == 4 * len(m.bdr_tris)
    assert r.boundary_face_mask.sum() == 4 * m.boundary_face_mask.sum()  # conforming: no new boundary faces inside
    vol = lambda msh, a: np.einsum("ei,ei->e", np.cross(*(msh.verts[msh.tets[msh.attr == a]][:, k] - msh.verts[msh.tets[msh.attr == a]][:, 0]
                                                         for k in (1, 2))), msh.verts[msh.tets[msh.attr == a]][:, 3] - msh.verts[msh.tets[msh.attr == a]][:, 0]).sum() / 6  # noqa: E731
    for a in (1, 2):
        assert abs(vol(r, a) - vol(m, a)) < 1e-13
    assert np.array_equal(np.bincount(r.bdr_attr), 4 * np.bincount(m.bdr_attr))
    # every refined boundary triangle is a boundary face of the refined mesh
    key = {tuple(f) for f in map(tuple, r.face_verts[r.boundary_face_mask])}
    assert all(tuple(sorted(t)) in key for t in np.asarray(r.bdr_tris))

    nd, h1 = tet.NDTetSpace(r, 1), tet.H1TetSpace(r, 1)
    G = tet.lowest_order_gradient(h1, nd)
    xyz = tet.vertex_coordinates(h1)
    pts, wts = tet.default_tet_rule(1)
    interp, curl = nd.elem.tables(pts)
    J = r.jacobians(pts)
    og = po.build_geom_factor_33(np.ones(r.ne), wts, np.transpose(J, (0, 1, 3, 2)).reshape(r.ne, -1, 9))
    K = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, interp, curl, og, po.QF_HDIV, po.CoeffCtx()).assemble_sparse()
    M = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, interp, curl, og, po.QF_HCURL, po.CoeffCtx()).assemble_sparse()
    assert abs(K @ G).max() < 1e-12 * abs(K).max()
    for c in range(3):
        u = G @ xyz[:, c]
        assert abs(u @ (M @ u) - 1.0) < 1e-12  # |e_c|^2 over the unit cube


@pytest.mark.parametrize("p", [1, 2, 3])
def test_h1_boundary_block_numbering(p):
    """H1TetBoundaryBlock (round 5: the surface blocks of the auxiliary H1 operators, spaceoperator.cpp AddAuxIntegrators): the
    triangle element's nodes, taken in the ascending-vertex frame of every boundary face, carry the tetrahedral space's dof values
    (vertices, edge nodes counted from the smaller vertex, face nodes in the sorted-vertex lattice)."""
    from palace_amd.fem import tet

    mesh = tet.cube_tet_mesh(2)
    h1 = tet.H1TetSpace(mesh, p)
    V = mesh.verts[mesh.tets]
    nodes = h1.elem.nodes
    lam = np.stack([1 - nodes.sum(1), *nodes.T], 1)
    X = np.einsum("pn,eni->epi", lam, V)
    f = lambda X: (1.0 + 0.3 * X[..., 0] - 0.7 * X[..., 1] + 0.2 * X[..., 2]) ** p  # noqa: E731
    x = np.zeros(h1.ndofs)
    x[h1.offsets.ravel()] = f(X).ravel()
    faces = np.nonzero(mesh.boundary_face_mask)[0]
    blk = tet.H1TetBoundaryBlock(h1, faces)
    tn = blk.elem.nodes
    lt = np.stack([1 - tn.sum(1), tn[:, 0], tn[:, 1]], 1)
    Xt = np.einsum("pn,eni->epi", lt, mesh.verts[blk.elem_nodes])
    assert blk.P == (p + 1) * (p + 2) // 2
    assert np.abs(x[blk.offsets] - f(Xt)).max() < 1e-12
    # the surface diffusion form of a linear function: sum over the faces of area * |tangential gradient|^2
    if p >= 1:
        from palace_amd.fem import tri

        pts, wts = tri.tri_quadrature(p + 1)
        _, grad = blk.elem.tables(pts)                      # [2, Q, P]
        J = blk.jacobians(pts)                              # [ne, Q, 3, 2]
        g = np.array([0.3, -0.7, 0.2])
        xl = np.zeros(h1.ndofs)
        xl[h1.offsets.ravel()] = (1.0 + X @ g).ravel()
        total, want = 0.0, 0.0
        for e in range(blk.ne):
            for q in range(len(wts)):
                Jq = J[e, q]
                G = Jq.T @ Jq
                dref = grad[:, q, :] @ xl[blk.offsets[e]]   # reference gradient
                gs = Jq @ np.linalg.solve(G, dref)          # surface gradient in 3-D
                total += wts[q] * np.sqrt(np.linalg.det(G)) * (gs @ gs)
            n = np.cross(J[e, 0][:, 0], J[e, 0][:, 1])
            area = 0.5 * np.linalg.norm(n)
            nh = n / np.linalg.norm(n)
            gt = g - (g @ nh) * nh
            want += area * (gt @ gt)
        assert abs(total - want) < 1e-12 * want
