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
