"""h-refinement transfer (reference fem/fespace.cpp:246-251: mfem::TransferOperator between the spaces of two multigrid levels
on different meshes; the h-levels of fem/multigrid.hpp:103-112), oracle restatement and host tables, on the CPU.  MFEM is an
external dependency of the reference (not in its tree), so the restatement is pinned on the DEFINING properties of that
operator: the prolonged function is the same function (pointwise, through the covariant map between child and parent reference
coordinates), gradients are prolonged to gradients (P_ND G_coarse = G_fine P_H1), constants stay constants."""
import numpy as np
import pytest

from oracle import palace_oracle as po
from palace_amd.fem import htransfer
from palace_amd.fem.fespace import H1HexSpace, NDHexSpace, lowest_order_gradient
from palace_amd.fem.mesh import ogrid_cylinder, refine_uniform


def _hex_eval_nd(p, coef, x):
    """Reference-space value [3] of the ND tensor element function with lexicographic coefficients `coef` at x."""
    cp, op = po.gll_points(p + 1), po.gl_points(p)[0]
    out = np.zeros(3)
    n = p * (p + 1) ** 2
    for comp in range(3):
        nd = [p + 1] * 3
        nd[comp] = p
        nodes = [cp] * 3
        nodes[comp] = op
        for k in range(nd[2]):
            for j in range(nd[1]):
                for i in range(nd[0]):
                    out[comp] += (coef[comp * n + i + nd[0] * (j + nd[1] * k)] * po.lagrange(nodes[0], x[0], i)[0]
                                  * po.lagrange(nodes[1], x[1], j)[0] * po.lagrange(nodes[2], x[2], k)[0])
    return out


def _oracle_transfer(coarse, fine, p, hcurl):
    parent = np.arange(fine.mesh.ne) // 8
    ones_c = np.ones_like(coarse.elem_dof_lex, dtype=np.int8)
    return po.RefinementTransferOracle(coarse.elem_dof_lex[parent], (coarse.elem_sign_lex if hcurl else ones_c)[parent],
                                       fine.elem_dof_lex, fine.elem_sign_lex if hcurl else np.ones_like(fine.elem_dof_lex, dtype=np.int8),
                                       coarse.ndofs, fine.ndofs, po.hex_refinement_matrices(p, hcurl), np.arange(fine.mesh.ne) % 8)


@pytest.mark.parametrize("p", [1, 2])
def test_hex_prolonged_function_is_the_same_function(p):
    mc = ogrid_cylinder(1, 2)
    mf = refine_uniform(mc)
    c, f = NDHexSpace(mc, p), NDHexSpace(mf, p)
    P = _oracle_transfer(c, f, p, True)
    rng = np.random.default_rng(4)
    xc = rng.uniform(-1, 1, c.ndofs)
    yf = P.mult(xc)
    for e in rng.choice(mf.ne, 12, replace=False):
        E, k = e // 8, e % 8
        half = np.array([k & 1, (k >> 1) & 1, k >> 2], dtype=np.float64)
        cf = yf[f.elem_dof_lex[e]] * f.elem_sign_lex[e]
        cc = xc[c.elem_dof_lex[E]] * c.elem_sign_lex[E]
        for xh in rng.uniform(0, 1, (3, 3)):
            # covariant map between the child's and the parent's reference coordinates: u_child = A^T u_parent, A = I / 2
            assert np.abs(_hex_eval_nd(p, cf, xh) - 0.5 * _hex_eval_nd(p, cc, 0.5 * (half + xh))).max() < 1e-12
    # the transpose is the transpose
    yc = rng.uniform(-1, 1, f.ndofs)
    assert abs(yc @ yf - P.mult_transpose(yc) @ xc) < 1e-11 * np.linalg.norm(yc) * np.linalg.norm(yf)


def test_hex_gradients_are_prolonged_to_gradients_and_constants_to_constants():
    mc = ogrid_cylinder(1, 2)
    mf = refine_uniform(mc)
    ndc, ndf, h1c, h1f = NDHexSpace(mc, 1), NDHexSpace(mf, 1), H1HexSpace(mc, 1), H1HexSpace(mf, 1)
    Pn, Ph = _oracle_transfer(ndc, ndf, 1, True), _oracle_transfer(h1c, h1f, 1, False)
    Gc, Gf = lowest_order_gradient(h1c, ndc), lowest_order_gradient(h1f, ndf)
    assert np.abs(Ph.mult(np.ones(h1c.ndofs)) - 1.0).max() < 1e-14
    rng = np.random.default_rng(5)
    for _ in range(4):
        phi = rng.uniform(-1, 1, h1c.ndofs)
        assert np.abs(Pn.mult(Gc @ phi) - Gf @ Ph.mult(phi)).max() < 1e-13


def test_host_tables_match_the_restatement():
    for p in (1, 2, 3):
        for hcurl in (True, False):
            assert np.abs(htransfer.hex_child_matrices(p, hcurl) - po.hex_refinement_matrices(p, hcurl)).max() < 1e-14
    mc = ogrid_cylinder(1, 1)
    mf = refine_uniform(mc)
    c, f = NDHexSpace(mc, 1), NDHexSpace(mf, 1)
    dom, rng_, M, mid = htransfer.hex_refinement(c, f)
    P = _oracle_transfer(c, f, 1, True)
    assert np.array_equal(dom["offsets"], P.dc) and np.array_equal(rng_["offsets"], P.df) and np.array_equal(mid, P.mid)
    assert np.array_equal(np.where(dom["orients"], -1, 1), P.sc) and np.array_equal(np.where(rng_["orients"], -1, 1), P.sf)


def test_tet_prolonged_function_is_the_same_function():
    from palace_amd.fem import tet

    mc = tet.cube_tet_mesh(2)
    mf = tet.refine_uniform(mc)
    c, f = tet.NDTetSpace(mc, 1), tet.NDTetSpace(mf, 1)
    dom, rng_, Ms, mid = htransfer.tet_refinement(c, f)
    sg = lambda o: np.where(o, -1.0, 1.0)  # noqa: E731
    P = po.RefinementTransferOracle(dom["offsets"], sg(dom["orients"]), rng_["offsets"], sg(rng_["orients"]), c.ndofs, f.ndofs, Ms, mid)
    rng = np.random.default_rng(6)
    xc = rng.uniform(-1, 1, c.ndofs)
    yf = P.mult(xc)
    # physical-space comparison (straight-sided elements): u(x) = J^-T u_hat(x_hat) on both meshes at the same physical point
    for e in rng.choice(mf.ne, 16, replace=False):
        E = e // 8
        Xf, Xc = mf.verts[mf.tets[e]], mc.verts[mc.tets[E]]
        Jf, Jc = (Xf[1:] - Xf[0]).T, (Xc[1:] - Xc[0]).T
        xh = rng.dirichlet(np.ones(4), 3)[:, 1:]
        xphys = Xf[0] + xh @ Jf.T
        xhc = np.linalg.solve(Jc, (xphys - Xc[0]).T).T
        vf, _ = f.elem.tables(xh)      # [3, n, P]
        vc, _ = c.elem.tables(xhc)
        uf = np.einsum("dnj,j->nd", vf, yf[f.offsets[e]] * sg(f.orients[e])) @ np.linalg.inv(Jf)
        uc = np.einsum("dnj,j->nd", vc, xc[c.offsets[E]] * sg(c.orients[E])) @ np.linalg.inv(Jc)
        assert np.abs(uf - uc).max() < 1e-11 * max(1.0, np.abs(uc).max())
    h1c, h1f = tet.H1TetSpace(mc, 1), tet.H1TetSpace(mf, 1)
    dh, rh, Mh, mh = htransfer.tet_refinement(h1c, h1f)
    one = np.ones_like(dh["offsets"], dtype=np.float64)
    Ph = po.RefinementTransferOracle(dh["offsets"], one, rh["offsets"], one, h1c.ndofs, h1f.ndofs, Mh, mh)
    assert np.abs(Ph.mult(np.ones(h1c.ndofs)) - 1.0).max() < 1e-14
    Gc, Gf = tet.lowest_order_gradient(h1c, c), tet.lowest_order_gradient(h1f, f)
    phi = rng.uniform(-1, 1, h1c.ndofs)
    assert np.abs(P.mult(Gc @ phi) - Gf @ Ph.mult(phi)).max() < 1e-13
