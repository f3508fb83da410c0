"""The reference's spheres example (examples/spheres: two conducting spheres in a grounded far-field sphere, order-3 H1 on
14 362 cubic tetrahedra) through the oracle's 3-D H1 path: the Maxwell capacitance matrix of
test/data/regression/ref/spheres/terminal-C.csv (the reference's own gate is rtol 1e-4).
C_ij = eps0 L0 phi_i^T K phi_j with phi_i the potential for V_i = 1, all other conductors and the far field at 0.
With the symmetric 24-point rule (the order-2p rule the reference takes from MFEM) all four entries agree with the
regression file to 1.3e-10 relative: element, isoparametric cubic geometry, quadrature, D stage and Dirichlet handling are
the reference's, digit for digit."""
import os

import numpy as np
import pytest

from oracle import palace_oracle as po


def test_spheres_capacitance_matrix():
    import scipy.sparse.linalg as spl

    from palace_amd.fem import tet

    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "spheres_mesh.npz"))
    nodes, en = d["nodes"], d["elem_nodes"].astype(np.int64)
    used, inv = np.unique(en[:, :4], return_inverse=True)
    mesh = tet.TetMesh(nodes[used], inv.reshape(-1, 4), d["attr"])   # topology (vertices); geometry is cubic, below
    p = 3
    h1 = tet.H1TetSpace(mesh, p)
    pts, wts = tet.default_tet_rule(p)                                # symmetric 24-point rule, degree 6 = 2 p
    interp, grad = h1.elem.tables(pts)
    # isoparametric cubic geometry: the order-3 nodal basis on the fixture's node order (= h1_tet_nodes(3))
    G = tet.H1TetElement(3).tables(pts)[1]                            # [3, Q, 20]
    J = np.einsum("dqn,eni->eqid", G, nodes[en])
    assert np.linalg.det(J).min() > 0
    vol = float((np.linalg.det(J) * wts[None, :]).sum())
    geom = po.build_geom_factor_33(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9))
    K = po.CeedOperatorOracle(h1.ndofs, h1.offsets, None, interp, grad, geom, po.QF_HCURL, po.CoeffCtx(),
                              vector_fe=False).assemble_sparse().tocsr()
    assert abs(K @ np.ones(h1.ndofs)).max() < 1e-9 * abs(K).max()
    # boundary faces by attribute: 2 far field (ground), 3 sphere A, 4 sphere B
    bt = np.sort(np.searchsorted(used, d["bdr_tris"].astype(np.int64)), axis=1)
    fkey = {tuple(f): i for i, f in enumerate(map(tuple, mesh.face_verts))}
    masks = {}
    for a in (2, 3, 4):
        m = np.zeros(mesh.face_verts.shape[0], dtype=bool)
        m[[fkey[tuple(f)] for f in bt[d["bdr_attr"] == a]]] = True
        assert np.all(mesh.boundary_face_mask[m])
        masks[a] = h1.ess_dofs(m)
    ess = np.unique(np.concatenate(list(masks.values())))
    free = np.setdiff1d(np.arange(h1.ndofs), ess)
    Kff = K[free][:, free].tocsr()
    dinv = 1.0 / Kff.diagonal()
    jac = spl.LinearOperator(Kff.shape, matvec=lambda r: dinv * r)
    phi = []
    for a in (3, 4):
        v = np.zeros(h1.ndofs)
        v[masks[a]] = 1.0
        x, info = spl.cg(Kff, -(K[free][:, ess] @ v[ess]), rtol=1e-13, maxiter=5000, M=jac)
        assert info == 0
        v[free] = x
        phi.append(v)
    eps0 = 1.0 / (1.25663706127e-6 * 299792458.0 ** 2)               # utils/constants.hpp:21-30
    L0 = 1.0e-2                                                       # spheres.json "L0": mesh in cm
    C = np.array([[eps0 * L0 * (pi @ (K @ pj)) for pj in phi] for pi in phi])
    ref = d["C_F"]
    assert np.abs(C - ref).max() < 1e-9 * np.abs(ref).max(), (C, ref)
    # physical sanity: symmetric, diagonally dominant Maxwell matrix, negative mutual term
    assert abs(C[0, 1] - C[1, 0]) < 1e-12 * abs(C[0, 0]) and C[0, 1] < 0 < C[0, 0] < C[1, 1]
    assert vol > 0
