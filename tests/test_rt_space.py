"""Raviart-Thomas tetrahedra and the discrete curl (palace_amd/fem/rt.py) pinned on their defining properties (CPU):
unisolvence, the commuting curl, div curl = 0, normal continuity of the global space, and the energy identity
(K u, u) = (M_RT C u, C u) through the oracle's operators (the flux B = curl A of drivers/eigensolver.cpp:469-477
carries exactly the magnetic energy the curl-curl form measures)."""
import numpy as np
import pytest

from oracle import palace_oracle as po
from palace_amd.fem import rt, tet


def _warp(X):
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    return np.stack([x + 0.04 * np.sin(2 * y + z), y + 0.05 * x * z, z - 0.03 * np.cos(3 * x) * y], axis=1)


def _geom(mesh, pts, wts):
    J = mesh.jacobians(pts)
    Jcm = np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9)
    return po.build_geom_factor_33(mesh.attr.astype(np.float64), wts, Jcm)


@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_rt_element_commuting_curl(p):
    e, nd = rt.RTTetElement(p), tet.NDTetElement(p)
    assert e.P == p * (p + 1) * (p + 3) // 2
    C = rt.tet_curl_matrix(p)
    x = np.random.default_rng(0).uniform(0.05, 0.3, (9, 3))
    irt, div = e.tables(x)
    _, curl = nd.tables(x)
    u = np.random.default_rng(1).normal(size=nd.P)
    direct = np.einsum("dqj,j->qd", curl, u)
    through = np.einsum("dqi,i->qd", irt, C @ u)
    assert np.abs(direct - through).max() < 1e-10 * np.abs(direct).max()
    assert np.abs(div @ (C @ u)).max() < 1e-9 * np.abs(direct).max()
    # the dofs of the basis are the identity
    val, _ = e.tables(e.dof_pts)
    assert np.abs(np.einsum("dij,id->ij", val, e.dof_dirs) - np.eye(e.P)).max() < 1e-9


@pytest.mark.parametrize("p", [1, 2, 3])
def test_rt_space_normal_continuity(p):
    mesh = tet.to_quadratic(tet.cube_tet_mesh(2), _warp)
    sp = rt.RTTetSpace(mesh, p)
    x = np.random.default_rng(p).normal(size=sp.ndofs)
    sgn = np.where(sp.orients, -1.0, 1.0)
    ue = x[sp.offsets] * sgn  # [ne, P]
    # every interior face: flux density v . n dS from both sides at the same physical points
    owners = {}
    for e in range(mesh.ne):
        for k in range(4):
            owners.setdefault(int(mesh.elem_faces[e, k]), []).append((e, k))
    bary = np.array([[0.2, 0.3, 0.5], [0.6, 0.1, 0.3], [1 / 3, 1 / 3, 1 / 3]])
    checked = 0
    for f, own in owners.items():
        if len(own) != 2:
            continue
        flux = []
        for e, k in own:
            lf = tet.LOCAL_FACES[k]
            gv = mesh.tets[e, list(lf)]
            order = np.argsort(gv)  # local vertices in the sorted frame's order
            Vr = tet.REF_VERTS[[lf[i] for i in order]]
            pts = bary @ Vr  # the same physical points seen from both elements (P2 faces agree on shared nodes)
            val, _ = sp.elem.tables(pts)
            vhat = np.einsum("dqj,j->qd", val, ue[e])
            n_ref = np.cross(Vr[1] - Vr[0], Vr[2] - Vr[0])  # normal of the sorted frame, in this element's coordinates
            flux.append(vhat @ n_ref)  # Piola-invariant: v . n dS = vhat . nhat dShat
        assert np.abs(flux[0] - flux[1]).max() < 1e-10 * max(1.0, np.abs(flux[0]).max())
        checked += 1
    assert checked > 10


@pytest.mark.parametrize("kind", ["tet4", "tet10"])
@pytest.mark.parametrize("p", [1, 2, 3])
def test_discrete_curl_energy_identity(kind, p):
    mesh = tet.cube_tet_mesh(2)
    if kind == "tet10":
        mesh = tet.to_quadratic(mesh, _warp)
    nd, sp = tet.NDTetSpace(mesh, p), rt.RTTetSpace(mesh, p)
    pts, wts = tet.tet_quadrature(p + 1)
    ogeom = _geom(mesh, pts, wts)
    one = po.CoeffCtx(attr_mat=[0], mat_coeff=[np.eye(3)], dim=3)
    interp, curl = nd.elem.tables(pts)
    if nd.diagonal_transform:
        K = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, interp, curl, ogeom, po.QF_HDIV, one)
    else:
        K = po.CeedOperatorOracle(nd.ndofs, nd.offsets, None, interp, curl, ogeom, po.QF_HDIV, one,
                                  curl_orients=nd.curl_orients)
    rint, _ = sp.elem.tables(pts)
    M = po.CeedOperatorOracle(sp.ndofs, sp.offsets, sp.orients, rint, rint, ogeom, po.QF_HDIV, one)
    C = po.DenseInterpOracle(nd.restriction(), sp.restriction(interp_range=True), rt.tet_curl_matrix(p))
    u = np.random.default_rng(7).uniform(-1, 1, nd.ndofs)
    b = C.mult(u)
    e_k = u @ K.apply_add(u, np.zeros(nd.ndofs))
    e_m = b @ M.apply_add(b, np.zeros(sp.ndofs))
    assert abs(e_k - e_m) < 1e-11 * abs(e_k)
    # every element sharing a face computes the same flux dof (the interpolator averages identical values)
    ue = po.DenseInterpOracle._apply_rows(nd.restriction(), u[nd.offsets]) @ rt.tet_curl_matrix(p).T
    ge = b[sp.offsets] * np.where(sp.orients, -1.0, 1.0)
    assert np.abs(ue - ge).max() < 1e-10 * np.abs(ue).max()
    # adjoint
    v = np.random.default_rng(8).uniform(-1, 1, sp.ndofs)
    assert abs(v @ b - C.mult_transpose(v) @ u) < 1e-11 * abs(v @ b)
    if kind == "tet4" and p >= 2:
        # F = (-y, x, 0) lies in the space: the flux is the constant field (0, 0, 2)
        uF = nd.interpolate(lambda X: np.stack([-X[..., 1], X[..., 0], 0 * X[..., 0]], axis=-1))
        bF = C.mult(uF)
        J = mesh.jacobians(pts)
        det = np.linalg.det(J)
        vhat = np.einsum("dqj,ej->eqd", rint, bF[sp.offsets] * np.where(sp.orients, -1.0, 1.0))
        vphys = np.einsum("eqid,eqd->eqi", J, vhat) / det[..., None]
        assert np.abs(vphys - np.array([0.0, 0.0, 2.0])).max() < 1e-10


# ---- hexahedra ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("p", [1, 2, 3])
def test_rt_hex_discrete_curl_energy_identity(cylinder_mesh, p):
    """On the O-grid cylinder (faces seen in every relative orientation): the flux dofs computed by the elements sharing
    a face agree (tests the face permutation + sign of RTHexSpace against NDHexSpace's), and (K u, u) = (M_RT C u, C u)
    with the oracle's operators, whose Nedelec curl tables are evaluated independently (oracle/palace_oracle.py)."""
    from palace_amd.fem import rthex
    from palace_amd.fem.basis1d import gauss_legendre
    from palace_amd.fem.fespace import NDHexSpace
    from tests import util

    mesh = cylinder_mesh
    q1d = p + 1
    nd, rt_ = NDHexSpace(mesh, p), rthex.RTHexSpace(mesh, p)
    ogeom = util.oracle_geom(mesh, q1d)
    one = po.CoeffCtx()
    ident = np.arange(nd.P)
    interp, curl = po.nd_hex_dense_tables(p, q1d, ident)
    K = po.CeedOperatorOracle(nd.ndofs, nd.elem_dof_lex, nd.elem_sign_lex < 0, interp, curl, ogeom, po.QF_HDIV, one)
    rint, rdiv = rthex.rt_hex_tables(p, gauss_legendre(q1d)[0])
    M = po.CeedOperatorOracle(rt_.ndofs, rt_.elem_dof_lex, rt_.elem_sign_lex < 0, rint, rint, ogeom, po.QF_HDIV, one)
    Cm = rthex.hex_curl_matrix(p)
    dom = dict(offsets=nd.elem_dof_lex, lsize=nd.ndofs, orients=nd.elem_sign_lex < 0)
    C = po.DenseInterpOracle(dom, rt_.restriction(interp_range=True), Cm)
    u = np.random.default_rng(11).uniform(-1, 1, nd.ndofs)
    b = C.mult(u)
    ue = (u[nd.elem_dof_lex] * nd.elem_sign_lex) @ Cm.T
    ge = b[rt_.elem_dof_lex] * rt_.elem_sign_lex
    assert np.abs(ue - ge).max() < 1e-10 * np.abs(ue).max()
    e_k = u @ K.apply_add(u, np.zeros(nd.ndofs))
    e_m = b @ M.apply_add(b, np.zeros(rt_.ndofs))
    assert abs(e_k - e_m) < 1e-11 * abs(e_k)
    # div curl = 0 point-wise (reference divergence of the element flux)
    assert np.abs(ue @ rdiv.T).max() < 1e-9 * np.abs(ue).max()
    # every face dof is shared by at most two elements, interior dofs by one
    cnt = np.bincount(rt_.elem_dof_lex.ravel(), minlength=rt_.ndofs)
    assert cnt.min() == 1 and cnt.max() == 2 and rt_.ndofs == mesh.nfaces * p * p + mesh.ne * 3 * p * p * (p - 1)


@pytest.mark.parametrize("mesh_kind", ["tet10", "hex"])
@pytest.mark.parametrize("p", [1, 2])
def test_mixed_curl_operator_is_the_weak_form_of_the_discrete_curl(cylinder_mesh, mesh_kind, p):
    """MixedVectorCurlIntegrator (fem/integ/mixedveccurl.cpp:22-68, the right-hand side of the flux error estimator's
    projection): since curl ND_p lies in RT_p, (Q curl a, v) = (Q C a, v), i.e. B_mixed a = M_RT(Q) (C a); and the weak
    curl (:70-117, coefficient scaled by -1) is minus its transpose."""
    from palace_amd.fem import rthex
    from palace_amd.fem.basis1d import gauss_legendre
    from palace_amd.fem.fespace import NDHexSpace
    from tests import util

    if mesh_kind == "tet10":
        mesh = tet.to_quadratic(tet.cube_tet_mesh(2), _warp)
        nd, sp = tet.NDTetSpace(mesh, p), rt.RTTetSpace(mesh, p)
        pts, wts = tet.tet_quadrature(p + 1)
        ogeom = _geom(mesh, pts, wts)
        _, curl = nd.elem.tables(pts)
        rint, _ = sp.elem.tables(pts)
        Cm = rt.tet_curl_matrix(p)
        dom, rng_r, rng_i = nd.restriction(), sp.restriction(), sp.restriction(interp_range=True)
    else:
        mesh = cylinder_mesh
        nd, sp = NDHexSpace(mesh, p), rthex.RTHexSpace(mesh, p)
        ogeom = util.oracle_geom(mesh, p + 1)
        _, curl = po.nd_hex_dense_tables(p, p + 1, np.arange(nd.P))
        curl = np.asarray(curl).reshape(3, -1, nd.P)
        rint, _ = rthex.rt_hex_tables(p, gauss_legendre(p + 1)[0])
        Cm = rthex.hex_curl_matrix(p)
        dom = dict(offsets=nd.elem_dof_lex, lsize=nd.ndofs, orients=nd.elem_sign_lex < 0)
        rng_r = rng_i = sp.restriction()
    rng = np.random.default_rng(3)
    A = rng.uniform(-1, 1, (3, 3))
    Qc = po.CoeffCtx(attr_mat=[0] * int(mesh.attr.max()), mat_coeff=[A @ A.T + 2 * np.eye(3)])
    B = po.MixedCurlOperatorOracle(dom, rng_r, curl, rint, ogeom, Qc)
    M = po.CeedOperatorOracle(sp.ndofs, rng_r["offsets"], rng_r["orients"], rint, rint, ogeom, po.QF_HDIV, Qc)
    C = po.DenseInterpOracle(dom, rng_i, Cm)
    a = rng.uniform(-1, 1, nd.ndofs)
    lhs = B.mult(a)
    rhs = M.apply_add(C.mult(a), np.zeros(sp.ndofs))
    assert np.abs(lhs - rhs).max() < 1e-11 * np.abs(rhs).max()
    v = rng.uniform(-1, 1, sp.ndofs)
    assert abs(v @ lhs + a @ B.mult(v, weak=True)) < 1e-11 * abs(v @ lhs)
