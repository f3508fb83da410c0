"""Tetrahedral Nedelec / H1 spaces through the dense MFMA path vs the oracle.

The spaces come from palace_amd.fem.tet (the role MFEM plays for the reference): order-p first-kind
Nedelec tets with the curl-oriented (tridiagonal int8) restriction of
fem/libceed/restriction.cpp:299-369 for p >= 2, nodal H1 tets, straight (tet4) and curved (tet10)
geometry.  Parity is GPU vs oracle on the same tables; the tables themselves are pinned by
tests/test_tet_space.py (exact sequence property, analytic cavity eigenvalues)."""
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

    G = mesh.geometry_grad_table(pts)
    g = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, G, wts)
    J = mesh.jacobians(pts)
    Jcm = np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9)
    return g, po.build_geom_factor_33(mesh.attr.astype(np.float64), wts, Jcm)


@pytest.mark.parametrize("kind", ["tet4", "tet10"])
def test_tet_geometry(kind):
    from palace_amd.fem import tet

    mesh = _mesh(kind)
    pts, wts = tet.tet_quadrature(3)
    g, ref = _geom(mesh, pts, wts)
    got = g.to_numpy()
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
    # volumes: sum of w detJ
    if kind == "tet4":
        assert abs(ref[:, 1, :].sum() - 1.0) < 1e-13


ND_MODES = [("curl", "QF_HDIV_33", po.QF_HDIV, "C"), ("vmass", "QF_HCURL_33", po.QF_HCURL, "I"),
            ("curlmass", "QF_HDIVMASS_33", po.QF_HDIVMASS, "CI")]


@pytest.mark.parametrize("mode", ND_MODES, ids=[m[0] for m in ND_MODES])
@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("kind", ["tet4", "tet10"])
def test_nd_tet_apply(kind, p, mode):
    import torch

    from palace_amd import ceed
    from palace_amd.fem import tet

    name, qf_name, qf_o, ops_s = mode
    mesh = _mesh(kind)
    nd = tet.NDTetSpace(mesh, p)
    pts, wts = tet.tet_quadrature(p + 1)
    interp, curl = nd.elem.tables(pts)
    geom, ogeom = _geom(mesh, pts, wts)
    c3, b3 = util.make_ctx("aniso", 2)
    cm, bm = util.make_ctx("scalar", 2)
    ctxs, blob = ((cm, c3), np.concatenate([bm, b3])) if name == "curlmass" else ((c3, None), b3)
    if nd.diagonal_transform:
        block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, orients=nd.orients)
        orc = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, interp, curl, ogeom, qf_o, *ctxs)
    else:
        block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, curl_orients=nd.curl_orients)
        orc = po.CeedOperatorOracle(nd.ndofs, nd.offsets, None, interp, curl, ogeom, qf_o, *ctxs,
                                    curl_orients=nd.curl_orients)
    ops = sum({"C": ceed.EVAL_CURL, "I": ceed.EVAL_INTERP}[c] for c in ops_s)
    op = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, getattr(ceed, qf_name), blob, ops).finalize()
    # straight-sided tets run in the affine form (D of one point per element and field), curved ones in the general form
    # (unless the tables of the two fields leave no LDS for twelve waves: p = 3 with this 64-point rule)
    if kind == "tet10":
        assert op.dense_affine() == 0
    elif not (p == 3 and name == "curlmass"):
        assert op.dense_affine() == 1
    rng = np.random.default_rng(p)
    x = rng.uniform(-1, 1, nd.ndofs)
    y_ref = orc.apply_add(x, np.zeros(nd.ndofs))
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    op.mult(xd, yd)
    err = np.abs(yd.cpu().numpy() - y_ref).max() / np.abs(y_ref).max()
    assert err < REL, f"Mult rel err {err:.3e}"
    if op.dense_affine():  # the general form on the same elements: same result up to the rounding of w_q / w_0
        import os

        os.environ["PALACE_AMD_DENSE_AFFINE"] = "0"
        try:
            op_g = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, getattr(ceed, qf_name), blob, ops).finalize()
        finally:
            del os.environ["PALACE_AMD_DENSE_AFFINE"]
        assert op_g.dense_affine() == 0
        yg = torch.empty_like(xd)
        op_g.mult(xd, yg)
        assert (yg - yd).abs().max().item() <= 1e-14 * yd.abs().max().item()
    d_ref = orc.diagonal()
    dd = torch.empty_like(xd)
    op.assemble_diagonal(dd)
    err = np.abs(dd.cpu().numpy() - d_ref).max() / np.abs(d_ref).max()
    assert err < REL, f"diagonal rel err {err:.3e}"


H1_MODES = [("diff", "QF_HCURL_33", po.QF_HCURL, "G"), ("mass", "QF_H1_1", po.QF_H1MASS, "I"),
            ("diffmass", "QF_HCURLMASS_33", po.QF_HCURLMASS, "GI")]


@pytest.mark.parametrize("mode", H1_MODES, ids=[m[0] for m in H1_MODES])
@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_h1_tet_apply(p, mode):
    import torch

    from palace_amd import ceed
    from palace_amd.fem import tet

    name, qf_name, qf_o, ops_s = mode
    mesh = _mesh("tet10")
    h1 = tet.H1TetSpace(mesh, p)
    pts, wts = tet.tet_quadrature(p + 1)
    interp, grad = h1.elem.tables(pts)
    geom, ogeom = _geom(mesh, pts, wts)
    c3, b3 = util.make_ctx("aniso", 2)
    c1 = po.CoeffCtx(attr_mat=[0, 0], mat_coeff=[np.array([1.7])], dim=1)
    if name == "diff":
        ctxs, blob = (c3, None), b3
    elif name == "mass":
        ctxs, blob = (c1, None), c1.pack()
    else:
        ctxs, blob = (c1, c3), np.concatenate([c1.pack(), b3])
    block = ceed.DenseBlock(ceed.FE_H1, h1.ndofs, h1.offsets, interp, grad)
    ops = sum({"G": ceed.EVAL_GRAD, "I": ceed.EVAL_INTERP}[c] for c in ops_s)
    op = ceed.Operator(h1.ndofs, h1.ndofs).add_dense_integrator(geom, block, getattr(ceed, qf_name), blob, ops).finalize()
    orc = po.CeedOperatorOracle(h1.ndofs, h1.offsets, None, interp, grad, ogeom, qf_o, *ctxs, vector_fe=False)
    rng = np.random.default_rng(p)
    x = rng.uniform(-1, 1, h1.ndofs)
    y_ref = orc.apply_add(x, np.zeros(h1.ndofs))
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    op.mult(xd, yd)
    err = np.abs(yd.cpu().numpy() - y_ref).max() / np.abs(y_ref).max()
    assert err < REL, f"Mult rel err {err:.3e}"
    if name == "diff":  # constants are in the kernel of the stiffness operator
        one = torch.ones_like(xd)
        op.mult(one, yd)
        assert float(yd.abs().max()) < 1e-11 * float(np.abs(y_ref).max())


@pytest.mark.parametrize("mode", ND_MODES, ids=[m[0] for m in ND_MODES])
@pytest.mark.parametrize("p", [1, 2, 3])
def test_nd_tet_apply_partly_curved_mesh(p, mode):
    """A tet10 mesh that is curved in one half only (what a Palace mesh with a curved boundary looks like): the blocks of
    straight elements run the affine form, the others the general one, in two launches on block lists."""
    import torch

    from palace_amd import ceed
    from palace_amd.fem import tet

    name, qf_name, qf_o, ops_s = mode
    base = tet.cube_tet_mesh(5)

    def half_warp(X):
        w = np.clip(X[:, 0] - 0.45, 0.0, None) ** 2  # identity for x <= 0.45
        return X + np.stack([0.3 * w * np.sin(3 * X[:, 1]), 0.4 * w * X[:, 2], -0.35 * w * np.cos(2 * X[:, 1])], axis=1)

    mesh = tet.to_quadratic(base, half_warp)
    mesh.attr[:] = 1 + (np.arange(mesh.ne) % 2)
    nd = tet.NDTetSpace(mesh, p)
    pts, wts = tet.tet_quadrature(p + 1) if not (p == 3 and name == "curlmass") else tet.default_tet_rule(p)
    interp, curl = nd.elem.tables(pts)
    geom, ogeom = _geom(mesh, pts, wts)
    c3, b3 = util.make_ctx("aniso", 2)
    cm, bm = util.make_ctx("scalar", 2)
    ctxs, blob = ((cm, c3), np.concatenate([bm, b3])) if name == "curlmass" else ((c3, None), b3)
    kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
    okw = dict(curl_orients=nd.curl_orients) if not nd.diagonal_transform else {}
    block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, **kw)
    orc = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients if nd.diagonal_transform else None, interp, curl, ogeom, qf_o,
                                *ctxs, **okw)
    ops = sum({"C": ceed.EVAL_CURL, "I": ceed.EVAL_INTERP}[c] for c in ops_s)
    op = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, getattr(ceed, qf_name), blob, ops).finalize()
    assert op.dense_affine() == 1  # (some blocks: the straight half)
    x = np.random.default_rng(p).uniform(-1, 1, nd.ndofs)
    y_ref = orc.apply_add(x, np.zeros(nd.ndofs))
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    op.mult(xd, yd)
    assert np.abs(yd.cpu().numpy() - y_ref).max() / np.abs(y_ref).max() < REL
    dd = torch.empty_like(xd)
    op.assemble_diagonal(dd)
    d_ref = orc.diagonal()
    assert np.abs(dd.cpu().numpy() - d_ref).max() / np.abs(d_ref).max() < REL


@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("kind", ["tet4", "tet10"])
def test_nd_tet_boundary_curlcurl_and_pair(kind, p):
    """The other ND boundary terms of SpaceOperator (models/spaceoperator.cpp:292-302): the surface curl-curl
    (CurlCurlIntegrator on boundary elements: f_apply_l2_1 on the scalar surface curl, with the q_w input) and curl-curl + mass
    in one QFunction (f_apply_hdivmass_32) on the boundary triangles of a tet mesh."""
    import torch

    from palace_amd import ceed
    from palace_amd.fem import tet, tri

    mesh = _mesh(kind)
    nd = tet.NDTetSpace(mesh, p)
    faces = np.nonzero(mesh.boundary_face_mask)[0]
    blk = tet.NDTetBoundaryBlock(nd, faces, 1 + (np.arange(faces.size) % 2))
    pts, wts = tri.tri_quadrature(p + 1)
    interp, curl = blk.elem.tables(pts)
    bgeom = ceed.DenseGeomFactorData(blk.elem_nodes, blk.nodes, blk.attr, blk.geometry_grad_table(pts), wts)
    J = blk.jacobians(pts)
    og = po.build_geom_factor_32(blk.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(blk.ne, -1, 6))
    c3, b3 = util.make_ctx("aniso", 2)
    c1 = po.CoeffCtx(attr_mat=[1, 0], mat_coeff=[np.array([1.9]), np.array([0.4])], dim=1)
    block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, blk.offsets, interp, curl, orients=blk.orients)
    x = np.random.default_rng(p).uniform(-1, 1, nd.ndofs)
    xd = torch.from_numpy(x).cuda()
    for qf, oqf, blob, ctxs, ops in (
            (ceed.QF_L2_1, po.QF_L2_1, c1.pack(), (c1, None), ceed.EVAL_CURL | ceed.EVAL_WEIGHT),
            (ceed.QF_HDIVMASS_32, po.QF_HDIVMASS_32, np.concatenate([b3, c1.pack()]), (c3, c1),
             ceed.EVAL_CURL | ceed.EVAL_INTERP | ceed.EVAL_WEIGHT)):
        op = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(bgeom, block, qf, blob, ops).finalize()
        orc = po.CeedOperatorOracle(nd.ndofs, blk.offsets, blk.orients, interp, curl, og, oqf, *ctxs, qw=wts)
        ref = orc.apply_add(x, np.zeros(nd.ndofs))
        y = torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")
        op.mult(xd, y)
        assert np.abs(y.cpu().numpy() - ref).max() < REL * np.abs(ref).max(), oqf
        d = torch.empty_like(y)
        op.assemble_diagonal(d)
        dref = orc.diagonal()
        assert np.abs(d.cpu().numpy() - dref).max() < REL * np.abs(dref).max(), oqf


@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("kind", ["tet4", "tet10"])
def test_nd_tet_boundary_mass(kind, p):
    """Surface (impedance / absorbing-boundary type) mass term on the boundary triangles of a tet mesh:
    f_apply_hcurl_32 on 2-D Nedelec triangles in 3-D space whose dofs are the tetrahedral space's face / edge dofs
    (SURVEY.md 8f-2), alone and added to the volume operator in one ceed::Operator."""
    import torch

    from palace_amd import ceed
    from palace_amd.fem import tet, tri

    mesh = _mesh(kind)
    nd = tet.NDTetSpace(mesh, p)
    faces = np.nonzero(mesh.boundary_face_mask)[0]
    battr = 1 + (np.arange(faces.size) % 2)
    blk = tet.NDTetBoundaryBlock(nd, faces, battr)
    pts, wts = tri.tri_quadrature(p + 1)
    interp, curl = blk.elem.tables(pts)
    bgeom = ceed.DenseGeomFactorData(blk.elem_nodes, blk.nodes, blk.attr, blk.geometry_grad_table(pts), wts)
    J = blk.jacobians(pts)
    og = po.build_geom_factor_32(blk.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(blk.ne, -1, 6))
    got = bgeom.to_numpy()
    assert got.shape == og.shape and np.abs(got - og).max() <= 1e-13 * np.abs(og).max()
    c3, b3 = util.make_ctx("aniso", 2)
    block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, blk.offsets, interp, None, orients=blk.orients)
    op = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(bgeom, block, ceed.QF_HCURL_32, b3, ceed.EVAL_INTERP).finalize()
    orc = po.CeedOperatorOracle(nd.ndofs, blk.offsets, blk.orients, interp, curl, og, po.QF_HCURL_32, c3)
    x = np.random.default_rng(p).uniform(-1, 1, nd.ndofs)
    ref = orc.apply_add(x, np.zeros(nd.ndofs))
    y = torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")
    op.mult(torch.from_numpy(x).cuda(), y)
    assert np.abs(y.cpu().numpy() - ref).max() < REL * np.abs(ref).max()
    # volume mass + boundary term as two sub-operators of ONE ceed::Operator (how Palace composes K, M and the
    # impedance term): the apply is the sum of the two
    vpts, vwts = tet.default_tet_rule(p)
    vint, vcurl = nd.elem.tables(vpts)
    vgeom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(vpts), vwts)
    kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
    vblock = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, vint, vcurl, **kw)
    vol = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(vgeom, vblock, ceed.QF_HCURL_33, b3, ceed.EVAL_INTERP).finalize()
    both = (ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(vgeom, vblock, ceed.QF_HCURL_33, b3, ceed.EVAL_INTERP)
            .add_dense_integrator(bgeom, block, ceed.QF_HCURL_32, b3, ceed.EVAL_INTERP).finalize())
    yv, yb = torch.empty_like(y), torch.empty_like(y)
    vol.mult(torch.from_numpy(x).cuda(), yv)
    both.mult(torch.from_numpy(x).cuda(), yb)
    assert float((yb - (yv + y)).abs().max()) < 1e-13 * float(yv.abs().max())
    if kind == "tet4":  # flat faces of the unit cube, identity material: sum over faces of area * |E_tangential|^2
        ident = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(
            bgeom, block, ceed.QF_HCURL_32, ceed.coefficient_context(3), ceed.EVAL_INTERP).finalize()
        xe = nd.interpolate(lambda X: np.broadcast_to(np.array([1.0, 2.0, 3.0]), X.shape))
        xd = torch.from_numpy(xe).cuda()
        ident.mult(xd, y)
        assert abs(float(xd @ y) - 56.0) < 1e-11 * 56.0


@pytest.mark.parametrize("group", ["0", "2", "4", "8"])  # (2: reachable through the switch only)
@pytest.mark.parametrize("layout", ["rows", "block"])
@pytest.mark.parametrize("p", [2, 3])
def test_nd_tet_gather_forms(monkeypatch, p, layout, group):
    """Round 6: the dense path's E-vector with the dofs of an element together (`rows`) or one row per dof (`block`), and its E^T
    gather with 1 / 2 / 4 / 8 lanes per dof: curl-curl + mass `Mult` and the diagonal against the oracle on curved, curl-oriented
    tetrahedra -- and the fused smoother step (the gather's epilogue) against the unfused one in every form."""
    import torch

    from palace_amd import ceed, linalg
    from palace_amd.fem import tet

    monkeypatch.setenv("PALACE_AMD_DENSE_ELAYOUT", layout)
    monkeypatch.setenv("PALACE_AMD_DENSE_GATHER_GROUP", group)
    mesh = _mesh("tet10")
    nd = tet.NDTetSpace(mesh, p)
    pts, wts = tet.tet_quadrature(p + 1)
    interp, curl = nd.elem.tables(pts)
    geom, ogeom = _geom(mesh, pts, wts)
    c3, b3 = util.make_ctx("aniso", 2)
    cm, bm = util.make_ctx("scalar", 2)
    block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, curl_orients=nd.curl_orients)
    orc = po.CeedOperatorOracle(nd.ndofs, nd.offsets, None, interp, curl, ogeom, po.QF_HDIVMASS, cm, c3, curl_orients=nd.curl_orients)
    op = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, ceed.QF_HDIVMASS_33, np.concatenate([bm, b3]),
                                                                 ceed.EVAL_CURL + ceed.EVAL_INTERP).finalize()
    rows, lanes = op.dense_gather_form()
    assert rows == (layout == "rows") and lanes == (1 if group == "0" else int(group))
    x = np.random.default_rng(p).uniform(-1, 1, nd.ndofs)
    y_ref = orc.apply_add(x, np.zeros(nd.ndofs))
    xd = torch.from_numpy(x).cuda()
    yd = op.mult(xd, torch.full_like(xd, np.nan))
    assert np.abs(yd.cpu().numpy() - y_ref).max() / np.abs(y_ref).max() < REL
    dd = op.assemble_diagonal(torch.empty_like(xd))
    d_ref = orc.diagonal()
    assert np.abs(dd.cpu().numpy() - d_ref).max() / np.abs(d_ref).max() < REL
    # ParOperator with essential rows + the Chebyshev smoother, fused step against PALACE_AMD_FUSED_STEP=0
    ess = np.unique(nd.offsets[:5].ravel())[:40].astype(np.int32)
    ctx = linalg.Context()
    A = linalg.ParOperator(ctx, op, ess, linalg.DIAG_ONE)
    S = linalg.chebyshev(ctx, A, order=3)
    assert S.fused_step()
    monkeypatch.setenv("PALACE_AMD_FUSED_STEP", "0")
    S0 = linalg.chebyshev(ctx, A, order=3)
    assert not S0.fused_step()
    b = np.random.default_rng(5).uniform(-1, 1, nd.ndofs)
    b[ess] = 0.0
    bd = torch.from_numpy(b).cuda()
    z = S.mult(bd, torch.empty_like(bd)).cpu().numpy()
    z0 = S0.mult(bd, torch.empty_like(bd)).cpu().numpy()
    assert np.linalg.norm(z - z0) < 1e-12 * np.linalg.norm(z0)


def test_dense_e_vector_layout_follows_the_mesh_numbering(capfd, monkeypatch):
    """The layout of a large block is chosen by timing its gather both ways at creation: a Kuhn-split cube, numbered element by element
    with shared entities at the same local index, keeps one row per dof; the same mesh with its elements shuffled (what an unstructured
    mesh looks like to the gather) gets the rows by element.  Either way `Mult` gives the same bits (same copies, same order)."""
    import torch

    from palace_amd import ceed
    from palace_amd.fem import tet

    monkeypatch.setenv("PALACE_AMD_DENSE_VERBOSE", "1")

    def make(mesh):
        nd = tet.NDTetSpace(mesh, 3)
        pts, wts = tet.tet_quadrature(4)
        interp, curl = nd.elem.tables(pts)
        geom, _ = _geom(mesh, pts, wts)
        _, b3 = util.make_ctx("identity")
        block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, curl_orients=nd.curl_orients)
        return nd, ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, ceed.QF_HDIV_33, b3, ceed.EVAL_CURL).finalize()

    m = tet.cube_tet_mesh(20)  # 48 000 tetrahedra, 0.9M order-3 dofs
    nd, op = make(m)
    assert op.dense_gather_form()[0] is False, capfd.readouterr().err
    perm = np.random.default_rng(0).permutation(m.ne)
    nds, ops = make(tet.TetMesh(m.nodes, m.elem_nodes[perm], m.attr[perm]))
    assert ops.dense_gather_form()[0] is True, capfd.readouterr().err
    monkeypatch.setenv("PALACE_AMD_DENSE_ELAYOUT", "block")
    _, opb = make(tet.TetMesh(m.nodes, m.elem_nodes[perm], m.attr[perm]))
    assert opb.dense_gather_form()[0] is False
    x = torch.rand(nds.ndofs, dtype=torch.float64, device="cuda")
    assert torch.equal(ops.mult(x, torch.empty_like(x)), opb.mult(x, torch.empty_like(x)))
