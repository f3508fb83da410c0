"""2-D Nedelec triangles (the reference's examples/cavity2d) through the dense MFMA path:
f_apply_hcurl_22, f_apply_l2_1 (q_w input), f_apply_hdivmass_22 against the oracle on the reference's
own tri6 mesh, and the cavity eigenfrequency through device operators."""
import os

import numpy as np
import pytest

from oracle import palace_oracle as po

pytestmark = pytest.mark.gpu


def _mesh():
    from palace_amd.fem import tri

    M_ = np.load(os.path.join(os.path.dirname(__file__), "golden", "cavity2d_mesh.npz"))
    en = M_["elem_nodes"].astype(np.int64)
    used, inv = np.unique(en[:, :3], return_inverse=True)
    attr = 1 + (np.arange(en.shape[0]) % 2)
    return tri.TriMesh(M_["nodes"][used], inv.reshape(-1, 3), attr, elem_nodes=en, nodes=M_["nodes"]), M_


def _setup(p):
    from palace_amd import ceed
    from palace_amd.fem import tri

    mesh, M_ = _mesh()
    nd = tri.NDTriSpace(mesh, p)
    pts, wts = tri.tri_quadrature(p + 1)
    interp, curl = nd.elem.tables(pts)
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
    J = mesh.jacobians(pts)
    ogeom = po.build_geom_factor_22(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 4))
    block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, orients=nd.orients)
    return mesh, nd, wts, interp, curl, geom, ogeom, block, M_


def test_geometry_2d():
    mesh, nd, wts, interp, curl, geom, ogeom, block, _ = _setup(2)
    got = geom.to_numpy()
    assert got.shape == ogeom.shape
    assert np.abs(got - ogeom).max() <= 1e-13 * np.abs(ogeom).max()


@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("mode", ["curl", "mass", "curlmass"])
def test_apply_2d(p, mode):
    import torch

    from palace_amd import ceed

    mesh, nd, wts, interp, curl, geom, ogeom, block, _ = _setup(p)
    rng = np.random.default_rng(p)
    A = rng.uniform(-1, 1, (2, 2))
    c2 = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[A @ A.T + 2 * np.eye(2), np.array([0.6])], a=1.2, dim=2)
    c1 = po.CoeffCtx(attr_mat=[1, 0], mat_coeff=[np.array([1.9]), np.array([0.4])], dim=1)
    if mode == "curl":
        qf, oqf, blob, ctxs, ops = ceed.QF_L2_1, po.QF_L2_1, c1.pack(), (c1, None), ceed.EVAL_CURL | ceed.EVAL_WEIGHT
    elif mode == "mass":
        qf, oqf, blob, ctxs, ops = ceed.QF_HCURL_22, po.QF_HCURL_22, c2.pack(), (c2, None), ceed.EVAL_INTERP
    else:
        qf, oqf, blob, ctxs = ceed.QF_HDIVMASS_22, po.QF_HDIVMASS_22, np.concatenate([c2.pack(), c1.pack()]), (c2, c1)
        ops = ceed.EVAL_CURL | ceed.EVAL_INTERP | ceed.EVAL_WEIGHT
    op = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, qf, blob, ops).finalize()
    orc = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, interp, curl, ogeom, oqf, *ctxs, qw=wts)
    x = rng.uniform(-1, 1, nd.ndofs)
    ref = orc.apply_add(x, np.zeros(nd.ndofs))
    y = torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")
    op.mult(torch.from_numpy(x).cuda(), y)
    assert np.abs(y.cpu().numpy() - ref).max() < 1e-12 * np.abs(ref).max()
    d = torch.empty_like(y)
    op.assemble_diagonal(d)
    dref = orc.diagonal()
    assert np.abs(d.cpu().numpy() - dref).max() < 1e-12 * np.abs(dref).max()


@pytest.mark.parametrize("surface", [False, True])
@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("mode", ["diff", "mass", "diffmass"])
def test_h1_apply_2d_and_surface(p, mode, surface):
    """Nodal H1 triangles through the dense path: diffusion (f_apply_hcurl_22 / _32 on grad u, integ/diffusion.cpp), mass
    (f_apply_h1_1) and diffusion + mass (f_apply_hcurlmass_22 / _32) in the plane and on the same triangulation lifted to
    a curved surface in 3-D (2-D elements in 3-D space: what the auxiliary-space boundary terms of the preconditioner use,
    models/spaceoperator.cpp:321-326)."""
    import torch

    from palace_amd import ceed
    from palace_amd.fem import tri

    mesh, _ = _mesh()
    sp = tri.H1TriSpace(mesh, p)
    pts, wts = tri.tri_quadrature(p + 1)
    interp, grad = sp.elem.tables(pts)
    G = mesh.geometry_grad_table(pts)
    rng = np.random.default_rng(10 * p + surface)
    if surface:
        xy = mesh.nodes
        L = np.ptp(xy[:, 0])
        nodes = np.column_stack([xy, 0.15 * L * np.sin(3.0 * xy[:, 0] / L) + 0.3 * xy[:, 0] * xy[:, 1] / L])
        J = np.einsum("dqn,eni->eqid", G, nodes[mesh.elem_nodes])  # [e, q, 3, 2]
        ogeom = po.build_geom_factor_32(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 6))
        B = rng.uniform(-1, 1, (3, 3))
        cm = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[B @ B.T + 2 * np.eye(3), np.array([0.8])], a=0.9)
        qf_d, oqf_d, qf_dm, oqf_dm = ceed.QF_HCURL_32, po.QF_HCURL_32, ceed.QF_HCURLMASS_32, po.QF_HCURLMASS_32
    else:
        nodes = mesh.nodes
        J = mesh.jacobians(pts)
        ogeom = po.build_geom_factor_22(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 4))
        A = rng.uniform(-1, 1, (2, 2))
        cm = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[A @ A.T + 2 * np.eye(2), np.array([0.6])], a=1.2, dim=2)
        qf_d, oqf_d, qf_dm, oqf_dm = ceed.QF_HCURL_22, po.QF_HCURL_22, ceed.QF_HCURLMASS_22, po.QF_HCURLMASS_22
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, nodes, mesh.attr, G, wts)
    got = geom.to_numpy()
    assert got.shape == ogeom.shape and np.abs(got - ogeom).max() <= 1e-13 * np.abs(ogeom).max()
    c1 = po.CoeffCtx(attr_mat=[1, 0], mat_coeff=[np.array([1.9]), np.array([0.4])], dim=1)
    if mode == "diff":
        qf, oqf, blob, ctxs, ops = qf_d, oqf_d, cm.pack(), (cm, None), ceed.EVAL_GRAD
    elif mode == "mass":
        qf, oqf, blob, ctxs, ops = ceed.QF_H1_1, po.QF_H1MASS, c1.pack(), (c1, None), ceed.EVAL_INTERP
    else:
        qf, oqf, blob, ctxs, ops = qf_dm, oqf_dm, np.concatenate([c1.pack(), cm.pack()]), (c1, cm), ceed.EVAL_GRAD | ceed.EVAL_INTERP
    block = ceed.DenseBlock(ceed.FE_H1, sp.ndofs, sp.offsets, interp, grad)
    op = ceed.Operator(sp.ndofs, sp.ndofs).add_dense_integrator(geom, block, qf, blob, ops).finalize()
    orc = po.CeedOperatorOracle(sp.ndofs, sp.offsets, None, interp, grad, ogeom, oqf, *ctxs, vector_fe=False)
    x = rng.uniform(-1, 1, sp.ndofs)
    ref = orc.apply_add(x, np.zeros(sp.ndofs))
    y = torch.empty(sp.ndofs, dtype=torch.float64, device="cuda")
    op.mult(torch.from_numpy(x).cuda(), y)
    assert np.abs(y.cpu().numpy() - ref).max() < 1e-12 * np.abs(ref).max()
    d = torch.empty_like(y)
    op.assemble_diagonal(d)
    dref = orc.diagonal()
    assert np.abs(d.cpu().numpy() - dref).max() < 1e-12 * np.abs(dref).max()
    if mode != "mass":  # constants are in the kernel of the diffusion part
        ones = torch.ones_like(y)
        op.mult(ones, d)
        if mode == "diff":
            assert float(d.abs().max()) < 1e-12 * float(y.abs().max())


def test_cavity2d_first_mode_on_device():
    """Rayleigh quotient of the oracle's first eigenvector evaluated with the DEVICE operators reproduces the
    reference's first eigenfrequency (eig.csv, 0.1039343283770 GHz)."""
    import scipy.sparse.linalg as spl
    import torch

    from palace_amd import ceed
    from palace_amd.fem import tri

    M_ = np.load(os.path.join(os.path.dirname(__file__), "golden", "cavity2d_mesh.npz"))
    en = M_["elem_nodes"].astype(np.int64)
    used, inv = np.unique(en[:, :3], return_inverse=True)
    mesh = tri.TriMesh(M_["nodes"][used], inv.reshape(-1, 3), M_["attr"], elem_nodes=en, nodes=M_["nodes"])
    nd = tri.NDTriSpace(mesh, 2)
    pts, wts = tri.tri_quadrature(3)
    interp, curl = nd.elem.tables(pts)
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
    block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, orients=nd.orients)
    K = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, ceed.QF_L2_1, ceed.coefficient_context(1),
                                                               ceed.EVAL_CURL | ceed.EVAL_WEIGHT).finalize()
    Mm = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, ceed.QF_HCURL_22, ceed.coefficient_context(2),
                                                                ceed.EVAL_INTERP).finalize()
    Ks, Ms = K.full_assemble(), Mm.full_assemble()
    free = np.setdiff1d(np.arange(nd.ndofs), nd.ess_dofs())
    lam, vec = spl.eigsh(Ks[free][:, free].tocsc(), k=1, M=Ms[free][:, free].tocsc(), sigma=9.9, which="LM")
    x = np.zeros(nd.ndofs)
    x[free] = vec[:, 0]
    xd = torch.from_numpy(x).cuda()
    kx, mx = torch.empty_like(xd), torch.empty_like(xd)
    K.mult(xd, kx)
    Mm.mult(xd, mx)
    rq = float(xd @ kx) / float(xd @ mx)
    f = np.sqrt(rq / (2.08 * (1.0 - 4e-4j))) * 299792458.0 / (2 * np.pi) / 1e9
    assert abs(f.real - M_["eig_re_GHz"][0]) < 1e-8 * M_["eig_re_GHz"][0]
    assert abs(f.imag - M_["eig_im_GHz"][0]) < 1e-5 * M_["eig_im_GHz"][0]


def test_cavity2d_magnetostatic_inductance_on_device():
    """Config 4's magnetostatic half on the reference's own case (examples/cavity2d/cavity2d_magnetostatic.json: order 2, PEC
    walls, unit surface current along +x on the bottom edge; "Linear": AMS + CG, Tol 1e-8): the singular 2-D curl-curl system
    solved ON THE DEVICE (palace_amd/fem/triproblem.py: PCG + p-multigrid of orders 1, 2 with plain Chebyshev smoothers,
    iodata.cpp:533-564, the native AMS in its singular mode on the assembled order-1 level, linalg/ams.cpp:28-30) and
    M11 against the reference's regression value 6.283185306350e-07 H (test/data/regression/ref/cavity2d/magnetostatic/
    terminal-M.csv) to 1e-7 (the field is exactly representable; the reference's own gate for this file is looser)."""
    from palace_amd import linalg
    from palace_amd.fem import tri, triproblem

    mesh, bv, battr, M_ = triproblem.load_cavity2d(os.path.join(os.path.dirname(__file__), "golden", "cavity2d_mesh.npz"))
    ctx = linalg.Context()
    out = triproblem.magnetostatic_inductance(ctx, mesh, bv, battr, 2, [1.0, 0.0], order=2, rel_tol=1e-8, max_it=100)
    assert out["width"] == pytest.approx(1.0, abs=1e-12)
    assert out["converged"] and out["iterations"] <= 40, out["iterations"]
    assert out["rel_residual"] < 1e-6, out["rel_residual"]
    assert out["M11"] == pytest.approx(float(M_["M11_H"]), rel=1e-7)
    # the device operator of the solve against the oracle's (CPU restatement of f_apply_l2_1) on the computed field
    nd, ess = out["spaces"][-1], out["ess"][-1]
    pts, wts = tri.tri_quadrature(3)
    J = mesh.jacobians(pts)
    og = po.build_geom_factor_22(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 4))
    interp, curl = nd.elem.tables(pts)
    Ko = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients, interp, curl, og, po.QF_L2_1, po.CoeffCtx(dim=1), qw=wts)
    xo = out["x"].cpu().numpy()
    xz = xo.copy()
    xz[ess] = 0.0
    ko = Ko.apply_add(xz, np.zeros_like(xo))
    ko[ess] = xo[ess]
    assert np.linalg.norm(ko - out["Kx"].cpu().numpy()) < 1e-11 * np.linalg.norm(ko)
