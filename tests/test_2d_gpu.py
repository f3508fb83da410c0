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
