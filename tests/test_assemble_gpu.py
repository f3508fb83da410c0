"""pa_op_full_assemble (CeedOperatorFullAssemble, fem/libceed/operator.cpp:455-523) against the oracle's
element-matrix assembly — the reference pins its own full assembly the same way, against the legacy
assembled matrix (test/unit/test-libceed.cpp:284-300)."""
import numpy as np
import pytest

from oracle import palace_oracle as po
from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("p", [1, 2])
@pytest.mark.parametrize("skip_zeros", [False, True])
def test_full_assemble_hex(cylinder_mesh, p, skip_zeros):
    from palace_amd import ceed
    from palace_amd.fem.fespace import NDHexSpace

    mesh = cylinder_mesh
    q1d = p + 1
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, q1d)
    ca, ba = util.make_ctx("aniso", int(mesh.attr.max()))
    cs, bs = util.make_ctx("scalar", int(mesh.attr.max()))
    op = ceed.curlcurlmass_operator(geom, nd, bs, ba)
    A = op.full_assemble(skip_zeros=skip_zeros)
    ref = util.oracle_operator(nd, util.oracle_geom(mesh, q1d), "hdivmass", cs, ca, q1d).assemble_sparse()
    assert abs(A - ref).max() < 1e-12 * abs(ref).max()
    assert np.all(np.diff(A.indptr) > 0) and A.has_sorted_indices
    if skip_zeros:
        assert np.all(A.data != 0.0)
    # the assembled matrix reproduces Mult
    import torch

    x = np.random.default_rng(0).uniform(-1, 1, nd.ndofs)
    y = torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")
    op.mult(torch.from_numpy(x).cuda(), y)
    assert np.abs(A @ x - y.cpu().numpy()).max() < 1e-12 * np.abs(y.cpu().numpy()).max()


@pytest.mark.parametrize("p", [1, 2])
def test_full_assemble_tets(p):
    from palace_amd import ceed
    from palace_amd.fem import tet

    mesh = tet.cube_tet_mesh(2)
    nd = tet.NDTetSpace(mesh, p)
    pts, wts = tet.default_tet_rule(p)
    interp, curl = nd.elem.tables(pts)
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
    kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
    block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, **kw)
    op = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, ceed.QF_HDIV_33, ceed.coefficient_context(3),
                                                                ceed.EVAL_CURL).finalize()
    A = op.full_assemble()
    J = mesh.jacobians(pts)
    og = po.build_geom_factor_33(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9))
    ref = po.CeedOperatorOracle(nd.ndofs, nd.offsets, None if not nd.diagonal_transform else nd.orients, interp, curl, og,
                                po.QF_HDIV, po.CoeffCtx(),
                                curl_orients=None if nd.diagonal_transform else nd.curl_orients).assemble_sparse()
    assert abs(A - ref).max() < 1e-12 * abs(ref).max()


@pytest.mark.parametrize("policy", ["one", "zero"])
@pytest.mark.parametrize("p", [1, 2])
def test_assembled_par_operator(cylinder_mesh, p, policy):
    """ParOperator around the device CSR (pa_par_op_create_assembled; ParOperator::ParallelAssemble,
    linalg/rap.cpp:84-152) = ParOperator around the matrix-free operator: Mult, AddMult, diagonal, EliminateRHS."""
    import torch

    from palace_amd import ceed, linalg
    from palace_amd.fem.fespace import NDHexSpace

    mesh = cylinder_mesh
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, p + 1)
    _, bs = util.make_ctx("scalar", int(mesh.attr.max()))
    _, bc = util.make_ctx("scalar", int(mesh.attr.max()))
    op = ceed.curlcurlmass_operator(geom, nd, bs, bc)
    ctx = linalg.Context()
    ess = nd.ess_dofs()
    pol = linalg.DIAG_ONE if policy == "one" else linalg.DIAG_ZERO
    A_mf = linalg.ParOperator(ctx, op, ess, pol)
    csr = op.full_assemble_device()
    assert csr.nrows == nd.ndofs and csr.nnz > 0
    A_as = linalg.AssembledParOperator(ctx, csr, ess, pol)
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.uniform(-1, 1, nd.ndofs)).cuda()
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    A_mf.mult(x, y0)
    A_as.mult(x, y1)
    scale = float(y0.abs().max())
    assert float((y0 - y1).abs().max()) < 1e-12 * scale
    z0 = torch.from_numpy(rng.uniform(-1, 1, nd.ndofs)).cuda()
    z1 = z0.clone()
    A_mf.add_mult(x, z0, -0.7)
    A_as.add_mult(x, z1, -0.7)
    assert float((z0 - z1).abs().max()) < 1e-12 * scale
    d0, d1 = torch.empty_like(x), torch.empty_like(x)
    A_mf.assemble_diagonal(d0)
    A_as.assemble_diagonal(d1)
    assert float((d0 - d1).abs().max()) < 1e-12 * float(d0.abs().max())
    b0 = torch.from_numpy(rng.uniform(-1, 1, nd.ndofs)).cuda()
    b1 = b0.clone()
    A_mf.eliminate_rhs(x, b0)
    A_as.eliminate_rhs(x, b1)
    assert float((b0 - b1).abs().max()) < 1e-12 * max(scale, float(b0.abs().max()))
