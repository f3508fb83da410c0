"""Dense interpolators (p-prolongation, discrete gradient) and PCG + p-multigrid on tetrahedra."""
import numpy as np
import pytest

from oracle import palace_oracle as po

pytestmark = pytest.mark.gpu


def _warp(X):
    return X + 0.03 * np.sin(3 * X[:, [1, 2, 0]])


@pytest.mark.parametrize("pc,pf", [(1, 2), (2, 3), (1, 3)])
def test_dense_prolongation_matches_oracle(pc, pf):
    import torch

    from palace_amd import linalg
    from palace_amd.fem import tet

    ctx = linalg.Context()
    mesh = tet.to_quadratic(tet.cube_tet_mesh(3), _warp)
    c, f = tet.NDTetSpace(mesh, pc), tet.NDTetSpace(mesh, pf)
    M = tet.nd_tet_transfer_matrix(pc, pf)
    P = linalg.DenseInterp(ctx, c.restriction(), f.restriction(interp_range=True), M)
    orc = po.DenseInterpOracle(c.restriction(), f.restriction(interp_range=True), M)
    rng = np.random.default_rng(pc + 10 * pf)
    x = rng.uniform(-1, 1, c.ndofs)
    y = torch.empty(f.ndofs, dtype=torch.float64, device="cuda")
    P.mult(torch.from_numpy(x).cuda(), y)
    ref = orc.mult(x)
    assert np.abs(y.cpu().numpy() - ref).max() < 1e-12 * np.abs(ref).max()
    # the transpose is the exact transpose of the device operator (owner-copy form): <z, P x> = <P^T z, x>
    z = rng.uniform(-1, 1, f.ndofs)
    yt = torch.empty(c.ndofs, dtype=torch.float64, device="cuda")
    P.mult_transpose(torch.from_numpy(z).cuda(), yt)
    assert abs(z @ y.cpu().numpy() - yt.cpu().numpy() @ x) < 1e-12 * np.abs(z).sum()
    # and agrees with the reference's sum / multiplicity form on vectors that are consistent across
    # element copies (R^T of a range-space vector): compare through the oracle on P^T (P x)
    w = orc.mult(x)
    P.mult_transpose(torch.from_numpy(w).cuda(), yt)
    ref_t = orc.mult_transpose(w)
    assert np.abs(yt.cpu().numpy() - ref_t).max() < 1e-11 * np.abs(ref_t).max()


@pytest.mark.parametrize("p", [1, 2, 3])
def test_dense_gradient_matches_oracle_and_is_in_curl_kernel(p):
    import torch

    from palace_amd import ceed, linalg
    from palace_amd.fem import tet
    from palace_amd.fem.tetproblem import TetProblem

    ctx = linalg.Context()
    mesh = tet.to_quadratic(tet.cube_tet_mesh(3), _warp)
    prob = TetProblem(ctx, mesh, p, orders=[p])
    nd, h1 = prob.spaces[0], tet.H1TetSpace(mesh, p)
    Mg = tet.tet_gradient_matrix(p)
    G = linalg.DenseInterp(ctx, h1.restriction(), nd.restriction(interp_range=True), Mg)
    orc = po.DenseInterpOracle(h1.restriction(), nd.restriction(interp_range=True), Mg)
    phi = np.random.default_rng(p).uniform(-1, 1, h1.ndofs)
    g = torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")
    G.mult(torch.from_numpy(phi).cuda(), g)
    ref = orc.mult(phi)
    assert np.abs(g.cpu().numpy() - ref).max() < 1e-12 * np.abs(ref).max()
    # curl-curl annihilates discrete gradients (curved elements, curl-oriented restriction)
    K = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(
        prob.geom, prob.nd_block(nd), ceed.QF_HDIV_33, ceed.coefficient_context(3), ceed.EVAL_CURL).finalize()
    y = torch.empty_like(g)
    K.mult(g, y)
    Mm = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(
        prob.geom, prob.nd_block(nd), ceed.QF_HCURL_33, ceed.coefficient_context(3), ceed.EVAL_INTERP).finalize()
    ym = torch.empty_like(g)
    Mm.mult(g, ym)
    assert float(y.abs().max()) < 1e-10 * float(ym.abs().max())


@pytest.mark.parametrize("hiptmair", [False, True])
def test_tet_pcg_gmg_converges_to_direct_solution(hiptmair):
    import scipy.sparse.linalg as spl
    import torch

    from palace_amd import linalg
    from palace_amd.fem import tet
    from palace_amd.fem.tetproblem import TetProblem

    ctx = linalg.Context()
    mesh = tet.to_quadratic(tet.cube_tet_mesh(3), _warp)
    prob = TetProblem(ctx, mesh, 3)
    solver, b, x = prob.pcg_gmg_solver(max_it=200, rel_tol=1e-10, hiptmair=hiptmair)
    solver.mult(b, x)
    st = solver.stats()
    assert st["converged"] and st["iterations"] < (25 if hiptmair else 120), st
    # direct solve of the oracle-assembled system
    nd = prob.spaces[-1]
    interp, curl = nd.elem.tables(prob.pts)
    J = mesh.jacobians(prob.pts)
    geom = po.build_geom_factor_33(mesh.attr.astype(np.float64), prob.wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9))
    cm = po.CoeffCtx(attr_mat=[0], mat_coeff=[np.array([2.08])])
    A = po.CeedOperatorOracle(nd.ndofs, nd.offsets, None, interp, curl, geom, po.QF_HDIVMASS, cm, po.CoeffCtx(),
                              curl_orients=nd.curl_orients).assemble_sparse().tolil()
    ess = prob.ess[-1]
    A[ess, :] = 0.0
    A[:, ess] = 0.0
    for d in ess:
        A[d, d] = 1.0
    ref = spl.spsolve(A.tocsc(), b.cpu().numpy())
    err = np.abs(x.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 1e-8, err


@pytest.mark.parametrize("p", [1, 2, 3])
def test_tet_chebyshev_steps_fused_into_the_gather(monkeypatch, p):
    """Round 6: on a dense-table block the E^T gather owns every row, so the smoother step / residual is its epilogue
    (pa_op_mult_cheb_step, dense branch): Chebyshev on K + M of Nedelec tetrahedra (oriented and curl-oriented restrictions, curved
    elements) with essential dofs, fused against the same smoother with the step as a vector kernel (PALACE_AMD_FUSED_STEP=0), zero
    and non-zero initial guess; the PCG + p-multigrid solves of this file run through it as well."""
    import torch

    from palace_amd import ceed, linalg
    from palace_amd.fem import tet
    from palace_amd.fem.tetproblem import TetProblem

    ctx = linalg.Context()
    mesh = tet.to_quadratic(tet.cube_tet_mesh(3), _warp)
    prob = TetProblem(ctx, mesh, p, orders=[p])
    s = prob.spaces[0]
    mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
    op = ceed.Operator(s.ndofs, s.ndofs).add_dense_integrator(prob.geom, prob.nd_block(s), ceed.QF_HDIVMASS_33,
                                                              np.concatenate([mass, ceed.coefficient_context(3)]),
                                                              ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize()
    ess = s.ess_dofs()
    A = linalg.ParOperator(ctx, op, ess, linalg.DIAG_ONE)
    S = linalg.chebyshev(ctx, A, order=4)
    assert S.fused_step()
    monkeypatch.setenv("PALACE_AMD_FUSED_STEP", "0")
    S0 = linalg.chebyshev(ctx, A, order=4)
    assert not S0.fused_step() and S0.lambda_max() == S.lambda_max()
    n = s.ndofs
    rng = np.random.default_rng(41 + p)
    b, g = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    b[ess] = 0.0
    g[ess] = 0.0
    dev = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    y = S.mult(dev(b), torch.empty(n, dtype=torch.float64, device="cuda")).cpu().numpy()
    y0 = S0.mult(dev(b), torch.empty(n, dtype=torch.float64, device="cuda")).cpu().numpy()
    z = S.mult(dev(b), dev(g.copy()), initial_guess=True).cpu().numpy()
    z0 = S0.mult(dev(b), dev(g.copy()), initial_guess=True).cpu().numpy()
    assert np.linalg.norm(y - y0) < 1e-13 * np.linalg.norm(y0) and np.linalg.norm(z - z0) < 1e-13 * np.linalg.norm(z0)
    assert np.all(y[ess] == 0.0 * y[ess] + y0[ess])


@pytest.mark.parametrize("p", [2, 3])
def test_tet_chebyshev_step_with_surface_blocks(monkeypatch, p):
    """Round 6: a driven problem's operator is a volume block plus a few row-limited surface blocks (absorbing boundary, ports).  The
    fused smoother step takes them too: the surface blocks' element kernels, ONE gather over the union of their rows into a side
    vector, the volume block's gather adds it and runs the step.  Against the unfused smoother (PALACE_AMD_FUSED_STEP_SURFACE=0 at
    set-up: apply + essential rows + vector kernel), zero and non-zero initial guess."""
    import torch

    from palace_amd import ceed, linalg
    from palace_amd.fem import tet, tri
    from tests import util

    mesh = tet.cube_tet_mesh(6)
    nd = tet.NDTetSpace(mesh, p)
    vpts, vwts = tet.default_tet_rule(p)
    vint, vcurl = nd.elem.tables(vpts)
    vgeom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(vpts), vwts)
    kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
    vblock = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, vint, vcurl, **kw)
    _, bm = util.make_ctx("scalar", 1)
    _, bc = util.make_ctx("identity")
    _, b3 = util.make_ctx("aniso", 1)
    op = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(vgeom, vblock, ceed.QF_HDIVMASS_33, np.concatenate([bm, bc]),
                                                                 ceed.EVAL_CURL + ceed.EVAL_INTERP)
    faces = np.nonzero(mesh.boundary_face_mask)[0]
    pts, wts = tri.tri_quadrature(p + 1)
    for sub in (faces[:30], faces[20:55]):  # two surface blocks with common rows
        blk = tet.NDTetBoundaryBlock(nd, sub, np.ones(sub.size, dtype=np.int32))
        interp, _ = blk.elem.tables(pts)
        bgeom = ceed.DenseGeomFactorData(blk.elem_nodes, blk.nodes, blk.attr, blk.geometry_grad_table(pts), wts)
        op.add_dense_integrator(bgeom, ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, blk.offsets, interp, None, orients=blk.orients),
                                ceed.QF_HCURL_32, b3, ceed.EVAL_INTERP)
    op.finalize()
    ess = np.unique(nd.offsets[:8].ravel())[:60].astype(np.int32)
    ctx = linalg.Context()
    A = linalg.ParOperator(ctx, op, ess, linalg.DIAG_ONE)
    S = linalg.chebyshev(ctx, A, order=4)
    assert S.fused_step(), "the surface blocks are not row-limited on this mesh?"
    monkeypatch.setenv("PALACE_AMD_FUSED_STEP_SURFACE", "0")
    S0 = linalg.chebyshev(ctx, A, order=4)
    assert not S0.fused_step() and S0.lambda_max() == S.lambda_max()
    rng = np.random.default_rng(4)
    n = nd.ndofs
    b, g = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    b[ess] = 0.0
    g[ess] = 0.0
    dev = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    y = S.mult(dev(b), torch.empty(n, dtype=torch.float64, device="cuda")).cpu().numpy()
    y0 = S0.mult(dev(b), torch.empty(n, dtype=torch.float64, device="cuda")).cpu().numpy()
    z = S.mult(dev(b), dev(g.copy()), initial_guess=True).cpu().numpy()
    z0 = S0.mult(dev(b), dev(g.copy()), initial_guess=True).cpu().numpy()
    rel = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)  # noqa: E731
    assert rel(y, y0) < 1e-13 and rel(z, z0) < 1e-13
    # the plain apply is what it was
    x = rng.uniform(-1, 1, n)
    t0 = A.mult(dev(x), torch.empty(n, dtype=torch.float64, device="cuda")).cpu().numpy()
    assert np.array_equal(t0[ess], x[ess])
