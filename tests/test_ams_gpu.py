"""The native coarse-level solvers (palace_amd/csrc/amg_solver.hip: AmgSolver, AmsSolver -- where the reference calls HYPRE's
BoomerAMG / AMS, linalg/amg.cpp, linalg/ams.cpp) on the device: one application against the host restatement of the cycle on
the hierarchy the library built, symmetry and definiteness of the preconditioners, preconditioned solves against sparse direct
solutions, and the p-multigrid solve of the bench with AMS on its coarsest level."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from palace_amd import ceed, linalg  # noqa: E402
from palace_amd.fem.fespace import H1HexSpace, NDHexSpace, lowest_order_gradient, vertex_coordinates  # noqa: E402
from palace_amd.fem.mesh import ogrid_cylinder  # noqa: E402
from oracle import palace_oracle as po  # noqa: E402


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.fixture(scope="module")
def problem():
    mesh = ogrid_cylinder(4, 10)  # 800 hexahedra
    q1d = 2
    geom = ceed.GeomFactorData(mesh, q1d)
    h1, nd = H1HexSpace(mesh, 1), NDHexSpace(mesh, 1)
    mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
    ident = ceed.coefficient_context(3)
    return dict(mesh=mesh, geom=geom, h1=h1, nd=nd, mass=mass, ident=ident)


def _eliminated(Asp, ess):
    import scipy.sparse as sp

    keep = np.ones(Asp.shape[0])
    keep[ess] = 0.0
    D = sp.diags(keep)
    return (D @ Asp @ D + sp.diags(1.0 - keep)).tocsr()


def test_amg_cycle_matches_restatement_and_solves(problem):
    """H1 diffusion with Dirichlet rows (the gradient-space problem of the auxiliary-space smoothers), order 1: the device V-cycle equals the
    restated one on the library's own hierarchy; as a preconditioner it is symmetric positive definite and PCG reaches the
    sparse direct solution in a mesh-independent handful of iterations."""
    import scipy.sparse.linalg as spl

    h1, geom = problem["h1"], problem["geom"]
    op = ceed.diffusion_operator(geom, h1, problem["mass"])
    ess = h1.ess_dofs()
    ctx = linalg.Context()
    csr = op.full_assemble_device()
    B = linalg.amg(ctx, csr, ess, coarse_size=60)
    A_l, P_l, cinv = linalg.amg_hierarchy(B)
    assert len(A_l) >= 2 and cinv is not None and A_l[-1].shape[0] <= 240
    Asp = _eliminated(op.full_assemble(), ess)
    assert abs(A_l[0] - Asp).max() < 1e-13 * abs(Asp).max()
    rng = np.random.default_rng(0)
    b = rng.uniform(-1, 1, h1.ndofs)
    y = B.mult(_dev(b), torch.full((h1.ndofs,), np.nan, dtype=torch.float64, device="cuda")).cpu().numpy()
    ref = po.amg_vcycle(A_l, P_l, cinv, b)
    assert _rel(y, ref) < 1e-10
    # symmetric, positive definite
    c = rng.uniform(-1, 1, h1.ndofs)
    z = B.mult(_dev(c), torch.empty(h1.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
    assert abs(c @ y - b @ z) < 1e-10 * abs(c @ y) and b @ y > 0
    # PCG with it against the sparse direct solution
    A = linalg.AssembledParOperator(ctx, csr, ess, linalg.DIAG_ONE)
    rhs = b.copy()
    rhs[ess] = 0.0
    K = linalg.cg(ctx, A, B, rel_tol=1e-10, max_it=100)
    x = K.mult(_dev(rhs), torch.zeros(h1.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
    st = K.stats()
    assert st["converged"] and st["iterations"] <= 25, st
    assert _rel(x, spl.spsolve(Asp.tocsc(), rhs)) < 1e-8


@pytest.mark.parametrize("singular", [False, True])
def test_ams_cycle_matches_restatement_and_solves(problem, singular):
    """Lowest-order Nedelec curl-curl + mass (the coarsest level of the bench's p-multigrid): one AMS application against the
    restated cycle (auxiliary solves through the restated V-cycles on the library's hierarchies), symmetry, and PCG against the
    direct solution -- far fewer iterations than with the Jacobi preconditioner."""
    import scipy.sparse.linalg as spl

    nd, h1, geom = problem["nd"], problem["h1"], problem["geom"]
    op = ceed.curlcurlmass_operator(geom, nd, problem["mass"], problem["ident"])
    ess = nd.ess_dofs()
    G, X = lowest_order_gradient(h1, nd), vertex_coordinates(h1)
    # the gradient of a linear function is its (constant) direction in every edge dof: G x_c = edge vectors, curl-free
    Ksp = ceed.curlcurl_operator(geom, nd, problem["ident"]).full_assemble()
    assert abs(Ksp @ (G @ X[:, 0])).max() < 1e-10 * abs(Ksp).max()
    ctx = linalg.Context()
    csr = op.full_assemble_device()
    B = linalg.ams(ctx, csr, ess, G, X, singular=singular, amg_coarse_size=60)
    Asp = _eliminated(op.full_assemble(), ess)
    flag = np.zeros(nd.ndofs, dtype=bool)
    flag[ess] = True
    Gb, Pi = po.ams_nodal_interpolation(G, X, flag)
    hp = linalg.amg_hierarchy(B, 2)
    solve_pi = lambda r: po.amg_vcycle(*hp, r)  # noqa: E731
    if singular:
        solve_g = None
    else:
        hg = linalg.amg_hierarchy(B, 1)
        solve_g = lambda r: po.amg_vcycle(*hg, r)  # noqa: E731
    rng = np.random.default_rng(1)
    b = rng.uniform(-1, 1, nd.ndofs)
    b[ess] = 0.0
    y = B.mult(_dev(b), torch.full((nd.ndofs,), np.nan, dtype=torch.float64, device="cuda")).cpu().numpy()
    ref = po.ams_cycle(Asp, Gb, Pi, solve_g, solve_pi, b, singular=singular)
    assert _rel(y, ref) < 1e-9
    assert np.all(y[ess] == 0.0)
    c = rng.uniform(-1, 1, nd.ndofs)
    c[ess] = 0.0
    z = B.mult(_dev(c), torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
    assert abs(c @ y - b @ z) < 1e-9 * abs(c @ y) and b @ y > 0
    A = linalg.AssembledParOperator(ctx, csr, ess, linalg.DIAG_ONE)
    xs = spl.spsolve(Asp.tocsc(), b)
    its = {}
    for name, pc in (("ams", B), ("jacobi", linalg.jacobi(ctx, A))):
        K = linalg.cg(ctx, A, pc, rel_tol=1e-10, max_it=2000)
        x = K.mult(_dev(b), torch.zeros(nd.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
        st = K.stats()
        assert st["converged"], (name, st)
        assert _rel(x, xs) < 1e-7
        its[name] = st["iterations"]
    # (the singular option leaves the gradients to the smoother: on this operator, which HAS a mass term, that costs iterations)
    if singular:
        assert its["ams"] <= 80 and its["ams"] * 2 < its["jacobi"], its
    else:
        assert its["ams"] <= 30 and its["ams"] * 4 < its["jacobi"], its


def test_pmg_with_ams_on_the_coarsest_level():
    """The bench's PCG + p-multigrid (auxiliary-space smoothers) with the native AMS as the level-0 solver instead of the
    Jacobi-PCG stand-in: same solution, not more iterations."""
    from palace_amd.fem.partition import SlabProblem

    ctx = linalg.Context()
    prob = SlabProblem(ctx, 0, 1, 3, 2.0e5, levels=True)
    out = {}
    for coarse in ("cg", "ams"):
        K, b, x = prob.pcg_gmg_solver(max_it=200, rel_tol=1e-10, hiptmair=True, coarse=coarse)
        K.mult(b, x)
        st = K.stats()
        assert st["converged"], (coarse, st)
        out[coarse] = (st["iterations"], x.cpu().numpy())
        prob._keep.clear()
    assert out["ams"][0] <= out["cg"][0], (out["ams"][0], out["cg"][0])
    assert _rel(out["ams"][1], out["cg"][1]) < 1e-7


def test_cycles_with_the_steps_in_the_sparse_products_epilogues(problem, monkeypatch):
    """Round 6: the smoothers and residuals of the algebraic cycles run as CsrOperator::MultChebyStep / MultResidual (the accumulated
    form of the same polynomial, two launches per smoothing instead of six); PALACE_AMD_FUSED_STEP_CSR=0 at creation keeps product +
    vector kernel.  One AMG and one AMS application either way: equal to rounding (and each equal to the restated cycle: the tests
    above run with the default, fused, form)."""
    nd, h1, geom = problem["nd"], problem["h1"], problem["geom"]
    ctx = linalg.Context()
    rng = np.random.default_rng(3)
    op = ceed.curlcurlmass_operator(geom, nd, problem["mass"], problem["ident"])
    ess = nd.ess_dofs()
    G, X = lowest_order_gradient(h1, nd), vertex_coordinates(h1)
    csr = op.full_assemble_device()
    b = rng.uniform(-1, 1, nd.ndofs)
    out = {}
    for form in ("1", "0"):
        monkeypatch.setenv("PALACE_AMD_FUSED_STEP_CSR", form)
        B = linalg.ams(ctx, csr, ess, G, X, amg_coarse_size=60)
        out[form] = B.mult(_dev(b), torch.full((nd.ndofs,), np.nan, dtype=torch.float64, device="cuda")).cpu().numpy()
    assert _rel(out["1"], out["0"]) < 1e-12
    assert not np.array_equal(out["1"], out["0"]), "both runs took the same form?"


@pytest.mark.parametrize("policy", ["one", "zero"])
def test_chebyshev_on_an_assembled_operator_with_the_step_in_the_sparse_product(problem, monkeypatch, policy):
    """ChebyshevSmoother over ParOperator(CsrOperator) -- the level-0 smoother of the bench's plain p-multigrid: the step in the
    product's epilogue against product + vector kernel, zero and non-zero initial guess."""
    nd, geom = problem["nd"], problem["geom"]
    op = ceed.curlcurlmass_operator(geom, nd, problem["mass"], problem["ident"])
    ess = nd.ess_dofs()
    ctx = linalg.Context()
    csr = op.full_assemble_device()
    pol = linalg.DIAG_ONE if policy == "one" else linalg.DIAG_ZERO
    A = linalg.AssembledParOperator(ctx, csr, ess, pol)
    if policy == "zero":
        pytest.skip("a Jacobi-scaled smoother needs the unit diagonal on the essential rows")
    S = linalg.chebyshev(ctx, A, order=4)
    assert S.fused_step()
    monkeypatch.setenv("PALACE_AMD_FUSED_STEP_CSR", "0")
    S0 = linalg.chebyshev(ctx, A, order=4)
    assert not S0.fused_step() and S0.lambda_max() == S.lambda_max()
    rng = np.random.default_rng(8)
    n = nd.ndofs
    b, g = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    b[ess] = 0.0
    g[ess] = 0.0
    y = S.mult(_dev(b), torch.empty(n, dtype=torch.float64, device="cuda")).cpu().numpy()
    y0 = S0.mult(_dev(b), torch.empty(n, dtype=torch.float64, device="cuda")).cpu().numpy()
    z = S.mult(_dev(b), _dev(g.copy()), initial_guess=True).cpu().numpy()
    z0 = S0.mult(_dev(b), _dev(g.copy()), initial_guess=True).cpu().numpy()
    assert _rel(y, y0) < 1e-13 and _rel(z, z0) < 1e-13
