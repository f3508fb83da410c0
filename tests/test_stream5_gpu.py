"""The streaming kernel for five points per direction (pa_nd_hex_stream5.hip: p = 4, BASELINE config 5's element, and its
p-coarsened levels) against the oracle and against the one-shot kernel it replaces for `y = A x`.

Same criterion as tests/test_apply_gpu.py (test/unit/test-libceed.cpp:245-282, gated at 1e-12 relative l2)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from palace_amd import ceed, linalg  # noqa: E402
from palace_amd.fem.fespace import NDHexSpace  # noqa: E402
from palace_amd.fem.mesh import refine_uniform  # noqa: E402
from oracle import palace_oracle as po  # noqa: E402
from tests import util  # noqa: E402

RTOL = 1e-12
Q1D = 5


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _streams(op):
    return bool(ceed._lib.load().pa_op_streams(op.handle))


def _make(geom, nd, qf, kind):
    _, b_a = util.make_ctx(kind, nattr=3)
    _, b_s = util.make_ctx("scalar", nattr=3)
    if qf == "hdiv":
        return ceed.curlcurl_operator(geom, nd, b_a), b_a
    if qf == "hcurl":
        return ceed.ndmass_operator(geom, nd, b_a), b_a
    return ceed.curlcurlmass_operator(geom, nd, b_s, b_a), np.concatenate([b_s, b_a])


@pytest.fixture(scope="module")
def mesh640(cylinder_mesh):
    m = refine_uniform(cylinder_mesh)
    # three attributes: the attribute -> material indirection of the per-element coefficients
    return type(m)(x=m.x, elem_nodes=m.elem_nodes, attr=(np.arange(m.ne) % 3 + 1).astype(np.int32))


@pytest.mark.parametrize("p", [1, 2, 3, 4])
@pytest.mark.parametrize("qf,kind,dstage", [("hdiv", "aniso", None), ("hcurl", "aniso", None), ("hdiv", "scalar", "metric"),
                                            ("hcurl", "scalar", "metric"), ("hdivmass", "scalar", None)])
@pytest.mark.parametrize("wgx", [None, "1"])
def test_stream5_matches_oracle_and_one_shot(mesh640, monkeypatch, p, qf, kind, dstage, wgx):
    """Packed q-data (anisotropic coefficient) and the metric form (isotropic), every order on the five-point rule; with
    one workgroup per XCD every wave walks ~20 batches (the cross-batch pipeline), with the default launch ~1."""
    if dstage:
        monkeypatch.setenv("PALACE_AMD_DSTAGE", dstage)
    if wgx:
        monkeypatch.setenv("PALACE_AMD_STREAM_WGX", wgx)
    mesh = mesh640
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, Q1D)
    op, blob = _make(geom, nd, qf, kind)
    assert _streams(op), "the streaming kernel was not selected"
    x = np.random.default_rng(3).uniform(-1, 1, nd.ndofs)
    y = op.mult(_dev(x), torch.full((nd.ndofs,), np.nan, dtype=torch.float64, device="cuda")).cpu().numpy()
    ref = util.oracle_apply_c(nd, util.oracle_geom(mesh, Q1D), qf, blob, x, Q1D)
    assert _rel(y, ref) < RTOL
    # bit-reproducible (fixed summation order of the run gather)
    y2 = op.mult(_dev(x), torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
    assert np.array_equal(y, y2)
    # AddMult keeps the one-shot kernel: same operator, different schedule
    y3 = op.add_mult(_dev(x), _dev(ref.copy())).cpu().numpy()
    assert _rel(y3, 2 * ref) < RTOL
    # the one-shot form of the same operator (PALACE_AMD_STREAM5=0 at creation)
    monkeypatch.setenv("PALACE_AMD_STREAM5", "0")
    op1, _ = _make(geom, nd, qf, kind)
    assert not _streams(op1)
    y1 = op1.mult(_dev(x), torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
    assert _rel(y, y1) < 1e-13


@pytest.mark.parametrize("p", [2, 4])
@pytest.mark.parametrize("policy", ["one", "zero"])
def test_stream5_par_operator_essential_rows(mesh640, monkeypatch, p, policy):
    """ParOperator::Mult (rap.cpp:195-234) on the five-point rule: essential dofs read as zero inside the kernel, their rows
    written by the run gather (x or 0), bit-exactly."""
    monkeypatch.setenv("PALACE_AMD_STREAM_WGX", "2")
    mesh = mesh640
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, Q1D)
    cm, bm = util.make_ctx("scalar", nattr=3)
    cc, bc = util.make_ctx("identity")
    local = ceed.curlcurlmass_operator(geom, nd, bm, bc)
    assert _streams(local)
    ess = nd.ess_dofs()
    ctx = linalg.Context()
    pol = linalg.DIAG_ONE if policy == "one" else linalg.DIAG_ZERO
    A = linalg.ParOperator(ctx, local, ess, pol)
    x = np.random.default_rng(5).uniform(-1, 1, nd.ndofs)
    y = A.mult(_dev(x), torch.full((nd.ndofs,), np.nan, dtype=torch.float64, device="cuda")).cpu().numpy()
    oracle = util.FastParOperatorOracle(nd, util.oracle_geom(mesh, Q1D), "hdivmass", np.concatenate([bm, bc]), ess, Q1D, cm, cc,
                                        policy=po.DIAG_ONE if policy == "one" else po.DIAG_ZERO)
    ref = oracle.mult(x)
    assert _rel(y, ref) < RTOL
    assert np.array_equal(y[ess], x[ess] if policy == "one" else np.zeros(ess.size))


@pytest.mark.parametrize("p", [2, 4])
@pytest.mark.parametrize("policy", ["one"])  # (a Jacobi-scaled smoother needs the unit diagonal on the essential rows)
def test_stream5_chebyshev_steps_fused_into_the_gather(mesh640, monkeypatch, p, policy):
    """Round 6: the smoother step evaluated in the E^T epilogue (pa_op_mult_cheb_step) on the five-point kernel -- config 5's
    element and its p-coarsened levels: the fused smoother against the same smoother with the step as a vector kernel
    (PALACE_AMD_FUSED_STEP=0) and against the oracle's recurrence, zero and non-zero initial guess."""
    mesh = mesh640
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, Q1D)
    cm, bm = util.make_ctx("scalar", nattr=3)
    cc, bc = util.make_ctx("identity")
    local = ceed.curlcurlmass_operator(geom, nd, bm, bc)
    ess = nd.ess_dofs()
    ctx = linalg.Context()
    pol = linalg.DIAG_ONE if policy == "one" else linalg.DIAG_ZERO
    A = linalg.ParOperator(ctx, local, ess, pol)
    S = linalg.chebyshev(ctx, A, order=4)
    assert S.fused_step()
    monkeypatch.setenv("PALACE_AMD_FUSED_STEP", "0")
    S0 = linalg.chebyshev(ctx, A, order=4)
    assert not S0.fused_step() and S0.lambda_max() == S.lambda_max()
    n = nd.ndofs
    rng = np.random.default_rng(21)
    b, g = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    b[ess] = 0.0
    g[ess] = 0.0
    y = S.mult(_dev(b), torch.empty(n, dtype=torch.float64, device="cuda")).cpu().numpy()
    y0 = S0.mult(_dev(b), torch.empty(n, dtype=torch.float64, device="cuda")).cpu().numpy()
    z = S.mult(_dev(b), _dev(g.copy()), initial_guess=True).cpu().numpy()
    z0 = S0.mult(_dev(b), _dev(g.copy()), initial_guess=True).cpu().numpy()
    assert _rel(y, y0) < 1e-13 and _rel(z, z0) < 1e-13
    if policy == "one":
        oracle = util.FastParOperatorOracle(nd, util.oracle_geom(mesh, Q1D), "hdivmass", np.concatenate([bm, bc]), ess, Q1D, cm, cc)
        oracle._diag = A.assemble_diagonal(torch.empty(n, dtype=torch.float64, device="cuda")).cpu().numpy()
        o = po.ChebyshevOracle(oracle, 4, lambda_max=S.lambda_max())
        assert _rel(y, o.mult2(b, None, False)) < 1e-11 and _rel(z, o.mult2(b, g.copy(), True)) < 1e-11


def test_stream5_ragged_and_tiny_meshes(cylinder_mesh):
    """Odd element counts (the last batch holds one element and one pad), fewer batches than XCDs."""
    from palace_amd.fem.mesh import ogrid_cylinder

    for n, nz in ((1, 1), (1, 3), (2, 1)):
        mesh = ogrid_cylinder(n, nz)
        nd = NDHexSpace(mesh, 4)
        geom = ceed.GeomFactorData(mesh, Q1D)
        blob = po.CoeffCtx().pack()
        op = ceed.curlcurl_operator(geom, nd, blob)
        assert _streams(op)
        x = np.random.default_rng(7).uniform(-1, 1, nd.ndofs)
        y = op.mult(_dev(x), torch.full((nd.ndofs,), np.nan, dtype=torch.float64, device="cuda")).cpu().numpy()
        ref = util.oracle_apply_c(nd, util.oracle_geom(mesh, Q1D), "hdiv", blob, x, Q1D)
        assert _rel(y, ref) < RTOL, (n, nz, mesh.ne)
