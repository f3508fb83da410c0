"""Size-independent properties of the hot path at the BENCH size (ND p=3, ~10M dofs on one GPU), where
the oracle cannot run: symmetry, determinism, linearity, curl-curl of discrete gradients, the essential
rows of ParOperator, and the operator seen through a far smaller, oracle-checked problem of the same
family (the per-element action does not depend on how many elements there are)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def big():
    from palace_amd import linalg
    from palace_amd.fem.partition import SlabProblem

    ctx = linalg.Context()
    prob = SlabProblem(ctx, 0, 1, 3, 10.0e6, levels=False)
    return ctx, prob


def test_fullsize_symmetry_linearity_determinism(big):
    ctx, prob = big
    op = prob.local_curlcurl
    n = prob.n_local[-1]
    assert n > 9.0e6
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) - 0.5
    z = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) - 0.5
    Ax, Az = torch.empty_like(x), torch.empty_like(x)
    op.mult(x, Ax)
    op.mult(z, Az)
    a, b = ctx.dot(z, Ax), ctx.dot(x, Az)
    assert abs(a - b) <= 1e-12 * max(abs(a), abs(b))
    # the same launch twice gives the same bits (gather form of E^T: fixed summation order)
    Ax2 = torch.empty_like(x)
    op.mult(x, Ax2)
    assert torch.equal(Ax, Ax2)
    # linearity: A (2 x - 3 z) = 2 A x - 3 A z
    w = 2.0 * x - 3.0 * z
    Aw = torch.empty_like(x)
    op.mult(w, Aw)
    ref = 2.0 * Ax - 3.0 * Az
    assert float((Aw - ref).abs().max()) <= 1e-12 * float(ref.abs().max())
    # AddMult accumulates
    y = Az.clone()
    op.add_mult(x, y)
    assert float((y - (Az + Ax)).abs().max()) <= 1e-13 * float(Ax.abs().max())
    # positive semi-definite
    assert ctx.dot(x, Ax) > 0.0


def test_fullsize_curlcurl_annihilates_gradients_and_bc_rows(big):
    from palace_amd import ceed, linalg
    from palace_amd.fem.fespace import H1HexSpace

    ctx, prob = big
    nd = prob.spaces[-1]
    h1 = H1HexSpace(prob.mesh, 3)
    G = linalg.Gradient(ctx, h1, nd)
    phi = torch.rand(h1.ndofs, dtype=torch.float64, device="cuda")
    g = torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")
    G.mult(phi, g)
    Kg = torch.empty_like(g)
    prob.local_curlcurl.mult(g, Kg)
    mass = ceed.ndmass_operator(prob.geom, nd, ceed.coefficient_context(3))
    Mg = torch.empty_like(g)
    mass.mult(g, Mg)
    assert float(Kg.abs().max()) <= 1e-10 * float(Mg.abs().max())
    # ParOperator: essential rows are copied bit-exactly, the rest equals the local apply of the masked input
    K = prob.curlcurl_par_operator()
    x = torch.rand(nd.ndofs, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    K.mult(x, y)
    ess = torch.from_numpy(prob.ess[-1].astype(np.int64)).cuda()
    assert torch.equal(y[ess], x[ess])
    xm = x.clone()
    xm[ess] = 0.0
    ref = torch.empty_like(x)
    prob.local_curlcurl.mult(xm, ref)
    ref[ess] = x[ess]
    assert torch.equal(y, ref)


def test_midsize_matches_oracle():
    """The largest member of the same mesh family the C oracle still finishes in seconds (~200k dofs, the
    per-element action does not depend on the element count): full parity, 1e-12."""
    from oracle import palace_oracle as po
    from palace_amd import ceed
    from palace_amd.fem.fespace import NDHexSpace
    from palace_amd.fem.mesh import cylinder_for_dofs
    from tests import util

    mesh = cylinder_for_dofs(2.0e5, 3)
    nd = NDHexSpace(mesh, 3)
    geom = ceed.GeomFactorData(mesh, 4)
    op = ceed.curlcurl_operator(geom, nd, ceed.coefficient_context(3))
    x = np.random.default_rng(0).uniform(-1, 1, nd.ndofs)
    y = torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")
    op.mult(torch.from_numpy(x).cuda(), y)
    ref = util.oracle_apply_c(nd, util.oracle_geom(mesh, 4), "hdiv", po.CoeffCtx().pack(), x, 4)
    assert np.abs(y.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()


def test_fullsize_complex_one_pass_equals_separate_applies(big):
    """The one-pass complex apply (pa_op_mult_complex) at the bench size: y = (A_r + i A_i) x equals the four
    separate real applies, with essential dofs, is deterministic, and is complex-linear."""
    from palace_amd import ceed, linalg

    ctx, prob = big
    nd = prob.spaces[-1]
    n = nd.ndofs
    mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([-2.08 * 0.3])])
    cond = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([0.05])])
    Ar = ceed.curlcurlmass_operator(prob.geom, nd, mass, ceed.coefficient_context(3))
    Ai = ceed.ndmass_operator(prob.geom, nd, cond)
    assert ceed._lib.load().pa_op_complex_fused(Ar.handle, Ai.handle) == 1
    A = linalg.ComplexParOperator(ctx, Ar, Ai, prob.ess[-1], linalg.DIAG_ONE)
    g = torch.Generator(device="cuda").manual_seed(11)
    xr, xi = (torch.rand(n, dtype=torch.float64, device="cuda", generator=g) - 0.5 for _ in range(2))
    yr, yi = torch.empty_like(xr), torch.empty_like(xr)
    A.mult(xr, xi, yr, yi)
    y2r, y2i = torch.empty_like(xr), torch.empty_like(xr)
    A.mult(xr, xi, y2r, y2i)
    assert torch.equal(yr, y2r) and torch.equal(yi, y2i)
    # four real applies through the real ParOperators (fused essential handling of the real kernels)
    Pr = linalg.ParOperator(ctx, Ar, prob.ess[-1], linalg.DIAG_ONE)
    Pi = linalg.ParOperator(ctx, ceed.ndmass_operator(prob.geom, nd, cond), prob.ess[-1], linalg.DIAG_ZERO)
    t = [torch.empty_like(xr) for _ in range(4)]
    Pr.mult(xr, t[0]), Pi.mult(xi, t[1]), Pi.mult(xr, t[2]), Pr.mult(xi, t[3])
    rr, ri = t[0] - t[1], t[2] + t[3]
    assert float((yr - rr).abs().max()) <= 1e-13 * float(rr.abs().max())
    assert float((yi - ri).abs().max()) <= 1e-13 * float(ri.abs().max())
    # multiplication by i: A (i x) = i A x  (x -> (-xi, xr), y -> (-yi, yr)); essential rows carry x itself, so they follow too
    zr, zi = torch.empty_like(xr), torch.empty_like(xr)
    A.mult(-xi, xr, zr, zi)
    assert float((zr + yi).abs().max()) <= 1e-13 * float(yi.abs().max())
    assert float((zi - yr).abs().max()) <= 1e-13 * float(yr.abs().max())


def test_fullsize_tets_affine_form_equals_general_form():
    """280k straight-sided tetrahedra (the bench's tetrahedral leg): the affine form of the dense kernel (first-point D scaled by
    relative quadrature weights, 12 waves per CU) against the general form on the same elements."""
    import os

    from palace_amd import ceed
    from palace_amd.fem import tet

    mesh = tet.cube_tet_mesh(36)
    nd = tet.NDTetSpace(mesh, 3)
    pts, wts = tet.default_tet_rule(3)
    interp, curl = nd.elem.tables(pts)
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
    block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, curl_orients=nd.curl_orients)
    mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
    blob = np.concatenate([mass, ceed.coefficient_context(3)])
    n = nd.ndofs

    def build():
        return ceed.Operator(n, n).add_dense_integrator(geom, block, ceed.QF_HDIVMASS_33, blob, ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize()

    A = build()
    os.environ["PALACE_AMD_DENSE_AFFINE"] = "0"
    try:
        B = build()
    finally:
        del os.environ["PALACE_AMD_DENSE_AFFINE"]
    assert A.dense_affine() == 1 and B.dense_affine() == 0
    x = torch.rand(n, dtype=torch.float64, device="cuda") - 0.5
    ya, yb = torch.empty_like(x), torch.empty_like(x)
    A.mult(x, ya)
    B.mult(x, yb)
    assert float((ya - yb).abs().max()) <= 1e-14 * float(yb.abs().max())
    y2 = torch.empty_like(x)
    A.mult(x, y2)
    assert torch.equal(ya, y2)
