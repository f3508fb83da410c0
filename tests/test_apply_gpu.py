"""Parity of the HIP operator apply (through the C ABI) against the oracle.

Criterion = the reference's own operator test, test/unit/test-libceed.cpp:245-282:
||y_test - y_ref||^2 < 1e-12 * max(||y_ref||^2, 1) on a random x (we gate the relative L2 error at
1e-12, far tighter)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from palace_amd import ceed  # noqa: E402
from palace_amd.fem.fespace import NDHexSpace  # noqa: E402
from palace_amd.fem.mesh import ogrid_cylinder, refine_uniform  # noqa: E402
from oracle import palace_oracle as po  # noqa: E402
from tests import util  # noqa: E402

RTOL = 1e-12


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _multi_attr(mesh):
    """Give the mesh three attributes so the attribute->material indirection is exercised."""
    m = type(mesh)(x=mesh.x, elem_nodes=mesh.elem_nodes, attr=(np.arange(mesh.ne) % 3 + 1).astype(np.int32))
    return m


def test_geometry_factors(cylinder_mesh):
    for q1d in (2, 3, 4, 5):
        g = ceed.GeomFactorData(cylinder_mesh, q1d).to_numpy()
        ref = util.oracle_geom(cylinder_mesh, q1d)
        assert np.array_equal(g[:, 0, :], ref[:, 0, :])
        np.testing.assert_allclose(g[:, 1:, :], ref[:, 1:, :], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("p", [1, 2, 3, 4])
@pytest.mark.parametrize("qf", ["hdiv", "hcurl", "hdivmass"])
def test_apply_cylinder_mesh(cylinder_mesh, p, qf):
    mesh = _multi_attr(cylinder_mesh)
    q1d = p + 1
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, q1d)
    _, b_a = util.make_ctx("aniso", nattr=3)
    _, b_s = util.make_ctx("scalar", nattr=3)
    dense = util.dense_tables(nd, q1d)
    if qf == "hdiv":
        op, blob = ceed.curlcurl_operator(geom, nd, b_a, dense), b_a
    elif qf == "hcurl":
        op, blob = ceed.ndmass_operator(geom, nd, b_a, dense), b_a
    else:
        op, blob = ceed.curlcurlmass_operator(geom, nd, b_s, b_a, dense), np.concatenate([b_s, b_a])
    rng = np.random.default_rng(1)
    x = rng.uniform(0, 1, nd.ndofs)
    y = op.mult(_dev(x), torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
    ref = util.oracle_apply_c(nd, util.oracle_geom(mesh, q1d), qf, blob, x, q1d)
    assert _rel(y, ref) < RTOL
    # AddMult accumulates
    y2 = op.add_mult(_dev(x), _dev(ref.copy())).cpu().numpy()
    assert _rel(y2, 2 * ref) < RTOL


@pytest.mark.parametrize("p", [1, 2, 3, 4])
@pytest.mark.parametrize("qf", ["hdiv", "hcurl", "hdivmass"])
@pytest.mark.parametrize("variant", ["matrix_free", "nonsym", "iso_matrix_free", "atomic", "metric", "iso_qdata"])
def test_apply_variants(cylinder_mesh, monkeypatch, p, qf, variant):
    """The other forms of the same kernel: D recomputed from the geometry factors exactly as the
    reference QFunctions do (PALACE_AMD_QDATA=0: general / isotropic coefficient; always for a
    non-symmetric coefficient) and E^T as an atomic scatter (PALACE_AMD_SCATTER=atomic)."""
    if variant in ("matrix_free", "iso_matrix_free"):
        monkeypatch.setenv("PALACE_AMD_QDATA", "0")
    if variant == "atomic":
        monkeypatch.setenv("PALACE_AMD_SCATTER", "atomic")
    if variant == "metric":     # isotropic coefficients through H = (w/|detJ|) J^T J for every operator type
        monkeypatch.setenv("PALACE_AMD_DSTAGE", "metric")
    if variant == "iso_qdata":  # ... and through the per-operator packed D
        monkeypatch.setenv("PALACE_AMD_DSTAGE", "qdata")
    mesh = _multi_attr(cylinder_mesh)
    q1d = p + 1
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, q1d)
    kind = {"matrix_free": "aniso", "nonsym": "nonsym", "iso_matrix_free": "scalar", "atomic": "aniso", "metric": "scalar",
            "iso_qdata": "scalar"}[variant]
    _, b_a = util.make_ctx(kind, nattr=3)
    _, b_s = util.make_ctx("scalar", nattr=3)
    if qf == "hdiv":
        op, blob = ceed.curlcurl_operator(geom, nd, b_a), b_a
    elif qf == "hcurl":
        op, blob = ceed.ndmass_operator(geom, nd, b_a), b_a
    else:
        op, blob = ceed.curlcurlmass_operator(geom, nd, b_s, b_a), np.concatenate([b_s, b_a])
    x = np.random.default_rng(2).uniform(-1, 1, nd.ndofs)
    y = op.mult(_dev(x), torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
    ref = util.oracle_apply_c(nd, util.oracle_geom(mesh, q1d), qf, blob, x, q1d)
    assert _rel(y, ref) < RTOL
    y2 = op.add_mult(_dev(x), _dev(ref.copy())).cpu().numpy()
    assert _rel(y2, 2 * ref) < RTOL


@pytest.mark.parametrize("p,q1d", [(1, 3), (2, 4), (2, 5), (3, 5), (1, 5)])
@pytest.mark.parametrize("qf", ["hdiv", "hdivmass"])
@pytest.mark.parametrize("variant", ["qdata", "matrix_free"])
def test_apply_overintegrated(cylinder_mesh, monkeypatch, p, q1d, qf, variant):
    """Coarse-level shapes: basis of order p on a finer rule (every instantiated (P1, Q1) pair)."""
    if variant == "matrix_free":
        monkeypatch.setenv("PALACE_AMD_QDATA", "0")
    mesh = _multi_attr(cylinder_mesh)
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, q1d)
    _, b_a = util.make_ctx("aniso", nattr=3)
    _, b_s = util.make_ctx("scalar", nattr=3)
    if qf == "hdiv":
        op, blob = ceed.curlcurl_operator(geom, nd, b_a), b_a
    else:
        op, blob = ceed.curlcurlmass_operator(geom, nd, b_s, b_a), np.concatenate([b_s, b_a])
    x = np.random.default_rng(3).uniform(-1, 1, nd.ndofs)
    y = op.mult(_dev(x), torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
    ref = util.oracle_apply_c(nd, util.oracle_geom(mesh, q1d), qf, blob, x, q1d)
    assert _rel(y, ref) < RTOL


@pytest.mark.parametrize("p", [1, 2, 3, 4])
@pytest.mark.parametrize("qf", ["hdiv", "hdivmass"])
@pytest.mark.parametrize("coef", ["scalar", "aniso"])
def test_two_right_hand_sides(cylinder_mesh, p, qf, coef):
    """pa_op_mult2: both vectors through one pass of the one-shot kernel against two separate applies (which take the
    streaming kernel where it exists: the same contractions compiled into another kernel, so the results agree to the last
    bits -- FMA contraction may differ -- rather than bit for bit)."""
    mesh = _multi_attr(cylinder_mesh)
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, p + 1)
    _, b_a = util.make_ctx(coef, nattr=3)
    _, b_s = util.make_ctx("scalar", nattr=3)
    op = ceed.curlcurl_operator(geom, nd, b_a) if qf == "hdiv" else ceed.curlcurlmass_operator(geom, nd, b_s, b_a)
    rng = np.random.default_rng(4)
    x0, x1 = _dev(rng.uniform(-1, 1, nd.ndofs)), _dev(rng.uniform(-1, 1, nd.ndofs))
    y0, y1, r0, r1 = (torch.empty_like(x0) for _ in range(4))
    op.mult2(x0, x1, y0, y1)
    op.mult(x0, r0)
    op.mult(x1, r1)
    for y, r in ((y0, r0), (y1, r1)):
        assert float((y - r).abs().max()) <= 4e-15 * float(r.abs().max())


@pytest.mark.parametrize("p", [1, 2, 3, 4])
@pytest.mark.parametrize("coef", ["aniso", "nonsym"])
def test_mixed_curl_forms(cylinder_mesh, p, coef):
    """MixedVectorWeakCurlIntegrator (C u, curl v) and MixedVectorCurlIntegrator (C curl u, v) on one H(curl) space
    (fem/integ/mixedveccurl.cpp:21-120; f_apply_hcurlhdiv_33 / f_apply_hdivhcurl_33, the oracle's restatements are pinned on
    vectors from the reference header): apply, the transposes (each form is the other one's transpose with C^T), the
    diagonal, and the sum of the pair SpaceOperator adds for Floquet-periodic problems (spaceoperator.cpp:305-309: weak curl
    with C, mixed curl with C transposed -- a skew pair when C is symmetric)."""
    mesh = _multi_attr(cylinder_mesh)
    q1d = p + 1
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, q1d)
    ogeom = util.oracle_geom(mesh, q1d)
    c, blob = util.make_ctx(coef, nattr=3)
    ct = po.CoeffCtx.__new__(po.CoeffCtx)
    ct.dim, ct.attr_mat = c.dim, c.attr_mat
    ct.mat = np.array([m.reshape(3, 3).T.reshape(-1) for m in c.mat])  # C^T per material
    x = np.random.default_rng(p).uniform(-1, 1, nd.ndofs)
    new = lambda: torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")  # noqa: E731
    for kind, other, build in (("hcurlhdiv", "hdivhcurl", ceed.weakcurl_operator), ("hdivhcurl", "hcurlhdiv", ceed.mixedcurl_operator)):
        op = build(geom, nd, blob)
        o = util.oracle_operator(nd, ogeom, kind, c, None, q1d)
        ref = o.apply_add(x, np.zeros(nd.ndofs))
        assert np.linalg.norm(ref) > 1e-3
        assert _rel(op.mult(_dev(x), new()).cpu().numpy(), ref) < RTOL, kind
        y0 = np.random.default_rng(7).uniform(-1, 1, nd.ndofs)
        ya = _dev(y0.copy())
        op.add_mult(_dev(x), ya)
        assert _rel(ya.cpu().numpy(), y0 + ref) < RTOL, kind
        assert not op.is_symmetric()
        ot = util.oracle_operator(nd, ogeom, other, ct, None, q1d)
        assert _rel(op.mult_transpose(_dev(x), new()).cpu().numpy(), ot.apply_add(x, np.zeros(nd.ndofs))) < RTOL, (kind, "T")
        if p <= 2:
            d = op.assemble_diagonal(new()).cpu().numpy()
            dref = o.diagonal()
            assert np.abs(d - dref).max() < 1e-12 * max(1.0, np.abs(dref).max()), kind
    # the Floquet pair in one operator (two sub-operators)
    pair = ceed.Operator(nd.ndofs, nd.ndofs)
    pair.add_integrator(geom, nd, ceed.QF_HCURLHDIV_33, blob, ceed.EVAL_INTERP, test_ops=ceed.EVAL_CURL)
    pair.add_integrator(geom, nd, ceed.QF_HDIVHCURL_33, ct.pack(), ceed.EVAL_CURL, test_ops=ceed.EVAL_INTERP)
    pair.finalize()
    ref = (util.oracle_operator(nd, ogeom, "hcurlhdiv", c, None, q1d).apply_add(x, np.zeros(nd.ndofs))
           + util.oracle_operator(nd, ogeom, "hdivhcurl", ct, None, q1d).apply_add(x, np.zeros(nd.ndofs)))
    assert _rel(pair.mult(_dev(x), new()).cpu().numpy(), ref) < RTOL


@pytest.mark.parametrize("p_coarse,p_fine", [(1, 3), (2, 3), (1, 2), (2, 4), (1, 4)])
def test_coarsened_operator(cylinder_mesh, p_coarse, p_fine):
    """CeedOperatorCoarsen: coarse basis on the fine level's quadrature/geometry data."""
    mesh = cylinder_mesh
    q1d = p_fine + 1
    ndf, ndc = NDHexSpace(mesh, p_fine), NDHexSpace(mesh, p_coarse)
    geom = ceed.GeomFactorData(mesh, q1d)
    _, b_s = util.make_ctx("scalar")
    _, b_i = util.make_ctx("identity")
    fine = ceed.curlcurlmass_operator(geom, ndf, b_s, b_i)
    coarse = fine.coarsen(geom, ndc)
    x = np.random.default_rng(2).uniform(-1, 1, ndc.ndofs)
    y = coarse.mult(_dev(x), torch.empty(ndc.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
    ref = util.oracle_apply_c(ndc, util.oracle_geom(mesh, q1d), "hdivmass", np.concatenate([b_s, b_i]), x, q1d)
    assert _rel(y, ref) < RTOL


@pytest.mark.parametrize("p", [1, 2, 3])
def test_diagonal(cylinder_mesh, p):
    mesh = _multi_attr(cylinder_mesh)
    q1d = p + 1
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, q1d)
    cs, b_s = util.make_ctx("scalar", nattr=3)
    ca, b_a = util.make_ctx("aniso", nattr=3)
    op = ceed.curlcurlmass_operator(geom, nd, b_s, b_a)
    d = op.assemble_diagonal(torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
    ref = util.oracle_operator(nd, util.oracle_geom(mesh, q1d), "hdivmass", cs, ca, q1d).diagonal()
    assert _rel(d, ref) < RTOL


@pytest.mark.parametrize("n,nz", [(1, 1), (1, 3), (2, 3), (3, 2)])
def test_ragged_element_counts(n, nz):
    """Element counts that do not fill the last wave / workgroup (5, 15, 60, 90 elements)."""
    mesh = ogrid_cylinder(n, nz)
    p, q1d = 3, 4
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, q1d)
    _, b_s = util.make_ctx("scalar")
    _, b_i = util.make_ctx("identity")
    op = ceed.curlcurlmass_operator(geom, nd, b_s, b_i)
    x = np.random.default_rng(3).uniform(0, 1, nd.ndofs)
    y = op.mult(_dev(x), torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
    ref = util.oracle_apply_c(nd, util.oracle_geom(mesh, q1d), "hdivmass", np.concatenate([b_s, b_i]), x, q1d)
    assert _rel(y, ref) < RTOL


def test_refined_mesh_and_properties(cylinder_mesh):
    """640 elements (one uniform refinement of the reference mesh): parity, symmetry
    x^T K y = y^T K x, linearity, positive semi-definiteness of K."""
    mesh = refine_uniform(cylinder_mesh)
    p, q1d = 3, 4
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, q1d)
    _, b_i = util.make_ctx("identity")
    K = ceed.curlcurl_operator(geom, nd, b_i)
    rng = np.random.default_rng(4)
    x, z = rng.uniform(-1, 1, nd.ndofs), rng.uniform(-1, 1, nd.ndofs)
    n = nd.ndofs
    new = lambda: torch.empty(n, dtype=torch.float64, device="cuda")  # noqa: E731
    Kx, Kz = K.mult(_dev(x), new()).cpu().numpy(), K.mult(_dev(z), new()).cpu().numpy()
    ref = util.oracle_apply_c(nd, util.oracle_geom(mesh, q1d), "hdiv", b_i, x, q1d)
    assert _rel(Kx, ref) < RTOL
    assert abs(z @ Kx - x @ Kz) < 1e-11 * abs(z @ Kx)
    Kxz = K.mult(_dev(2.0 * x - 3.0 * z), new()).cpu().numpy()
    assert _rel(Kxz, 2.0 * Kx - 3.0 * Kz) < 1e-12
    assert x @ Kx > 0


def test_error_paths(cylinder_mesh):
    """Mismatched descriptors are rejected with a message (no exception crosses the ABI)."""
    from palace_amd import lib

    mesh = cylinder_mesh
    nd = NDHexSpace(mesh, 2)
    geom = ceed.GeomFactorData(mesh, 3)
    _, b_i = util.make_ctx("identity")
    with pytest.raises(lib.PalaceAmdError, match="evaluation modes"):
        ceed.Operator(nd.ndofs, nd.ndofs).add_integrator(geom, nd, ceed.QF_HDIV_33, b_i, ceed.EVAL_INTERP)
    geom4 = ceed.GeomFactorData(mesh, 4)
    bad = util.dense_tables(nd, 4)
    with pytest.raises(lib.PalaceAmdError, match="dense basis table"):
        ceed.curlcurl_operator(geom4, nd, b_i, dense=(bad[0] * 1.001, bad[1]))
    op = ceed.curlcurl_operator(geom, nd, b_i)
    x = torch.zeros(nd.ndofs, dtype=torch.float64, device="cuda")
    with pytest.raises(lib.PalaceAmdError, match="coefficient = 1.0"):
        op.add_mult(x, x, a=2.0)


@pytest.mark.parametrize("variant", ["w2g1", "w2g2", "park3g1", "park3g2"])
def test_geometry_from_the_nodes(cylinder_mesh, monkeypatch, variant):
    """The streaming curl-curl kernel with D recomputed from the 27 nodes of every element (PALACE_AMD_STREAM_GEOM=nodes: 648 B per
    element instead of 3 072 B of packed D; round 5) against the oracle, with a different isotropic coefficient per attribute, with and
    without essential dofs fused, on the reference's curved cylinder mesh once refined (a ragged last batch included)."""
    from palace_amd import linalg

    mesh = _multi_attr(refine_uniform(cylinder_mesh))
    p, q1d = 3, 4
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, q1d)
    coefs = [np.array([0.7]), np.array([1.9]), np.array([1.0])]
    blob = ceed.coefficient_context(3, attr_mat=[0, 1, 2], mat_coeff=coefs)
    octx = po.CoeffCtx(attr_mat=[0, 1, 2], mat_coeff=coefs)
    op = ceed.curlcurl_operator(geom, nd, blob)
    x = np.random.default_rng(11).uniform(-1, 1, nd.ndofs)
    ref = util.oracle_apply_c(nd, util.oracle_geom(mesh, q1d), "hdiv", octx.pack(), x, q1d)
    xd = _dev(x)
    y0 = op.mult(xd, torch.empty_like(xd)).cpu().numpy()
    monkeypatch.setenv("PALACE_AMD_STREAM_GEOM", "nodes")
    monkeypatch.setenv("PALACE_AMD_GEOMN_VARIANT", variant)
    y1 = op.mult(xd, torch.empty_like(xd)).cpu().numpy()
    assert _rel(y0, ref) < RTOL and _rel(y1, ref) < RTOL
    assert not np.array_equal(y0, y1)  # (the switch did take the other kernel: the two forms differ in the last bits)
    ctx = linalg.Context()
    ess = nd.ess_dofs()
    K = linalg.ParOperator(ctx, op, ess, linalg.DIAG_ONE)
    yk = K.mult(xd, torch.empty_like(xd)).cpu().numpy()
    xm = x.copy()
    xm[ess] = 0.0
    refk = util.oracle_apply_c(nd, util.oracle_geom(mesh, q1d), "hdiv", octx.pack(), xm, q1d)
    refk[ess] = x[ess]
    assert _rel(yk, refk) < RTOL


@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("form", ["curl_packed", "mass_packed", "km_metric", "km_packed12", "curl_metric"])
def test_affine_batches_of_the_streaming_kernel(monkeypatch, p, form):
    """Round 6: batches of four elements with constant Jacobians (the central block of the O-grid: 20 % of the elements) read the
    compact D -- 6 | 7 | 12 numbers per element instead of per point -- in the streaming kernel.  Every D form against the C
    oracle, plain and with essential dofs fused, next to the same operator built with the form switched off
    (PALACE_AMD_STREAM_AFFINE=0) and to the one-shot kernel (AddMult), which always reads the per-point data."""
    from palace_amd import linalg
    from palace_amd.fem.mesh import ogrid_cylinder

    mesh = _multi_attr(ogrid_cylinder(4, 5))
    q1d = 4
    nd = NDHexSpace(mesh, p)
    ogeom = util.oracle_geom(mesh, q1d)
    if form == "curl_metric":
        monkeypatch.setenv("PALACE_AMD_DSTAGE", "metric")
    _, b_a = util.make_ctx("aniso", nattr=3)
    _, b_s = util.make_ctx("scalar", nattr=3)

    def build():
        geom = ceed.GeomFactorData(mesh, q1d)
        if form in ("curl_packed", "curl_metric"):
            return geom, ceed.curlcurl_operator(geom, nd, b_s if form == "curl_metric" else b_a), "hdiv", (b_s if form == "curl_metric" else b_a)
        if form == "mass_packed":
            return geom, ceed.ndmass_operator(geom, nd, b_a), "hcurl", b_a
        if form == "km_metric":
            return geom, ceed.curlcurlmass_operator(geom, nd, b_s, b_s), "hdivmass", np.concatenate([b_s, b_s])
        return geom, ceed.curlcurlmass_operator(geom, nd, b_a, b_a), "hdivmass", np.concatenate([b_a, b_a])

    x = np.random.default_rng(3).uniform(-1, 1, nd.ndofs)
    xd = _dev(x)
    geom, op, qf, blob = build()
    assert op.streams()
    ne, n_aff, n_comp = op.stream_affine()
    assert ne == mesh.ne and n_aff == mesh.ne // 5 and 0 < n_comp <= n_aff and n_comp % 4 == 0, (ne, n_aff, n_comp)
    ref = util.oracle_apply_c(nd, ogeom, qf, blob, x, q1d)
    y = op.mult(xd, torch.empty_like(xd)).cpu().numpy()
    assert _rel(y, ref) < RTOL
    y_one_shot = op.add_mult(xd, torch.zeros_like(xd)).cpu().numpy()
    assert _rel(y_one_shot, ref) < RTOL
    ctx = linalg.Context()
    ess = nd.ess_dofs()
    K = linalg.ParOperator(ctx, op, ess, linalg.DIAG_ONE)
    yk = K.mult(xd, torch.empty_like(xd)).cpu().numpy()
    xm = x.copy()
    xm[ess] = 0.0
    refk = util.oracle_apply_c(nd, ogeom, qf, blob, xm, q1d)
    refk[ess] = x[ess]
    assert _rel(yk, refk) < RTOL
    monkeypatch.setenv("PALACE_AMD_STREAM_AFFINE", "0")
    geom0, op0, _, _ = build()
    assert op0.stream_affine()[2] == 0
    y0 = op0.mult(xd, torch.empty_like(xd)).cpu().numpy()
    assert _rel(y0, ref) < RTOL and _rel(y, y0) < 1e-13
    assert not np.array_equal(y, y0)  # (the compact rows were read: the two forms differ in the last bits)
