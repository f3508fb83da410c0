"""Pin the oracle's D stage on the reference's own QFunctions.

Golden vectors in tests/golden/qf_golden.npz were produced by the *reference headers*
(palace/fem/qfunctions/33/*.h compiled into oracle/_ref, see tests/golden/make_golden.py); when
oracle/_ref exists (build container) the live library is exercised too."""
import os

import numpy as np
import pytest

from oracle import capi
from oracle import palace_oracle as po

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "qf_golden.npz"))
TOL = 1e-13


def _ctx_from_blob(blob):
    iv = blob.view(np.int32).reshape(-1, 2)[:, 0]
    nattr = iv[0]
    nmat = iv[1 + nattr]
    c = po.CoeffCtx()
    c.attr_mat = iv[1 : 1 + nattr].copy()
    c.mat = blob[2 + nattr : 2 + nattr + 9 * nmat].reshape(nmat, 9).copy()
    return c, 2 + nattr + 9 * nmat


def test_geom_factor_numpy_and_c():
    Q = int(G["Q"])
    J = G["J"]  # [9][Q]
    geom = po.build_geom_factor_33(G["attr"][:1] * 0 + 1.0, G["qw"], J.T[None])  # one "element"
    # attribute row differs by construction (per-point attr in the fixture): compare rows 1..10
    np.testing.assert_allclose(geom[0, 1:], G["geom"][1:], rtol=TOL, atol=TOL)
    gc = capi.build_geom_33(np.ones(1), G["qw"], J[None])
    np.testing.assert_allclose(gc[0, 1:], G["geom"][1:], rtol=TOL, atol=TOL)


@pytest.mark.parametrize("tag", ["a", "b", "id"])
def test_hcurl_hdiv_numpy(tag):
    ctx, _ = _ctx_from_blob(G["ctx_" + tag])
    geom = G["geom"][None]
    v = po.apply_hcurl_33(ctx, geom, G["u"][None])[0]
    np.testing.assert_allclose(v, G["hcurl_" + tag], rtol=TOL, atol=TOL)
    cv = po.apply_hdiv_33(ctx, geom, G["cu"][None])[0]
    np.testing.assert_allclose(cv, G["hdiv_" + tag], rtol=TOL, atol=TOL)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_mixed_hcurl_hdiv_numpy(tag):
    """f_apply_hcurlhdiv_33 / f_apply_hdivhcurl_33 (hcurlhdiv_33_qf.h: the weak-curl and mixed-curl integrators) against the
    vectors produced by the reference header (anisotropic SPD and non-symmetric coefficients: the latter exposes a swapped
    A / C argument order)."""
    ctx, _ = _ctx_from_blob(G["ctx_" + tag])
    geom = G["geom"][None]
    np.testing.assert_allclose(po.apply_hcurlhdiv_33(ctx, geom, G["u"][None])[0], G["hcurlhdiv_" + tag], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(po.apply_hdivhcurl_33(ctx, geom, G["cu"][None])[0], G["hdivhcurl_" + tag], rtol=TOL, atol=TOL)


def test_error_qfunctions_numpy():
    """f_apply_hcurlhdiv_error_33 / f_apply_hdivhcurl_error_33 (hcurlhdiv_error_33_qf.h, the integrands of the flux error
    estimators) against the vectors of the reference header: pair context (anisotropic first, non-symmetric second)."""
    c1, n1 = _ctx_from_blob(G["ctx_pair"])
    c2, _ = _ctx_from_blob(G["ctx_pair"][n1:])
    geom = G["geom"][None]
    np.testing.assert_allclose(po.apply_hcurlhdiv_error_33(c1, c2, geom, G["u"][None], G["cu"][None])[0], G["hcurlhdiv_error"],
                               rtol=TOL, atol=TOL)
    np.testing.assert_allclose(po.apply_hdivhcurl_error_33(c1, c2, geom, G["u"][None], G["cu"][None])[0], G["hdivhcurl_error"],
                               rtol=TOL, atol=TOL)
    assert G["hcurlhdiv_error"].min() > 0 and not np.allclose(G["hcurlhdiv_error"], G["hdivhcurl_error"])


@pytest.mark.parametrize("tag", ["a", "b", "id"])
def test_hcurl_hdiv_c(tag):
    blob = G["ctx_" + tag]
    v, _ = capi.qfunction(capi.QF_HCURL, blob, G["geom"], u=G["u"])
    np.testing.assert_allclose(v, G["hcurl_" + tag], rtol=TOL, atol=TOL)
    _, cv = capi.qfunction(capi.QF_HDIV, blob, G["geom"], cu=G["cu"])
    np.testing.assert_allclose(cv, G["hdiv_" + tag], rtol=TOL, atol=TOL)


def test_hdivmass_pair_context():
    blob = G["ctx_pair"]
    c_mass, n = _ctx_from_blob(blob)
    c_curl, _ = _ctx_from_blob(blob[n:])
    v, cv = po.apply_hdivmass_33(c_mass, c_curl, G["geom"][None], G["u"][None], G["cu"][None])
    np.testing.assert_allclose(v[0], G["hdivmass_v"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(cv[0], G["hdivmass_cv"], rtol=TOL, atol=TOL)
    v2, cv2 = capi.qfunction(capi.QF_HDIVMASS, blob, G["geom"], u=G["u"], cu=G["cu"])
    np.testing.assert_allclose(v2, G["hdivmass_v"], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(cv2, G["hdivmass_cv"], rtol=TOL, atol=TOL)


def test_context_packing_matches_product_packer():
    """palace_amd.ceed.coefficient_context restates coefficient.cpp:51-118 independently."""
    from palace_amd.ceed import coefficient_context

    rng = np.random.default_rng(3)
    m = rng.uniform(-1, 1, (3, 3))
    a = po.CoeffCtx(attr_mat=[0, 1, -1, 1], mat_coeff=[m, np.array([0.5])], a=1.7).pack()
    b = coefficient_context(3, attr_mat=[0, 1, -1, 1], mat_coeff=[m, np.array([0.5])], a=1.7)
    assert a.tobytes() == b.tobytes()
    assert po.CoeffCtx(a=2.0).pack().tobytes() == coefficient_context(3, a=2.0).tobytes()


@pytest.mark.skipif(not capi.ref_available(), reason="oracle/_ref not built (no reference tree)")
def test_live_reference_library():
    rng = np.random.default_rng(11)
    Q = 27
    J = np.eye(3).reshape(9, 1) + 0.2 * rng.uniform(-1, 1, (9, Q))
    attr = np.ones(Q)
    qw = rng.uniform(0.1, 1, Q)
    geom = np.zeros((11, Q))
    capi.ref_call("f_build_geom_factor_33", None, Q, [attr, qw, np.ascontiguousarray(J)], [geom])
    np.testing.assert_allclose(po.build_geom_factor_33(np.ones(1), qw, J.T[None])[0], geom, rtol=TOL)
    ctx = po.CoeffCtx(attr_mat=[0], mat_coeff=[rng.uniform(-1, 1, (3, 3)) + 2 * np.eye(3)])
    u = rng.uniform(-1, 1, (3, Q))
    v = np.zeros((3, Q))
    capi.ref_call("f_apply_hdiv_33", ctx.pack(), Q, [geom, u], [v])
    np.testing.assert_allclose(po.apply_hdiv_33(ctx, geom[None], u[None])[0], v, rtol=TOL, atol=TOL)
