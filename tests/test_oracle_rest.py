"""The members of the `32 | 31 | 21` QFunction families no integrator of the hot path's callers uses, and the div-div + mass and
gradient forms (fem/qfunctions/{32,31,21}/hdiv_*, hcurlhdiv_*, fem/qfunctions/*/l2mass_*,
hcurlh1d_*): the oracle's restatements against vectors the reference's own headers produced (tests/golden/make_golden.py::
fixtures_rest, oracle/ref_shim.cpp), and -- when the reference tree is here -- against the compiled headers directly."""
import os

import numpy as np
import pytest

from oracle import palace_oracle as po

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "qf_rest_golden.npz"))
TOL = dict(rtol=1e-12, atol=1e-13)


def _ctx(blob, dim):
    iv = np.asarray(blob).view(np.int32).reshape(-1, 2)[:, 0]
    nattr = int(iv[0])
    nmat = int(iv[1 + nattr])
    c = po.CoeffCtx(dim=dim)
    c.attr_mat = iv[1 : 1 + nattr].astype(np.int32)
    c.mat = np.asarray(blob)[2 + nattr : 2 + nattr + nmat * dim * dim].reshape(nmat, dim * dim)
    return c


def _cm(tag):
    return _ctx(G["ctx3"], 3) if tag // 10 == 3 else _ctx(G["ctx2"], 2)


@pytest.mark.parametrize("tag", [32, 31, 21])
def test_contravariant_members_on_boundary_and_line_elements(tag):
    """f_apply_hdiv_*, f_apply_hcurlhdiv_*, f_apply_hdivhcurl_* (non-symmetric coefficient: the two mixed members differ)."""
    geom, u = G["geom%d" % tag][None], G["u%d" % tag][None]
    fs = ((po.apply_hdiv_32, po.apply_hcurlhdiv_32, po.apply_hdivhcurl_32) if tag == 32 else
          (po.apply_hdiv_line, po.apply_hcurlhdiv_line, po.apply_hdivhcurl_line))
    for f, name in zip(fs, ("hdiv", "hcurlhdiv", "hdivhcurl")):
        np.testing.assert_allclose(f(_cm(tag), geom, u)[0], G["%s_%d" % (name, tag)], err_msg=name, **TOL)
    if tag == 32:  # (on a line the two maps are parallel vectors: t^T C a = a^T C t for any C)
        assert not np.allclose(G["hcurlhdiv_%d" % tag], G["hdivhcurl_%d" % tag])
    assert not np.allclose(G["hdiv_%d" % tag], G["hcurlhdiv_%d" % tag])


@pytest.mark.parametrize("tag", [33, 22, 32, 31, 21])
def test_divdiv_mass_and_gradient_forms(tag):
    """f_apply_l2mass_* (pair context: space_dim x space_dim mass coefficient, then the scalar one) and f_apply_hcurlh1d_*."""
    geom, u, du = G["geom%d" % tag][None], G["u%d" % tag][None], G["du%d" % tag][None]
    v, dv = po.apply_l2mass(_cm(tag), _ctx(G["ctx1"], 1), geom, G["qw"], u, du)
    np.testing.assert_allclose(v[0], G["l2mass_%d_v" % tag], **TOL)
    np.testing.assert_allclose(dv[0], G["l2mass_%d_dv" % tag], **TOL)
    gv = po.apply_hcurlh1d(_cm(tag), geom, u)
    assert gv.shape[1] == tag // 10
    np.testing.assert_allclose(gv[0], G["hcurlh1d_%d" % tag], **TOL)


@pytest.mark.parametrize("n", [2, 3])
def test_vector_valued_scalar_spaces(n):
    """f_apply_h1_2 | _3 (MassIntegrator on a vector H1 space) and f_apply_l2_2 | _3 (non-symmetric N x N coefficient)."""
    tag = 11 * n
    geom, uv = G["geom%d" % tag][None], G["uv%d" % n][None]
    np.testing.assert_allclose(po.apply_h1_vec(_cm(tag), geom, uv)[0], G["h1_%d" % n], **TOL)
    np.testing.assert_allclose(po.apply_l2_vec(_cm(tag), geom, G["qw"], uv)[0], G["l2_%d" % n], **TOL)


def test_live_reference_headers_when_present():
    """The same functions against oracle/_ref (the reference's headers compiled in place) on fresh random draws."""
    from oracle import capi

    if not os.path.isdir("/root/reference/palace/fem/qfunctions"):
        pytest.skip("no reference tree on this machine")
    capi.build(ref=True)
    rng = np.random.default_rng(7)
    Q = 16
    attr, qw = rng.integers(1, 3, Q).astype(np.float64), rng.uniform(0.01, 0.2, Q)
    for tag in (32, 31, 21):
        sdim, dim = tag // 10, tag % 10
        J = rng.uniform(-1, 1, (sdim * dim, Q)) + np.array([1, 0, 0, 0, 1, 0.3, 0, 0, 1][: sdim * dim]).reshape(-1, 1)
        g = np.zeros((2 + sdim * dim, Q))
        capi.ref_call("f_build_geom_factor_%d" % tag, None, Q, [attr, qw, np.ascontiguousarray(J)], [g])
        u = rng.uniform(-1, 1, (dim, Q))
        v = np.zeros((dim, Q))
        capi.ref_call("f_apply_hdiv_%d" % tag, _cm(tag).pack(), Q, [g, u], [v])
        f = po.apply_hdiv_32 if tag == 32 else po.apply_hdiv_line
        np.testing.assert_allclose(f(_cm(tag), g[None], u[None])[0], v, **TOL)
