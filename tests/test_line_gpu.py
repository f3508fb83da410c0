"""Line elements through the dense path (the `21` / `31` QFunctions -- boundaries of plane problems and
curves in space): ND mass (f_apply_hcurl_21 / _31 on the tangential value), H1 mass (f_apply_h1_1), H1 diffusion (the same
line form on du/dxi) and H1 diffusion + mass (f_apply_hcurlmass_21 / _31) on a curved polyline with quadratic geometry, against
the oracle (pinned on the reference headers, tests/test_oracle_2d.py::test_line_element_qfunctions_21_31)."""
import numpy as np
import pytest

from oracle import palace_oracle as po

pytestmark = pytest.mark.gpu


def _curve(sdim, ne=37):
    """ne quadratic segments along a smooth curve: nodes [2 ne + 1, sdim], elem_nodes [ne, 3] = (end, end, middle)."""
    t = np.linspace(0.0, 1.0, 2 * ne + 1) ** 1.3  # non-uniform
    X = np.stack([np.cos(2.1 * t) + 0.2 * t, np.sin(1.7 * t), 0.4 * t * t + 0.1 * np.sin(5 * t)][:sdim], axis=1)
    en = np.stack([2 * np.arange(ne), 2 * np.arange(ne) + 2, 2 * np.arange(ne) + 1], axis=1)
    return X, en.astype(np.int32)


def _setup(sdim, p):
    from palace_amd import ceed
    from palace_amd.fem.basis1d import gauss_legendre, gauss_lobatto, lagrange_eval

    X, en = _curve(sdim)
    ne = en.shape[0]
    attr = (1 + (np.arange(ne) % 2)).astype(np.int32)
    qx, qw = gauss_legendre(p + 2)
    _, Gm = lagrange_eval(np.array([0.0, 1.0, 0.5]), qx)  # geometry: quadratic Lagrange on (0, 1, 1/2)
    mesh_grad = Gm[None]  # [1, Q, 3]
    geom = ceed.DenseGeomFactorData(en, X, attr, mesh_grad, qw)
    J = np.einsum("qn,eni->eqi", Gm, X[en])
    ogeom = po.build_geom_factor_line(attr.astype(np.float64), qw, J)
    # H1: Lagrange on Gauss-Lobatto nodes, end nodes shared with the neighbours
    B, G = lagrange_eval(gauss_lobatto(p + 1), qx)
    h1_off = np.zeros((ne, p + 1), dtype=np.int32)
    h1_off[:, 0] = np.arange(ne)
    h1_off[:, p] = np.arange(ne) + 1
    for k in range(1, p):
        h1_off[:, k] = ne + 1 + (p - 1) * np.arange(ne) + (k - 1)
    n_h1 = ne + 1 + (p - 1) * ne
    # ND: p tangential dofs per segment (Lagrange on Gauss-Legendre nodes), every second segment reversed (sign flips)
    Bo, _ = lagrange_eval(gauss_legendre(p)[0], qx)
    nd_off = (p * np.arange(ne)[:, None] + np.arange(p)[None, :]).astype(np.int32)
    nd_ori = np.zeros((ne, p), dtype=bool)
    nd_ori[1::2] = True
    return dict(geom=geom, ogeom=ogeom, qw=qw, ne=ne, h1=(n_h1, h1_off, B[None], G[None]), nd=(p * ne, nd_off, nd_ori, Bo[None]))


def _ctxs(sdim):
    rng = np.random.default_rng(5 + sdim)
    A = rng.uniform(-1, 1, (sdim, sdim))
    cm = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[A @ A.T + 2 * np.eye(sdim), np.array([0.6])], a=1.2, dim=sdim)
    c1 = po.CoeffCtx(attr_mat=[1, 0], mat_coeff=[np.array([1.9]), np.array([0.4])], dim=1)
    return cm, c1


@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("sdim", [2, 3])
def test_line_element_forms(sdim, p):
    import torch

    from palace_amd import ceed

    S = _setup(sdim, p)
    got = S["geom"].to_numpy()
    assert got.shape == S["ogeom"].shape and np.abs(got - S["ogeom"]).max() <= 1e-13 * np.abs(S["ogeom"]).max()
    cm, c1 = _ctxs(sdim)
    q_vec = ceed.QF_HCURL_31 if sdim == 3 else ceed.QF_HCURL_21
    q_pair = ceed.QF_HCURLMASS_31 if sdim == 3 else ceed.QF_HCURLMASS_21
    n_h1, h1_off, B, G = S["h1"]
    n_nd, nd_off, nd_ori, Bo = S["nd"]
    h1_block = ceed.DenseBlock(ceed.FE_H1, n_h1, h1_off, B, G)
    nd_block = ceed.DenseBlock(ceed.FE_HCURL, n_nd, nd_off, Bo, None, orients=nd_ori)
    rng = np.random.default_rng(p)
    cases = [
        ("nd mass", nd_block, n_nd, q_vec, cm.pack(), ceed.EVAL_INTERP,
         po.CeedOperatorOracle(n_nd, nd_off, nd_ori, Bo, Bo, S["ogeom"], po.QF_HCURL_LINE, cm)),
        ("h1 mass", h1_block, n_h1, ceed.QF_H1_1, c1.pack(), ceed.EVAL_INTERP,
         po.CeedOperatorOracle(n_h1, h1_off, None, B, G, S["ogeom"], po.QF_H1MASS, c1, vector_fe=False)),
        ("h1 diffusion", h1_block, n_h1, q_vec, cm.pack(), ceed.EVAL_GRAD,
         po.CeedOperatorOracle(n_h1, h1_off, None, B, G, S["ogeom"], po.QF_HCURL_LINE, cm, vector_fe=False)),
        ("h1 diffusion + mass", h1_block, n_h1, q_pair, np.concatenate([c1.pack(), cm.pack()]), ceed.EVAL_GRAD | ceed.EVAL_INTERP,
         po.CeedOperatorOracle(n_h1, h1_off, None, B, G, S["ogeom"], po.QF_HCURLMASS_LINE, c1, cm, vector_fe=False)),
    ]
    for name, block, n, qf, blob, ops, orc in cases:
        op = ceed.Operator(n, n).add_dense_integrator(S["geom"], block, qf, blob, ops).finalize()
        x = rng.uniform(-1, 1, n)
        ref = orc.apply_add(x, np.zeros(n))
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        op.mult(torch.from_numpy(x).cuda(), y)
        assert np.abs(y.cpu().numpy() - ref).max() < 1e-12 * np.abs(ref).max(), name
        d = torch.empty_like(y)
        op.assemble_diagonal(d)
        dref = orc.diagonal()
        assert np.abs(d.cpu().numpy() - dref).max() < 1e-12 * np.abs(dref).max(), name
    # the length of the curve: 1^T M 1 with the H1 mass operator and unit coefficient
    ident = po.CoeffCtx(dim=1)
    M = ceed.Operator(n_h1, n_h1).add_dense_integrator(S["geom"], h1_block, ceed.QF_H1_1, ident.pack(), ceed.EVAL_INTERP).finalize()
    ones = torch.ones(n_h1, dtype=torch.float64, device="cuda")
    y = torch.empty_like(ones)
    M.mult(ones, y)
    assert abs(float(y.sum()) - S["ogeom"][:, 1, :].sum()) < 1e-12 * S["ogeom"][:, 1, :].sum()
