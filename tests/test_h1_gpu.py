"""H1 hexahedra: diffusion / mass / diffusion+mass apply, diagonal, p-coarsening and p-prolongation
against the oracle (reference: fem/integ/{diffusion,mass,diffusionmass}.cpp, qfunctions hcurl_33 on
grad u, h1_1, hcurlmass_33; criterion test/unit/test-libceed.cpp:245-282)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import palace_oracle as po  # noqa: E402
from palace_amd import ceed, linalg  # noqa: E402
from palace_amd.fem.fespace import H1HexSpace  # noqa: E402
from tests import util  # noqa: E402

RTOL = 1e-12


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _new(n):
    return torch.zeros(n, dtype=torch.float64, device="cuda")


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _multi_attr(mesh):
    return type(mesh)(x=mesh.x, elem_nodes=mesh.elem_nodes, attr=(np.arange(mesh.ne) % 3 + 1).astype(np.int32))


def _oracle(h1, geom, qf, ctx, ctx2, q1d):
    interp, grad = po.h1_hex_dense_tables(h1.p, q1d)
    return po.CeedOperatorOracle(h1.ndofs, h1.elem_dof_lex, None, interp, grad, geom, qf, ctx, ctx2, vector_fe=False)


def _ctxs():
    rng = np.random.default_rng(7)
    A = rng.uniform(-1, 1, (3, 3))
    spd = A @ A.T + 2.0 * np.eye(3)
    c_diff = po.CoeffCtx(attr_mat=[0, 1, 0], mat_coeff=[spd, np.array([0.7])], a=1.3)
    c_mass = po.CoeffCtx(attr_mat=[1, 0, 1], mat_coeff=[np.array([2.08]), np.array([0.5])], dim=1)
    return c_mass, c_diff


@pytest.mark.parametrize("p,q1d", [(1, 2), (2, 3), (3, 4), (1, 4), (2, 4), (4, 5)])
@pytest.mark.parametrize("qf", ["diffusion", "mass", "diffusionmass"])
@pytest.mark.parametrize("dstage", ["qdata", "matrix_free"])
def test_h1_apply_and_diagonal(cylinder_mesh, monkeypatch, p, q1d, qf, dstage):
    if dstage == "matrix_free":  # D from the geometry factors, as the reference QFunctions compute it
        monkeypatch.setenv("PALACE_AMD_QDATA", "0")
    mesh = _multi_attr(cylinder_mesh)
    h1 = H1HexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, q1d)
    ogeom = util.oracle_geom(mesh, q1d)
    c_mass, c_diff = _ctxs()
    dense = po.h1_hex_dense_tables(p, q1d)
    if qf == "diffusion":
        op = ceed.diffusion_operator(geom, h1, c_diff.pack(), dense)
        o = _oracle(h1, ogeom, po.QF_HCURL, c_diff, None, q1d)
    elif qf == "mass":
        op = ceed.h1mass_operator(geom, h1, c_mass.pack(), dense)
        o = _oracle(h1, ogeom, po.QF_H1MASS, c_mass, None, q1d)
    else:
        op = ceed.diffusionmass_operator(geom, h1, c_mass.pack(), c_diff.pack(), dense)
        o = _oracle(h1, ogeom, po.QF_HCURLMASS, c_mass, c_diff, q1d)
    x = np.random.default_rng(1).uniform(-1, 1, h1.ndofs)
    y = op.mult(_dev(x), _new(h1.ndofs)).cpu().numpy()
    ref = o.apply_add(x, np.zeros(h1.ndofs))
    assert _rel(y, ref) < RTOL
    d = op.assemble_diagonal(_new(h1.ndofs)).cpu().numpy()
    assert _rel(d, o.diagonal()) < RTOL


def test_h1_constant_in_diffusion_nullspace_and_mass_volume(cylinder_mesh):
    """Basis-invariant checks: K 1 = 0 and 1^T M 1 = volume of the (Q2) cylinder mesh."""
    mesh = cylinder_mesh
    h1 = H1HexSpace(mesh, 3)
    geom = ceed.GeomFactorData(mesh, 4)
    K = ceed.diffusion_operator(geom, h1, ceed.coefficient_context(3))
    M = ceed.h1mass_operator(geom, h1, ceed.coefficient_context(1))
    one = torch.ones(h1.ndofs, dtype=torch.float64, device="cuda")
    k1 = K.mult(one, _new(h1.ndofs)).cpu().numpy()
    m1 = M.mult(one, _new(h1.ndofs)).cpu().numpy()
    assert np.abs(k1).max() < 1e-12
    vol = util.oracle_geom(mesh, 4)[:, 1, :].sum()
    assert abs(m1.sum() - vol) < 1e-12 * vol


def test_h1_prolongation(cylinder_mesh):
    mesh = cylinder_mesh
    ctx = linalg.Context()
    for pc, pf in ((1, 2), (2, 3), (1, 3)):
        hc, hf = H1HexSpace(mesh, pc), H1HexSpace(mesh, pf)
        P = linalg.Interp(ctx, hc, hf)
        # a trilinear-per-element-exact test: interpolate a smooth function given at the coarse nodes
        xc = np.random.default_rng(3).uniform(-1, 1, hc.ndofs)
        yf = P.mult(_dev(xc), _new(hf.ndofs)).cpu().numpy()
        # oracle: dense element interpolation in tensor order
        cpc, cpf = po.gll_points(pc + 1), po.gll_points(pf + 1)
        I1 = np.array([[po.lagrange(cpc, xf, a)[0] for a in range(pc + 1)] for xf in cpf])
        M = np.einsum("kc,jb,ia->kjicba", I1, I1, I1).reshape((pf + 1) ** 3, (pc + 1) ** 3)
        ones_c = np.ones(hc.elem_dof_lex.shape, dtype=np.int8)
        ones_f = np.ones(hf.elem_dof_lex.shape, dtype=np.int8)
        o = po.InterpOracle(hc.elem_dof_lex, ones_c, hf.elem_dof_lex, ones_f, hc.ndofs, hf.ndofs, M)
        assert _rel(yf, o.mult(xc)) < 1e-13
        xf = np.random.default_rng(4).uniform(-1, 1, hf.ndofs)
        yc = P.mult_transpose(_dev(xf), _new(hc.ndofs)).cpu().numpy()
        assert _rel(yc, o.mult_transpose(xf)) < 1e-13


def test_h1_streaming_kernel_all_orders():
    """The streaming H1 kernel (pa_h1_hex_stream.hip) is on by default at p = 3 only; PALACE_AMD_STREAM_H1=all switches it on
    for every order (the switch is read once per process, hence the subprocess): diffusion, mass and diffusion + mass at
    p = 1, 2, 3 with four quadrature points per direction, with and without essential dofs, against the oracle."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch
from oracle import palace_oracle as po
from palace_amd import ceed, linalg
from palace_amd.fem.fespace import H1HexSpace
from palace_amd.fem.mesh import HexMesh
from tests import util
from tests.test_h1_gpu import _multi_attr, _oracle, _ctxs, _dev, _new, _rel
d = np.load(%r)
mesh = _multi_attr(HexMesh(x=d["x"], elem_nodes=d["elem_nodes"].astype(np.int64), attr=d["attr"], bdr_faces=d["bdr_faces"],
                           bdr_attr=d["bdr_attr"]))
q1d = 4
geom = ceed.GeomFactorData(mesh, q1d)
ogeom = util.oracle_geom(mesh, q1d)
c_mass, c_diff = _ctxs()
ctx = linalg.Context()
for p in (1, 2, 3):
    h1 = H1HexSpace(mesh, p)
    for qf in ("diffusion", "mass", "diffusionmass"):
        if qf == "diffusion":
            op, o = ceed.diffusion_operator(geom, h1, c_diff.pack()), _oracle(h1, ogeom, po.QF_HCURL, c_diff, None, q1d)
        elif qf == "mass":
            op, o = ceed.h1mass_operator(geom, h1, c_mass.pack()), _oracle(h1, ogeom, po.QF_H1MASS, c_mass, None, q1d)
        else:
            op, o = (ceed.diffusionmass_operator(geom, h1, c_mass.pack(), c_diff.pack()),
                     _oracle(h1, ogeom, po.QF_HCURLMASS, c_mass, c_diff, q1d))
        x = np.random.default_rng(p).uniform(-1, 1, h1.ndofs)
        y = op.mult(_dev(x), _new(h1.ndofs)).cpu().numpy()
        ref = o.apply_add(x, np.zeros(h1.ndofs))
        assert _rel(y, ref) < 1e-12, (p, qf, _rel(y, ref))
        ess = h1.ess_dofs()
        A = linalg.ParOperator(ctx, op, ess, linalg.DIAG_ONE)
        ya = A.mult(_dev(x), _new(h1.ndofs)).cpu().numpy()
        tx = x.copy(); tx[ess] = 0.0
        rb = o.apply_add(tx, np.zeros(h1.ndofs)); rb[ess] = x[ess]
        assert _rel(ya, rb) < 1e-12, (p, qf, "bc", _rel(ya, rb))
print("OK")
''' % (root, os.path.join(root, "tests", "golden", "cylinder_hex_mesh.npz"))
    env = dict(os.environ, PALACE_AMD_STREAM_H1="all")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("qf", ["diffusion", "diffusionmass"])
def test_h1_chebyshev_steps_fused_into_the_gather(cylinder_mesh, monkeypatch, qf):
    """Round 6: the smoother step in the epilogue of the E^T gather on the H1 streaming kernel (order 3: the auxiliary-space
    operators of the Hiptmair smoother): against the same smoother with the step as a vector kernel (PALACE_AMD_FUSED_STEP=0), zero
    and non-zero initial guess, and against the oracle's recurrence."""
    mesh = _multi_attr(cylinder_mesh)
    p, q1d = 3, 4
    h1 = H1HexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, q1d)
    ogeom = util.oracle_geom(mesh, q1d)
    c_mass, c_diff = _ctxs()
    if qf == "diffusion":
        op, o = ceed.diffusion_operator(geom, h1, c_diff.pack()), _oracle(h1, ogeom, po.QF_HCURL, c_diff, None, q1d)
    else:
        op, o = (ceed.diffusionmass_operator(geom, h1, c_mass.pack(), c_diff.pack()),
                 _oracle(h1, ogeom, po.QF_HCURLMASS, c_mass, c_diff, q1d))
    ess = h1.ess_dofs()
    ctx = linalg.Context()
    A = linalg.ParOperator(ctx, op, ess, linalg.DIAG_ONE)
    S = linalg.chebyshev(ctx, A, order=4)
    assert S.fused_step()
    monkeypatch.setenv("PALACE_AMD_FUSED_STEP", "0")
    S0 = linalg.chebyshev(ctx, A, order=4)
    assert not S0.fused_step() and S0.lambda_max() == S.lambda_max()
    n = h1.ndofs
    rng = np.random.default_rng(31)
    b, g = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    b[ess] = 0.0
    g[ess] = 0.0
    y = S.mult(_dev(b), _new(n)).cpu().numpy()
    y0 = S0.mult(_dev(b), _new(n)).cpu().numpy()
    z = S.mult(_dev(b), _dev(g.copy()), initial_guess=True).cpu().numpy()
    z0 = S0.mult(_dev(b), _dev(g.copy()), initial_guess=True).cpu().numpy()
    assert _rel(y, y0) < 1e-13 and _rel(z, z0) < 1e-13

    class _Par:  # the oracle's ParOperator semantics around its local operator (rap.cpp:195-234), diagonal from the device
        def __init__(self):
            self.n = n
            self._d = A.assemble_diagonal(_new(n)).cpu().numpy()

        def mult(self, v):
            t = v.copy()
            t[ess] = 0.0
            w = o.apply_add(t, np.zeros(n))
            w[ess] = v[ess]
            return w

        def diagonal(self):
            return self._d

    ch = po.ChebyshevOracle(_Par(), 4, lambda_max=S.lambda_max())
    assert _rel(y, ch.mult2(b, None, False)) < 1e-11 and _rel(z, ch.mult2(b, g.copy(), True)) < 1e-11
