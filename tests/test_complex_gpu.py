"""Complex operators and complex GMRES (reference: linalg/operator.cpp:58-134 ComplexWrapperOperator,
linalg/iterative.cpp:543-705 GmresSolver<ComplexOperator>, real preconditioner on both parts):
a lossy cavity system A = K - w^2 eps (1 - i tan d) M at 2 GHz, below the first resonance, solved with
GMRES + Hiptmair p-multigrid built on the shifted real matrix K + w^2 eps M, against a sparse direct
solve of the oracle's assembled complex matrix."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import scipy.sparse.linalg as spla  # noqa: E402

from oracle import palace_oracle as po  # noqa: E402
from palace_amd import ceed, linalg  # noqa: E402
from palace_amd.fem.fespace import H1HexSpace, NDHexSpace  # noqa: E402
from tests import util  # noqa: E402


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_complex_wrapper_and_gmres(cylinder_mesh):
    mesh, orders, q1d = cylinder_mesh, [1, 2], 3
    w2eps, tand = (2 * np.pi * 2.0e9 * 1e-2 / 299792458.0) ** 2 * 2.08, 0.1
    ctx = linalg.Context()
    nds = [NDHexSpace(mesh, p) for p in orders]
    h1s = [H1HexSpace(mesh, p) for p in orders]
    nd, n = nds[-1], nds[-1].ndofs
    ess = nd.ess_dofs()
    geom = ceed.GeomFactorData(mesh, q1d)
    ident = ceed.coefficient_context(3)
    coef = lambda v: ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([v])])  # noqa: E731
    Ar = linalg.ParOperator(ctx, ceed.curlcurlmass_operator(geom, nd, coef(-w2eps), ident), ess, linalg.DIAG_ONE)
    Ai = linalg.ParOperator(ctx, ceed.ndmass_operator(geom, nd, coef(w2eps * tand)), ess, linalg.DIAG_ZERO)
    # oracle matrix
    ogeom = util.oracle_geom(mesh, q1d)
    off, ori = nd.native_restriction()
    interp, curl = po.nd_hex_dense_tables(2, q1d, nd.dof_map_native())
    Ko = po.CeedOperatorOracle(n, off, ori, interp, curl, ogeom, po.QF_HDIV, po.CoeffCtx()).assemble_sparse()
    Mo = po.CeedOperatorOracle(n, off, ori, interp, curl, ogeom, po.QF_HCURL, po.CoeffCtx()).assemble_sparse()
    Ao = (Ko - w2eps * (1 - 1j * tand) * Mo).tolil()
    for d in ess:  # ParOperator elimination: real part DIAG_ONE, imaginary part DIAG_ZERO
        Ao[d, :] = 0
        Ao[:, d] = 0
        Ao[d, d] = 1.0
    Ao = Ao.tocsc()
    rng = np.random.default_rng(12)
    x = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    yr, yi = linalg.complex_mult(ctx, Ar, Ai, _dev(x.real), _dev(x.imag), torch.empty(n, dtype=torch.float64, device="cuda"),
                                 torch.empty(n, dtype=torch.float64, device="cuda"))
    y = yr.cpu().numpy() + 1j * yi.cpu().numpy()
    assert np.linalg.norm(y - Ao @ x) < 1e-12 * np.linalg.norm(y)
    # preconditioner: Hiptmair p-multigrid on the shifted SPD matrix K + w^2 eps M
    pfine = ceed.curlcurlmass_operator(geom, nd, coef(w2eps), ident)
    ploc = [pfine.coarsen(geom, nds[0]), pfine]
    Pm = [linalg.ParOperator(ctx, o, s.ess_dofs()) for o, s in zip(ploc, nds)]
    hfine = ceed.diffusion_operator(geom, h1s[-1], coef(w2eps))
    hloc = [hfine.coarsen(geom, h1s[0]), hfine]
    Ph = [linalg.ParOperator(ctx, o, s.ess_dofs()) for o, s in zip(hloc, h1s)]
    G = [linalg.Gradient(ctx, h, s) for h, s in zip(h1s, nds)]
    coarse = linalg.cg(ctx, Pm[0], linalg.jacobi(ctx, Pm[0]), rel_tol=1e-13, max_it=2000)  # ~exact: keeps B linear
    B = linalg.gmg(ctx, Pm, [linalg.Interp(ctx, nds[0], nds[1])], coarse, cheby_order=4, A_aux=Ph, G=G)
    b = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    b[ess] = 0.0
    S = linalg.ComplexGmres(ctx, Ar, Ai, B, rel_tol=1e-10, max_it=300, restart=100)
    xr, xi = S.mult(_dev(b.real), _dev(b.imag), torch.zeros(n, dtype=torch.float64, device="cuda"),
                    torch.zeros(n, dtype=torch.float64, device="cuda"))
    st = S.stats()
    assert st["converged"], st
    xs = xr.cpu().numpy() + 1j * xi.cpu().numpy()
    ref = spla.spsolve(Ao, b)
    assert np.linalg.norm(Ao @ xs - b) < 1e-7 * np.linalg.norm(b)
    assert np.linalg.norm(xs - ref) < 1e-6 * np.linalg.norm(ref)


FUSED_CHECK = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, %r)
from palace_amd import ceed, linalg
from palace_amd.fem.fespace import NDHexSpace
from palace_amd.fem.mesh import ogrid_cylinder
p, q1d = int(sys.argv[1]), int(sys.argv[3])
ctx = linalg.Context()
shape = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (2, 3)
mesh = ogrid_cylinder(*shape)
mesh.attr[:] = 1 + (np.arange(mesh.ne) %% 2)
nd = NDHexSpace(mesh, p)
geom = ceed.GeomFactorData(mesh, q1d)
mat = sys.argv[4] if len(sys.argv) > 4 else "iso"
two = lambda a, b: ceed.coefficient_context(3, attr_mat=[0, 1], mat_coeff=[np.asarray(a, float), np.asarray(b, float)])
if mat == "iso":
    Ar = ceed.curlcurlmass_operator(geom, nd, two(-0.9, -0.35), two(1.0, 0.6))
    Ai = ceed.ndmass_operator(geom, nd, two(0.21, 0.05))
else:  # symmetric material tensors (tests/test_complex_gpu.py: _tensors): packed D of each operator, the complex kernel on them
    T = np.load(sys.argv[5])
    Ar = ceed.curlcurlmass_operator(geom, nd, two(T["mr0"], T["mr1"]), two(T["cr0"], T["cr1"]))
    Ai = (ceed.ndmass_operator(geom, nd, two(T["mi0"], T["mi1"])) if mat == "aniso" else
          ceed.curlcurlmass_operator(geom, nd, two(T["mi0"], T["mi1"]), two(T["ci0"], T["ci1"])))
ess = nd.ess_dofs()
n = nd.ndofs
rng = np.random.default_rng(3)
x = [torch.from_numpy(rng.uniform(-1, 1, n)).cuda() for _ in range(2)]
out = {}
for tag, e in (("plain", np.zeros(0, np.int32)), ("ess", ess)):
    A = linalg.ComplexParOperator(ctx, Ar, Ai, e, linalg.DIAG_ONE)
    yr, yi = torch.empty_like(x[0]), torch.empty_like(x[0])
    A.mult(x[0], x[1], yr, yi)
    out[tag] = (yr.cpu().numpy(), yi.cpu().numpy())
    lr, li = torch.empty_like(x[0]), torch.empty_like(x[0])
    A.mult(x[0], x[1], lr, li, local=True)  # the L-vector ComplexWrapperOperator (no essential dofs)
    out[tag + "_local"] = (lr.cpu().numpy(), li.cpu().numpy())
np.savez(sys.argv[2], fused=ceed._lib.load().pa_op_complex_fused(Ar.handle, Ai.handle),
         **{k + "_" + c: v[i] for k, v in out.items() for i, c in enumerate("ri")}, xr=x[0].cpu().numpy(), xi=x[1].cpu().numpy())
print("OK")
'''


def _tensors():
    """Two materials' symmetric tensors: mass (indefinite real part as in K - w^2 eps M, a loss tensor for the imaginary part) and
    curl-curl (inverse permeability; a second one for an imaginary curl-curl term)."""
    rng = np.random.default_rng(21)

    def spd(scale):
        a = rng.uniform(-1, 1, (3, 3))
        return scale * (a @ a.T + 1.5 * np.eye(3))

    return dict(mr0=-spd(0.4), mr1=-spd(0.2), cr0=spd(0.5), cr1=np.diag([0.6, 0.6, 0.9]), mi0=spd(0.1), mi1=np.diag([0.02, 0.02, 0.05]),
                ci0=spd(0.07), ci1=spd(0.03))


@pytest.mark.parametrize("p,q1d,mat", [(1, 4, "iso"), (2, 4, "iso"), (3, 4, "iso"), (4, 5, "iso"), (2, 5, "iso"), (1, 5, "iso"),
                                       (1, 4, "aniso"), (2, 4, "aniso"), (3, 4, "aniso"), (3, 4, "aniso2"), (2, 4, "aniso2"),
                                       (3, 4, "iso-odd"), (2, 4, "aniso-odd"), (3, 4, "aniso2-odd")])
def test_fused_complex_apply(p, q1d, mat, tmp_path):
    """y = (A_r + i A_i) x in one pass over the element data (pa_op_mult_complex): the complex streaming
    kernel (four points per direction: pa_nd_hex_stream.hip, five: pa_nd_hex_stream5.hip -- order 4 and the coarsened levels of an
    order-4 problem) against the four separate applies (PALACE_AMD_COMPLEX_FUSED=0, the path of linalg/operator.cpp:98-134) and against the
    oracle's operators, with two materials, plain and with essential dofs (DIAG_ONE on the real, DIAG_ZERO on the imaginary part)."""
    import os
    import subprocess
    import sys

    from palace_amd.fem.mesh import ogrid_cylinder

    # "-odd": 15 elements -- the complex kernel's batches are two elements, the last one is half empty (ragged input)
    shape = (1, 3) if mat.endswith("-odd") else (2, 3)
    mat = mat.replace("-odd", "")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for fused in (1, 0):
        f = str(tmp_path / f"out{fused}.npz")
        tf = str(tmp_path / "tensors.npz")
        np.savez(tf, **_tensors())
        r = subprocess.run([sys.executable, "-c", FUSED_CHECK % root, str(p), f, str(q1d), mat, tf, str(shape[0]), str(shape[1])], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, PALACE_AMD_COMPLEX_FUSED=str(fused)))
        assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
        res[fused] = np.load(f)
    assert int(res[1]["fused"]) == 1 and int(res[0]["fused"]) == 0
    for k in ("plain_r", "plain_i", "ess_r", "ess_i", "plain_local_r", "plain_local_i"):
        a, b = res[1][k], res[0][k]
        assert np.abs(a - b).max() < 1e-13 * np.abs(b).max(), k
    # oracle
    mesh = ogrid_cylinder(*shape)
    assert mesh.ne % 2 == (1 if shape == (1, 3) else 0)
    mesh.attr[:] = 1 + (np.arange(mesh.ne) % 2)
    nd = NDHexSpace(mesh, p)
    ogeom = util.oracle_geom(mesh, q1d)
    off, ori = nd.native_restriction()
    interp, curl = util.dense_tables(nd, q1d)
    two = lambda a, b: po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[np.asarray(a, float), np.asarray(b, float)])  # noqa: E731
    T = _tensors()
    if mat == "iso":
        Aro = po.CeedOperatorOracle(nd.ndofs, off, ori, interp, curl, ogeom, po.QF_HDIVMASS, two(-0.9, -0.35), two(1.0, 0.6))
        Aio = po.CeedOperatorOracle(nd.ndofs, off, ori, interp, curl, ogeom, po.QF_HCURL, two(0.21, 0.05))
    else:
        Aro = po.CeedOperatorOracle(nd.ndofs, off, ori, interp, curl, ogeom, po.QF_HDIVMASS, two(T["mr0"], T["mr1"]),
                                    two(T["cr0"], T["cr1"]))
        Aio = (po.CeedOperatorOracle(nd.ndofs, off, ori, interp, curl, ogeom, po.QF_HCURL, two(T["mi0"], T["mi1"])) if mat == "aniso" else
               po.CeedOperatorOracle(nd.ndofs, off, ori, interp, curl, ogeom, po.QF_HDIVMASS, two(T["mi0"], T["mi1"]),
                                     two(T["ci0"], T["ci1"])))
    xr, xi = res[1]["xr"], res[1]["xi"]
    z = lambda: np.zeros(nd.ndofs)  # noqa: E731
    yr = Aro.apply_add(xr, z()) - Aio.apply_add(xi, z())
    yi = Aio.apply_add(xr, z()) + Aro.apply_add(xi, z())
    assert np.abs(res[1]["plain_r"] - yr).max() < 1e-12 * np.abs(yr).max()
    assert np.abs(res[1]["plain_i"] - yi).max() < 1e-12 * np.abs(yi).max()
    ess = nd.ess_dofs()
    txr, txi = xr.copy(), xi.copy()
    txr[ess] = 0.0
    txi[ess] = 0.0
    er = Aro.apply_add(txr, z()) - Aio.apply_add(txi, z())
    ei = Aio.apply_add(txr, z()) + Aro.apply_add(txi, z())
    er[ess], ei[ess] = xr[ess], xi[ess]
    assert np.abs(res[1]["ess_r"] - er).max() < 1e-12 * np.abs(er).max()
    assert np.abs(res[1]["ess_i"] - ei).max() < 1e-12 * np.abs(ei).max()


FUSED_TET_CHECK = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, %r)
from palace_amd import ceed, linalg
from palace_amd.fem import tet
from tests import util
p, kind, imode = int(sys.argv[1]), sys.argv[2], sys.argv[3]
ctx = linalg.Context()
mesh = tet.cube_tet_mesh(3)
mesh.attr[:] = 1 + (np.arange(mesh.ne) %% 2)
if kind == "tet10":
    warp = lambda X: np.stack([X[:, 0] + 0.04 * np.sin(2 * X[:, 1] + X[:, 2]), X[:, 1] + 0.05 * X[:, 0] * X[:, 2],
                               X[:, 2] - 0.03 * np.cos(3 * X[:, 0]) * X[:, 1]], axis=1)
    m2 = tet.to_quadratic(mesh, warp); m2.attr[:] = mesh.attr; mesh = m2
if kind == "mixed":  # curved in one half only (a mesh with a curved boundary): affine blocks and curved blocks, one launch each
    mesh = tet.cube_tet_mesh(5)
    mesh.attr[:] = 1 + (np.arange(mesh.ne) %% 2)
    def half_warp(X):
        w = np.clip(X[:, 0] - 0.45, 0.0, None) ** 2
        return X + np.stack([0.3 * w * np.sin(3 * X[:, 1]), 0.4 * w * X[:, 2], -0.35 * w * np.cos(2 * X[:, 1])], axis=1)
    m2 = tet.to_quadratic(mesh, half_warp); m2.attr[:] = mesh.attr; mesh = m2
nd = tet.NDTetSpace(mesh, p)
pts, wts = tet.default_tet_rule(p)
interp, curl = nd.elem.tables(pts)
geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, **kw)
_, b3 = util.make_ctx("aniso", 2)
_, bm = util.make_ctx("scalar", 2)
_, bi = util.make_ctx("aniso", 2)
n = nd.ndofs
Ar = ceed.Operator(n, n).add_dense_integrator(geom, block, ceed.QF_HDIVMASS_33, np.concatenate([bm, b3]), ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize()
if imode == "mass":
    Ai = ceed.Operator(n, n).add_dense_integrator(geom, block, ceed.QF_HCURL_33, bi, ceed.EVAL_INTERP).finalize()
elif imode == "curl":
    Ai = ceed.Operator(n, n).add_dense_integrator(geom, block, ceed.QF_HDIV_33, bi, ceed.EVAL_CURL).finalize()
else:
    Ai = ceed.Operator(n, n).add_dense_integrator(geom, block, ceed.QF_HDIVMASS_33, np.concatenate([bi, bm]), ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize()
ess = nd.ess_dofs()
rng = np.random.default_rng(3)
x = [torch.from_numpy(rng.uniform(-1, 1, n)).cuda() for _ in range(2)]
out = {}
for tag, e in (("plain", np.zeros(0, np.int32)), ("ess", ess)):
    A = linalg.ComplexParOperator(ctx, Ar, Ai, e, linalg.DIAG_ONE)
    yr, yi = torch.empty_like(x[0]), torch.empty_like(x[0])
    A.mult(x[0], x[1], yr, yi)
    out[tag] = (yr.cpu().numpy(), yi.cpu().numpy())
np.savez(sys.argv[4], fused=ceed._lib.load().pa_op_complex_fused(Ar.handle, Ai.handle), affine=Ar.dense_affine(),
         **{k + "_" + c: v[i] for k, v in out.items() for i, c in enumerate("ri")})
print("OK")
'''


@pytest.mark.parametrize("p,imode,kind", [(p, m, "tet4") for p in (1, 2, 3) for m in ("mass", "curl", "both")] +
                         [(1, "both", "tet10"), (2, "mass", "tet10"), (2, "curl", "tet10"), (2, "both", "tet10"), (3, "both", "tet10")] +
                         [(2, "both", "mixed"), (3, "mass", "mixed"), (3, "both", "mixed")])  # (round 5: affine and curved blocks in one mesh)
def test_fused_complex_apply_tets(p, imode, kind, tmp_path):
    """The dense-table form of the one-pass complex apply (straight-sided tetrahedra: the D of one point per element; curved
    ones, round 4: the D of both operators at every point; anisotropic materials, curl-oriented restriction for p >= 2):
    8 elements x {real, imaginary} part in the 16 columns of the matrix-core products, against the four separate applies;
    essential dofs go through ComplexParOperator's copy / mask / fix path around the fused local apply."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for fused in (1, 0):
        f = str(tmp_path / f"out{fused}.npz")
        r = subprocess.run([sys.executable, "-c", FUSED_TET_CHECK % root, str(p), kind, imode, f], capture_output=True,
                           text=True, timeout=300, env=dict(os.environ, PALACE_AMD_COMPLEX_FUSED=str(fused)))
        assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
        res[fused] = np.load(f)
    assert int(res[1]["fused"]) == 2 and int(res[0]["fused"]) == 0
    for k in ("plain_r", "plain_i", "ess_r", "ess_i"):
        a, b = res[1][k], res[0][k]
        assert np.abs(a - b).max() < 1e-13 * np.abs(b).max(), k


def test_complex_pcg_on_a_hermitian_positive_definite_system(cylinder_mesh):
    """CgSolver<ComplexOperator> (linalg/iterative.cpp:360-486): complex right-hand side and iterate, the real positive
    definite K + M as the operator (a ComplexParOperator without an imaginary part), Jacobi preconditioner applied to both parts.
    Against a numpy restatement of the same recurrence on the oracle's matrix: iteration count, residual norms in the
    preconditioner's inner product, iterate; with and without an initial guess."""
    mesh, p, q1d = cylinder_mesh, 2, 3
    ctx = linalg.Context()
    nd = NDHexSpace(mesh, p)
    n, ess = nd.ndofs, nd.ess_dofs()
    geom = ceed.GeomFactorData(mesh, q1d)
    ident = ceed.coefficient_context(3)
    mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
    op = ceed.curlcurlmass_operator(geom, nd, mass, ident)
    A = linalg.ComplexParOperator(ctx, op, None, ess, linalg.DIAG_ONE)
    Ar = linalg.ParOperator(ctx, op, ess, linalg.DIAG_ONE)
    J = linalg.jacobi(ctx, Ar)
    ogeom = util.oracle_geom(mesh, q1d)
    off, ori = nd.native_restriction()
    interp, curl = po.nd_hex_dense_tables(p, q1d, nd.dof_map_native())
    Ao = po.CeedOperatorOracle(n, off, ori, interp, curl, ogeom, po.QF_HDIVMASS,
                               po.CoeffCtx(attr_mat=[0], mat_coeff=[np.array([2.08])]), po.CoeffCtx()).assemble_sparse().tolil()
    for d in ess:
        Ao[d, :] = 0
        Ao[:, d] = 0
        Ao[d, d] = 1.0
    Ao = Ao.tocsr()
    dinv = 1.0 / Ao.diagonal()

    def pcg(b, x0, rel_tol, max_it):  # iterative.cpp:360-486, ScalarType complex, Dot(x, y) = y^H x
        x = np.zeros_like(b) if x0 is None else x0.copy()
        r = b - Ao @ x if x0 is not None else b.copy()
        z = dinv * r
        beta = np.vdot(r, z)
        res = np.sqrt(abs(beta))
        init = np.sqrt(abs(np.vdot(b, dinv * b))) if x0 is not None else res
        eps, it, beta_prev, pv = rel_tol * init, 0, 0.0, None
        while it < max_it and not res < eps:
            pv = z.copy() if it == 0 else z + (beta / beta_prev) * pv
            z = Ao @ pv
            alpha = beta / np.vdot(pv, z)
            x, r = x + alpha * pv, r - alpha * z
            beta_prev = beta
            z = dinv * r
            beta = np.vdot(r, z)
            res = np.sqrt(abs(beta))
            it += 1
        return x, it, init, res

    rng = np.random.default_rng(21)
    b = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    b[ess] = 0.0
    S = linalg.ComplexParCg(ctx, A, J, rel_tol=1e-10, max_it=2000)
    for x0 in (None, 0.3 * (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n))):
        if x0 is not None:
            x0[ess] = 0.0
        xo, it_o, init_o, res_o = pcg(b, x0, 1e-10, 2000)
        xr = _dev(x0.real) if x0 is not None else torch.zeros(n, dtype=torch.float64, device="cuda")
        xi = _dev(x0.imag) if x0 is not None else torch.zeros(n, dtype=torch.float64, device="cuda")
        S.mult(_dev(b.real), _dev(b.imag), xr, xi, initial_guess=x0 is not None)
        st = S.stats()
        assert st["converged"] and abs(st["iterations"] - it_o) <= 1, (st, it_o)
        assert abs(st["initial_res"] - init_o) < 1e-12 * init_o
        xs = xr.cpu().numpy() + 1j * xi.cpu().numpy()
        assert np.linalg.norm(xs - xo) < 1e-8 * np.linalg.norm(xo)
        assert np.linalg.norm(Ao @ xs - b) < 1e-8 * np.linalg.norm(b)


@pytest.mark.parametrize("elements,mat", [("hex", "iso"), ("hex", "aniso"), ("tet", "aniso")])
def test_fused_complex_apply_with_surface_terms(elements, mat):
    """A(omega) = K + i omega C - omega^2 M + A2(omega) with surface terms in BOTH parts (models/spaceoperator.cpp:786-804: impedance /
    absorbing boundaries and ports add f_apply_hcurl_32 terms to C, A2 adds them to the real and the imaginary part): the volume
    operators pair up in the one-pass kernel (pa_op_complex_fused = 3 on hexahedra, 2 on tetrahedra), every further sub-operator of
    either part is applied after it to both parts of x.  Against the SAME operators applied one by one and combined term by term
    as linalg/operator.cpp:98-134 does (each of them is checked against the oracle in test_apply_gpu / test_dense_gpu /
    test_tet_gpu), plain and through ComplexParOperator with essential dofs (rap.cpp:450-457)."""
    from palace_amd.fem.mesh import ogrid_cylinder

    ctx = linalg.Context()
    T = _tensors()
    two = lambda a, b: ceed.coefficient_context(3, attr_mat=[0, 1], mat_coeff=[np.asarray(a, float), np.asarray(b, float)])  # noqa: E731
    if mat == "iso":
        cm_r, cc_r, cm_i = two(-0.9, -0.35), two(1.0, 0.6), two(0.21, 0.05)
    else:
        cm_r, cc_r, cm_i = two(T["mr0"], T["mr1"]), two(T["cr0"], T["cr1"]), two(T["mi0"], T["mi1"])
    if elements == "hex":
        from palace_amd.fem.fespace import NDHexBoundaryBlock

        mesh = ogrid_cylinder(2, 3)
        mesh.attr[:] = 1 + (np.arange(mesh.ne) % 2)
        nd = NDHexSpace(mesh, 3)
        vgeom = ceed.GeomFactorData(mesh, 4)
        nb = int(mesh.boundary_face_mask[mesh.elem_faces].sum())
        blk = NDHexBoundaryBlock(nd, attr=1 + (np.arange(nb) % 2))
        interp, grad, w = blk.tables(4)
        sgeom = ceed.DenseGeomFactorData(blk.elem_nodes, blk.nodes, blk.attr, grad, w)
        sblock = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, blk.offsets, interp, None, orients=blk.orients)

        def volume(op, qf, blob, ev):
            return op.add_integrator(vgeom, nd, qf, blob, ev)

        ess = nd.ess_dofs()[::2].copy()  # (part of the boundary: the surface terms touch free and essential dofs)
    else:
        from palace_amd.fem import tet, tri

        mesh = tet.cube_tet_mesh(3)
        mesh.attr[:] = 1 + (np.arange(mesh.ne) % 2)
        nd = tet.NDTetSpace(mesh, 2)
        pts, wts = tet.default_tet_rule(2)
        vi, vc = nd.elem.tables(pts)
        vgeom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
        kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
        vblock = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, vi, vc, **kw)
        faces = np.nonzero(mesh.boundary_face_mask)[0]
        blk = tet.NDTetBoundaryBlock(nd, faces, 1 + (np.arange(faces.size) % 2))
        spts, swts = tri.tri_quadrature(3)
        si, _ = blk.elem.tables(spts)
        sgeom = ceed.DenseGeomFactorData(blk.elem_nodes, blk.nodes, blk.attr, blk.geometry_grad_table(spts), swts)
        sblock = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, blk.offsets, si, None, orients=blk.orients)

        def volume(op, qf, blob, ev):
            return op.add_dense_integrator(vgeom, vblock, qf, blob, ev)

        ess = nd.ess_dofs()[::2].copy()
    n = nd.ndofs
    s_r, s_i = two(0.11, -0.07), two(0.4, 0.25)  # A2's real part (a reactive surface term) and the damping surface term

    def surface(op, blob):
        return op.add_dense_integrator(sgeom, sblock, ceed.QF_HCURL_32, blob, ceed.EVAL_INTERP)

    new = lambda: ceed.Operator(n, n)  # noqa: E731
    both = ceed.EVAL_CURL | ceed.EVAL_INTERP
    Ar = surface(volume(new(), ceed.QF_HDIVMASS_33, np.concatenate([cm_r, cc_r]), both), s_r).finalize()
    Ai = surface(volume(new(), ceed.QF_HCURL_33, cm_i, ceed.EVAL_INTERP), s_i).finalize()
    parts = [volume(new(), ceed.QF_HDIVMASS_33, np.concatenate([cm_r, cc_r]), both).finalize(), surface(new(), s_r).finalize(),
             volume(new(), ceed.QF_HCURL_33, cm_i, ceed.EVAL_INTERP).finalize(), surface(new(), s_i).finalize()]
    assert ceed._lib.load().pa_op_complex_fused(Ar.handle, Ai.handle) == (3 if elements == "hex" else 2)
    rng = np.random.default_rng(8)
    xr, xi = (_dev(rng.uniform(-1, 1, n)) for _ in range(2))

    def term_by_term(vr, vi):
        def app(o, v):
            y = torch.empty_like(v)
            o.mult(v, y)
            return y

        a_r = lambda v: app(parts[0], v) + app(parts[1], v)  # noqa: E731
        a_i = lambda v: app(parts[2], v) + app(parts[3], v)  # noqa: E731
        return a_r(vr) - a_i(vi), a_i(vr) + a_r(vi)

    for e in (np.zeros(0, np.int32), ess):
        A = linalg.ComplexParOperator(ctx, Ar, Ai, e, linalg.DIAG_ONE)
        yr, yi = torch.empty_like(xr), torch.empty_like(xr)
        A.mult(xr, xi, yr, yi)
        mr, mi = xr.clone(), xi.clone()
        ed = torch.from_numpy(e.astype(np.int64)).cuda()
        mr[ed], mi[ed] = 0.0, 0.0
        wr, wi = term_by_term(mr, mi)
        wr[ed], wi[ed] = xr[ed], xi[ed]
        scale = float(torch.maximum(wr.abs().max(), wi.abs().max()))
        assert float((yr - wr).abs().max()) < 1e-13 * scale and float((yi - wi).abs().max()) < 1e-13 * scale, (elements, mat, e.size)
