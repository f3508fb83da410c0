"""pa_op_mult_split: the local apply on split vectors (true dofs in the caller's x / y, ghosts read from one of two buffers
chosen by a device-resident counter and written to their own array) -- what lets a multi-rank ParOperator::Mult run without
L-vector copies (linalg/rap.cpp:195-234 does tx = x, lx = P tx, ly = A lx, y = P^T ly).  Against the plain apply on the
concatenated vectors, bit for bit: the same kernels read and write the same numbers through two base pointers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from palace_amd import ceed  # noqa: E402
from palace_amd.fem.fespace import NDHexSpace  # noqa: E402


@pytest.mark.parametrize("kind", ["curl", "curlmass", "mass"])
@pytest.mark.parametrize("p,q1d", [(1, 4), (2, 4), (3, 4), (4, 5), (2, 5)])
def test_split_apply_equals_plain_apply(cylinder_mesh, p, q1d, kind):
    """Four-point kernels (orders 1-3) and the five-point ones (order 4 and a coarsened level of an order-4 problem)."""
    mesh = cylinder_mesh
    nd = NDHexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, q1d)
    mass = ceed.coefficient_context(3, attr_mat=[0] * int(mesh.attr.max()), mat_coeff=[np.array([2.08])])
    ident = ceed.coefficient_context(3)
    make = {"curl": lambda: ceed.curlcurl_operator(geom, nd, ident),
            "curlmass": lambda: ceed.curlcurlmass_operator(geom, nd, mass, ident),
            "mass": lambda: ceed.ndmass_operator(geom, nd, mass)}[kind]
    n = nd.ndofs
    rng = np.random.default_rng(p)
    x = torch.from_numpy(rng.uniform(-1, 1, n)).cuda()
    for n_true in (n, int(0.7 * n), 1):
        ng = n - n_true
        op = make()
        assert op.supports_split()
        ref = torch.empty_like(x)
        op.mult(x, ref)
        # the ghost input in the second of two buffers, the first one poisoned: the selector's parity must be honoured
        xg0 = torch.full((max(ng, 1),), float("nan"), dtype=torch.float64, device="cuda")
        xg1 = x[n_true:].clone() if ng else torch.zeros(1, dtype=torch.float64, device="cuda")
        sel = torch.tensor([7], dtype=torch.int64, device="cuda")
        y, yg = torch.empty(n_true, dtype=torch.float64, device="cuda"), torch.zeros(max(ng, 1), dtype=torch.float64, device="cuda")
        op.mult_split(x[:n_true].clone(), xg0, y, yg, xg1=xg1, sel=sel)
        assert torch.equal(y, ref[:n_true])
        if ng:
            assert torch.equal(yg, ref[n_true:])
        # essential dofs among the true dofs: masked on input, rows fixed on output (policy 1: y = x, 0: y = 0)
        ess = np.sort(rng.choice(n_true, size=max(1, n_true // 7), replace=False)).astype(np.int32)
        op2 = make()
        op2.set_essential(ess)
        xm = x.clone()
        xm[torch.from_numpy(ess.astype(np.int64)).cuda()] = 0.0
        op.mult(xm, ref)
        for policy in (1, 0):
            want = ref.clone()
            ie = torch.from_numpy(ess.astype(np.int64)).cuda()
            want[ie] = x[ie] if policy else 0.0
            xg = x[n_true:].clone() if ng else torch.zeros(1, dtype=torch.float64, device="cuda")
            op2.mult_split(x[:n_true].clone(), xg, y, yg, ess_policy=policy)
            assert torch.equal(y, want[:n_true]), (kind, p, n_true, policy)
            if ng:
                assert torch.equal(yg, want[n_true:])


def test_split_apply_is_refused_where_there_is_no_such_form(cylinder_mesh):
    from palace_amd.fem.fespace import H1HexSpace

    nd = H1HexSpace(cylinder_mesh, 2)
    op = ceed.diffusion_operator(ceed.GeomFactorData(cylinder_mesh, 3), nd, ceed.coefficient_context(3))
    assert not op.supports_split()
    x = torch.zeros(nd.ndofs, dtype=torch.float64, device="cuda")
    with pytest.raises(Exception, match="split"):
        op.mult_split(x[:10].clone(), x[10:].clone(), torch.empty(10, dtype=torch.float64, device="cuda"), x[10:].clone())


def _check_split(make, n, seed, exact=True):
    """op.mult_split on (true, ghost) pieces of a vector against op.mult on the whole one, for several split points, with the
    ghost input in the second mailbox buffer, and with essential dofs fused (rows fixed to x or 0)."""
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.uniform(-1, 1, n)).cuda()
    same = (lambda a, b: torch.equal(a, b)) if exact else (lambda a, b: torch.allclose(a, b, rtol=0, atol=2e-13 * float(b.abs().max())))
    for n_true in (n, int(0.7 * n), 1):
        ng = n - n_true
        op = make()
        assert op.supports_split()
        ref = torch.empty_like(x)
        op.mult(x, ref)
        xg0 = torch.full((max(ng, 1),), float("nan"), dtype=torch.float64, device="cuda")
        xg1 = x[n_true:].clone() if ng else torch.zeros(1, dtype=torch.float64, device="cuda")
        sel = torch.tensor([7], dtype=torch.int64, device="cuda")
        y, yg = torch.empty(n_true, dtype=torch.float64, device="cuda"), torch.zeros(max(ng, 1), dtype=torch.float64, device="cuda")
        op.mult_split(x[:n_true].clone(), xg0, y, yg, xg1=xg1, sel=sel)
        assert same(y, ref[:n_true])
        if ng:
            assert same(yg, ref[n_true:])
        ess = np.sort(rng.choice(n_true, size=max(1, n_true // 7), replace=False)).astype(np.int32)
        op2 = make()
        op2.set_essential(ess)
        ie = torch.from_numpy(ess.astype(np.int64)).cuda()
        xm = x.clone()
        xm[ie] = 0.0
        op.mult(xm, ref)
        for policy in (1, 0):
            want = ref.clone()
            want[ie] = x[ie] if policy else 0.0
            xg = x[n_true:].clone() if ng else torch.zeros(1, dtype=torch.float64, device="cuda")
            op2.mult_split(x[:n_true].clone(), xg, y, yg, ess_policy=policy)
            assert same(y, want[:n_true]), (n_true, policy)
            if ng:
                assert same(yg, want[n_true:])


@pytest.mark.parametrize("kind", ["diffusion", "mass", "diffusionmass"])
@pytest.mark.parametrize("p", [1, 2, 3])
def test_split_apply_h1_hexahedra(cylinder_mesh, p, kind):
    """The auxiliary-space levels of the Hiptmair smoother: H1 hexahedra at four points per direction, every order through the
    streaming kernel (orders 1, 2 keep the one-shot kernel for the plain apply: another summation order, hence a tolerance)."""
    from palace_amd.fem.fespace import H1HexSpace

    mesh = cylinder_mesh
    h1 = H1HexSpace(mesh, p)
    geom = ceed.GeomFactorData(mesh, 4)
    eps = ceed.coefficient_context(3, attr_mat=[0] * int(mesh.attr.max()), mat_coeff=[np.array([2.08])])
    eps1 = ceed.coefficient_context(1, attr_mat=[0] * int(mesh.attr.max()), mat_coeff=[np.array([0.7])])
    make = {"diffusion": lambda: ceed.diffusion_operator(geom, h1, eps),
            "mass": lambda: ceed.h1mass_operator(geom, h1, eps1),
            "diffusionmass": lambda: ceed.diffusionmass_operator(geom, h1, eps1, eps)}[kind]
    _check_split(make, h1.ndofs, 10 + p, exact=(p == 3))


@pytest.mark.parametrize("kind", ["curl", "curlmass", "mass", "h1diffusion"])
@pytest.mark.parametrize("p", [1, 2, 3])
def test_split_apply_dense_tetrahedra(p, kind):
    """The dense-table path (LDS-resident kernel on the matrix cores): Nedelec and H1 tetrahedra, curl-oriented restriction."""
    from palace_amd.fem import tet

    mesh = tet.cube_tet_mesh(3)
    pts, wts = tet.default_tet_rule(p)
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
    ident = ceed.coefficient_context(3)
    mass = ceed.coefficient_context(3, attr_mat=[0] * int(mesh.attr.max()), mat_coeff=[np.array([2.08])])
    if kind == "h1diffusion":
        sp = tet.H1TetSpace(mesh, p)
        interp, grad = sp.elem.tables(pts)
        block = ceed.DenseBlock(ceed.FE_H1, sp.ndofs, sp.offsets, interp, grad)
        make = lambda: ceed.Operator(sp.ndofs, sp.ndofs).add_dense_integrator(geom, block, ceed.QF_HCURL_33, mass, ceed.EVAL_GRAD).finalize()  # noqa: E731
    else:
        sp = tet.NDTetSpace(mesh, p)
        interp, curl = sp.elem.tables(pts)
        kw = dict(orients=sp.orients) if sp.diagonal_transform else dict(curl_orients=sp.curl_orients)
        block = ceed.DenseBlock(ceed.FE_HCURL, sp.ndofs, sp.offsets, interp, curl, **kw)
        qf, blob, ev = {"curl": (ceed.QF_HDIV_33, ident, ceed.EVAL_CURL),
                        "mass": (ceed.QF_HCURL_33, mass, ceed.EVAL_INTERP),
                        "curlmass": (ceed.QF_HDIVMASS_33, np.concatenate([mass, ident]), ceed.EVAL_CURL | ceed.EVAL_INTERP)}[kind]
        make = lambda: ceed.Operator(sp.ndofs, sp.ndofs).add_dense_integrator(geom, block, qf, blob, ev).finalize()  # noqa: E731
    _check_split(make, sp.ndofs, 20 + p)
