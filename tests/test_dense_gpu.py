"""Dense-table (non-tensor) path on the FP64 matrix cores vs the oracle, bit-level up to rounding.

The kernel takes exactly what Palace gives libCEED on its non-tensor path (dense tables, native
restriction), so it can be exercised with ANY element's tables: here the hexahedral Nedelec / H1
tables the oracle evaluates point-wise (the reference uses the dense path for ND hexes too,
fem/libceed/basis.cpp:40-85), with the plain, the oriented and a synthetic curl-oriented
(tridiagonal) restriction.  Tetrahedral spaces are covered in test_tet_gpu.py."""
import numpy as np
import pytest

from oracle import palace_oracle as po
from tests import util

pytestmark = pytest.mark.gpu

REL = 1e-12


def _setup(cylinder_mesh, p, q1d, h1):
    from palace_amd import ceed
    from palace_amd.fem.fespace import H1HexSpace, NDHexSpace

    mesh = cylinder_mesh
    _, wts = po.hex_quadrature(q1d)
    G = po.mesh_q2_grad_table(q1d)
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.x, mesh.attr, G, wts)
    if h1:
        sp = H1HexSpace(mesh, p)
        interp, grad = po.h1_hex_dense_tables(p, q1d)
        blk = dict(fe_type=ceed.FE_H1, lsize=sp.ndofs, offsets=sp.elem_dof_lex,
                   interp=np.asarray(interp).reshape(1, -1, sp.P), deriv=np.asarray(grad).reshape(3, -1, sp.P))
        ori = None
    else:
        sp = NDHexSpace(mesh, p)
        off, ori = sp.native_restriction()
        interp, curl = util.dense_tables(sp, q1d)
        blk = dict(fe_type=ceed.FE_HCURL, lsize=sp.ndofs, offsets=off,
                   interp=np.asarray(interp).reshape(3, -1, sp.P), deriv=np.asarray(curl).reshape(3, -1, sp.P))
    return mesh, sp, geom, blk, ori


def test_dense_geometry_matches_oracle(cylinder_mesh):
    mesh, sp, geom, blk, ori = _setup(cylinder_mesh, 1, 3, False)
    ref = util.oracle_geom(mesh, 3)
    got = geom.to_numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()


MODES = [  # name, h1, qf (lib), qf (oracle), ops, contexts
    ("curl", False, "QF_HDIV_33", po.QF_HDIV, "C", 1),
    ("vmass", False, "QF_HCURL_33", po.QF_HCURL, "I", 1),
    ("curlmass", False, "QF_HDIVMASS_33", po.QF_HDIVMASS, "CI", 2),
    ("diff", True, "QF_HCURL_33", po.QF_HCURL, "G", 1),
    ("diffmass", True, "QF_HCURLMASS_33", po.QF_HCURLMASS, "GI", 2),
    ("mass", True, "QF_H1_1", po.QF_H1MASS, "I", 1),
]


def _contexts(name, kind, nattr):
    c3, b3 = util.make_ctx(kind, nattr)
    if name in ("curl", "vmass", "diff"):
        return (c3, None), b3
    if name == "curlmass":
        cm, bm = util.make_ctx("scalar", nattr)
        return (cm, c3), np.concatenate([bm, b3])
    c1 = po.CoeffCtx(attr_mat=[0] * nattr, mat_coeff=[np.array([1.7])], dim=1)
    if name == "mass":
        return (c1, None), c1.pack()
    return (c1, c3), np.concatenate([c1.pack(), b3])


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
@pytest.mark.parametrize("p,q1d", [(1, 2), (2, 3), (3, 4), (2, 4)])
@pytest.mark.parametrize("restr", ["native", "curl_oriented"])
def test_dense_apply_and_diagonal(cylinder_mesh, mode, p, q1d, restr):
    import torch

    from palace_amd import ceed

    name, h1, qf_name, qf_o, ops_s, _ = mode
    mesh, sp, geom, blk, ori = _setup(cylinder_mesh, p, q1d, h1)
    nattr = int(mesh.attr.max())
    (c_a, c_b), blob = _contexts(name, "aniso" if p < 3 else "scalar", nattr)
    rng = np.random.default_rng(11 * p + q1d)
    cor = None
    if restr == "curl_oriented":
        # synthetic tridiagonal element transformations with entries in {-1, 0, 1} (the values MFEM's
        # ND_DofTransformation produces), in 2x2 blocks on a random subset of dof pairs
        ne, P = blk["offsets"].shape
        cor = np.zeros((ne, P, 3), dtype=np.int8)
        cor[:, :, 1] = rng.choice([-1, 1], size=(ne, P))
        pairs = rng.random((ne, P // 2)) < 0.5
        for k in range(P // 2):
            sel = pairs[:, k]
            blkv = rng.integers(-1, 2, size=(ne, 4)).astype(np.int8)
            a, b2 = 2 * k, 2 * k + 1
            cor[sel, a, 1], cor[sel, a, 2] = blkv[sel, 0], blkv[sel, 1]
            cor[sel, b2, 0], cor[sel, b2, 1] = blkv[sel, 2], blkv[sel, 3]
        ori_l = None
    else:
        ori_l = ori
    block = ceed.DenseBlock(orients=ori_l, curl_orients=cor, **blk)
    ops = sum({"C": ceed.EVAL_CURL, "I": ceed.EVAL_INTERP, "G": ceed.EVAL_GRAD}[c] for c in ops_s)
    op = ceed.Operator(sp.ndofs, sp.ndofs).add_dense_integrator(geom, block, getattr(ceed, qf_name), blob, ops).finalize()

    ogeom = util.oracle_geom(mesh, q1d)
    orc = po.CeedOperatorOracle(sp.ndofs, blk["offsets"], ori_l, blk["interp"], blk["deriv"], ogeom, qf_o, c_a, c_b,
                                vector_fe=not h1, curl_orients=cor)
    x = rng.uniform(-1, 1, sp.ndofs)
    y_ref = orc.apply_add(x, np.zeros(sp.ndofs))
    xd = torch.from_numpy(x).cuda()
    yd = torch.full((sp.ndofs,), 7.0, dtype=torch.float64, device="cuda")
    op.mult(xd, yd)
    err = np.abs(yd.cpu().numpy() - y_ref).max() / np.abs(y_ref).max()
    assert err < REL, f"Mult rel err {err:.3e}"
    y0 = rng.uniform(-1, 1, sp.ndofs)
    yd = torch.from_numpy(y0).cuda()
    op.add_mult(xd, yd)
    err = np.abs(yd.cpu().numpy() - (y0 + y_ref)).max() / np.abs(y_ref).max()
    assert err < REL, f"AddMult rel err {err:.3e}"
    if p <= 2:
        d_ref = orc.diagonal()
        dd = torch.empty(sp.ndofs, dtype=torch.float64, device="cuda")
        op.assemble_diagonal(dd)
        err = np.abs(dd.cpu().numpy() - d_ref).max() / np.abs(d_ref).max()
        assert err < REL, f"diagonal rel err {err:.3e}"


def test_dense_matches_tensor_kernel(cylinder_mesh):
    """Same operator through the sum-factorised hex kernel and through the dense MFMA kernel."""
    import torch

    from palace_amd import ceed

    p, q1d = 2, 3
    mesh, sp, geom, blk, ori = _setup(cylinder_mesh, p, q1d, False)
    _, blob = util.make_ctx("aniso", int(mesh.attr.max()))
    _, bm = util.make_ctx("scalar", int(mesh.attr.max()))
    ctx = np.concatenate([bm, blob])
    ops = ceed.EVAL_CURL | ceed.EVAL_INTERP
    dense = ceed.Operator(sp.ndofs, sp.ndofs).add_dense_integrator(
        geom, ceed.DenseBlock(orients=ori, **blk), ceed.QF_HDIVMASS_33, ctx, ops).finalize()
    tgeom = ceed.GeomFactorData(mesh, q1d)
    tensor = ceed.curlcurlmass_operator(tgeom, sp, bm, blob)
    x = torch.rand(sp.ndofs, dtype=torch.float64, device="cuda")
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    dense.mult(x, y1)
    tensor.mult(x, y2)
    assert float((y1 - y2).abs().max() / y2.abs().max()) < REL
    # fused essential-dof masking behaves the same
    ess = np.unique(np.random.default_rng(0).integers(0, sp.ndofs, 50)).astype(np.int32)
    from palace_amd import lib as _lib
    import ctypes as C

    for o in (dense, tensor):
        _lib.check(_lib.load().pa_op_set_essential(o.handle, ess.ctypes.data_as(C.c_void_p), len(ess)))
    for o, y in ((dense, y1), (tensor, y2)):
        _lib.check(_lib.load().pa_op_mult_essential(o.handle, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert float((y1 - y2).abs().max() / y2.abs().max()) < REL


@pytest.mark.parametrize("p", [1, 2, 3])
def test_hex_boundary_mass(cylinder_mesh, p):
    """Surface Nedelec mass (f_apply_hcurl_32) on the boundary quadrilaterals of a hexahedral space, alone and as a
    second sub-operator next to the sum-factorised volume operator (impedance-type boundary term)."""
    import torch

    from palace_amd import ceed
    from palace_amd.fem.fespace import NDHexBoundaryBlock, NDHexSpace

    mesh = cylinder_mesh
    nd = NDHexSpace(mesh, p)
    blk = NDHexBoundaryBlock(nd, attr=1 + (np.arange(int(mesh.boundary_face_mask[mesh.elem_faces].sum())) % 2))
    interp, grad, w = blk.tables(p + 1)
    bgeom = ceed.DenseGeomFactorData(blk.elem_nodes, blk.nodes, blk.attr, grad, w)
    J = np.einsum("dqn,eni->eqid", grad, blk.nodes[blk.elem_nodes])
    og = po.build_geom_factor_32(blk.attr.astype(np.float64), w, np.transpose(J, (0, 1, 3, 2)).reshape(blk.ne, -1, 6))
    got = bgeom.to_numpy()
    assert np.abs(got - og).max() <= 1e-13 * np.abs(og).max()
    c3, b3 = util.make_ctx("aniso", 2)
    block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, blk.offsets, interp, None, orients=blk.orients)
    bop = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(bgeom, block, ceed.QF_HCURL_32, b3, ceed.EVAL_INTERP).finalize()
    orc = po.CeedOperatorOracle(nd.ndofs, blk.offsets, blk.orients, interp, np.zeros((1, len(w), blk.P)), og,
                                po.QF_HCURL_32, c3)
    x = np.random.default_rng(p).uniform(-1, 1, nd.ndofs)
    ref = orc.apply_add(x, np.zeros(nd.ndofs))
    xd = torch.from_numpy(x).cuda()
    yb = torch.empty_like(xd)
    bop.mult(xd, yb)
    assert np.abs(yb.cpu().numpy() - ref).max() < REL * np.abs(ref).max()
    # volume K + M (tensor kernel) and the boundary term in one operator
    q1d = p + 1
    vgeom = ceed.GeomFactorData(mesh, q1d)
    _, bs = util.make_ctx("scalar", int(mesh.attr.max()))
    _, bi = util.make_ctx("identity")
    vol = ceed.curlcurlmass_operator(vgeom, nd, bs, bi)
    both = (ceed.Operator(nd.ndofs, nd.ndofs)
            .add_integrator(vgeom, nd, ceed.QF_HDIVMASS_33, np.concatenate([bs, bi]), ceed.EVAL_CURL | ceed.EVAL_INTERP)
            .add_dense_integrator(bgeom, block, ceed.QF_HCURL_32, b3, ceed.EVAL_INTERP).finalize())
    yv, ya = torch.empty_like(xd), torch.empty_like(xd)
    vol.mult(xd, yv)
    both.mult(xd, ya)
    assert float((ya - (yv + yb)).abs().max()) < 1e-13 * float(yv.abs().max())
