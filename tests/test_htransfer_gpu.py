"""h-levels of the multigrid hierarchy on the device (reference fem/multigrid.hpp:103-112, fem/fespace.cpp:246-251):
the refinement transfer (pa_interp_create_refinement) against the oracle restatement on hexahedra and tetrahedra, and PCG with
the V-cycle over [h-levels] + [p-levels] against the oracle's PCG + V-cycle (iteration counts, iterate)."""
import numpy as np
import pytest
import torch

from oracle import palace_oracle as po
from palace_amd import linalg
from palace_amd.fem import htransfer
from palace_amd.fem.fespace import H1HexSpace, NDHexSpace
from palace_amd.fem.mesh import ogrid_cylinder, refine_uniform
from tests import util

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def _hex_oracle(c, f, hcurl):
    parent = np.arange(f.mesh.ne) // 8
    sc = c.elem_sign_lex if hcurl else np.ones_like(c.elem_dof_lex, dtype=np.int8)
    sf = f.elem_sign_lex if hcurl else np.ones_like(f.elem_dof_lex, dtype=np.int8)
    return po.RefinementTransferOracle(c.elem_dof_lex[parent], sc[parent], f.elem_dof_lex, sf, c.ndofs, f.ndofs,
                                       po.hex_refinement_matrices(c.p, hcurl), np.arange(f.mesh.ne) % 8)


@pytest.mark.parametrize("p", [1, 2])
@pytest.mark.parametrize("hcurl", [True, False])
def test_hex_refinement_transfer_matches_the_oracle(cylinder_mesh, p, hcurl):
    mc = cylinder_mesh
    mf = refine_uniform(mc)
    Space = NDHexSpace if hcurl else H1HexSpace
    c, f = Space(mc, p), Space(mf, p)
    ctx = linalg.Context()
    P = linalg.RefinementTransfer(ctx, *htransfer.hex_refinement(c, f))
    oP = _hex_oracle(c, f, hcurl)
    rng = np.random.default_rng(1)
    xc, xf = rng.uniform(-1, 1, c.ndofs), rng.uniform(-1, 1, f.ndofs)
    y = P.mult(_dev(xc), torch.full((f.ndofs,), np.nan, dtype=torch.float64, device="cuda")).cpu().numpy()
    assert _rel(y, oP.mult(xc)) < 1e-13
    z = P.mult_transpose(_dev(xf), torch.full((c.ndofs,), np.nan, dtype=torch.float64, device="cuda")).cpu().numpy()
    assert _rel(z, oP.mult_transpose(xf)) < 1e-13
    assert abs(xf @ y - z @ xc) < 1e-12 * np.linalg.norm(xf) * np.linalg.norm(y)


def test_tet_refinement_transfer_matches_the_oracle():
    from palace_amd.fem import tet

    mc = tet.cube_tet_mesh(3)
    mf = tet.refine_uniform(mc)
    ctx = linalg.Context()
    sg = lambda o: np.where(o, -1.0, 1.0)  # noqa: E731
    for Space in (tet.NDTetSpace, tet.H1TetSpace):
        c, f = Space(mc, 1), Space(mf, 1)
        dom, rng_, Ms, mid = htransfer.tet_refinement(c, f)
        P = linalg.RefinementTransfer(ctx, dom, rng_, Ms, mid)
        one = np.ones_like(dom["offsets"], dtype=np.float64)
        oP = po.RefinementTransferOracle(dom["offsets"], sg(dom["orients"]) if "orients" in dom else one, rng_["offsets"],
                                         sg(rng_["orients"]) if "orients" in rng_ else np.ones_like(rng_["offsets"], dtype=np.float64),
                                         c.ndofs, f.ndofs, Ms, mid)
        rng = np.random.default_rng(2)
        xc, xf = rng.uniform(-1, 1, c.ndofs), rng.uniform(-1, 1, f.ndofs)
        y = P.mult(_dev(xc), torch.empty(f.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
        z = P.mult_transpose(_dev(xf), torch.empty(c.ndofs, dtype=torch.float64, device="cuda")).cpu().numpy()
        assert _rel(y, oP.mult(xc)) < 1e-13 and _rel(z, oP.mult_transpose(xf)) < 1e-13


def test_pcg_with_an_h_level_under_the_p_levels_matches_the_oracle(cylinder_mesh):
    """The reference cylinder once refined, hierarchy = [order 1 on the coarse mesh] + [orders 1, 2, 3 on the fine mesh]
    (fem/multigrid.hpp:103-123), plain Chebyshev smoothers of order 6, Jacobi-PCG(8) on the coarsest level: the V-cycle and the
    PCG iteration count against the oracle's, the solution against the one of the p-levels-only hierarchy."""
    from palace_amd.fem.hproblem import HpProblem

    ctx = linalg.Context()
    prob = HpProblem(ctx, cylinder_mesh, 1, 3)
    assert [(m, q) for m, q in prob.levels] == [(0, 1), (1, 1), (1, 2), (1, 3)]
    K, b, x = prob.pcg_gmg_solver(max_it=200, rel_tol=1e-8, hiptmair=False, coarse="cg")
    K.mult(b, x)
    st = K.stats()
    assert st["converged"], st
    # oracle: the same hierarchy
    q1d = 4
    cm, bm = util.make_ctx("scalar")
    cc, bc = util.make_ctx("identity")
    blob = np.concatenate([bm, bc])
    ogeoms = [util.oracle_geom(m, q1d) for m in prob.meshes]
    oA = [util.FastParOperatorOracle(s, ogeoms[m], "hdivmass", blob, s.ess_dofs(), q1d, cm, cc) for s, (m, _) in zip(prob.spaces, prob.levels)]
    oP = []
    for l in range(len(prob.levels) - 1):
        c, f = prob.spaces[l], prob.spaces[l + 1]
        if prob.levels[l][0] == prob.levels[l + 1][0]:
            o = po.InterpOracle(c.elem_dof_lex, c.elem_sign_lex, f.elem_dof_lex, f.elem_sign_lex, c.ndofs, f.ndofs, po.nd_hex_interp_lex(c.p, f.p))
        else:
            o = _hex_oracle(c, f, True)
        oP.append((o.mult, o.mult_transpose))
    nl = len(prob.levels)
    sm = [None] + [po.ChebyshevOracle(oA[l], 6, lambda_max=prob.last_gmg.gmg_lambda_max(l)) for l in range(1, nl)]
    d0 = 1.0 / oA[0].diagonal()
    coarse = lambda r: po.pcg(oA[0].mult, r, lambda v: d0 * v, rel_tol=1e-2, max_it=8)[0]  # noqa: E731
    oB = po.GMGOracle(oA, oP, sm, coarse, [s.ess_dofs() for s in prob.spaces])
    n = prob.spaces[-1].ndofs
    r = np.random.default_rng(8).uniform(-1, 1, n)
    r[prob.ess[-1]] = 0.0
    z = prob.last_gmg.mult(_dev(r), torch.empty(n, dtype=torch.float64, device="cuda")).cpu().numpy()
    assert _rel(z, oB.mult(r)) < 1e-8
    xo, it, hist = po.pcg(oA[-1].mult, b.cpu().numpy(), oB.mult, rel_tol=1e-8, max_it=200)
    assert abs(st["iterations"] - it) <= 1, (st, it)
    assert _rel(x.cpu().numpy(), xo) < 1e-6
    # the p-levels alone on the fine mesh reach the same solution; the h-level under them does not cost iterations
    prob0 = HpProblem(ctx, prob.meshes[1], 0, 3)
    K0, b0, x0 = prob0.pcg_gmg_solver(max_it=200, rel_tol=1e-8, hiptmair=False, coarse="cg")
    K0.mult(b0, x0)
    assert _rel(x.cpu().numpy(), x0.cpu().numpy()) < 1e-6
    assert st["iterations"] <= K0.stats()["iterations"] + 2, (st, K0.stats())


@pytest.mark.parametrize("hiptmair", [False, True])
def test_h_and_p_levels_with_ams_on_the_coarse_mesh(hiptmair):
    """Two h-levels under p = 1, 2 with the native AMS on the 64x smaller coarsest mesh, plain and auxiliary-space smoothers:
    converges to the solution of the one-mesh hierarchy."""
    from palace_amd.fem.hproblem import HpProblem

    ctx = linalg.Context()
    prob = HpProblem(ctx, ogrid_cylinder(2, 2), 2, 2)
    K, b, x = prob.pcg_gmg_solver(max_it=100, rel_tol=1e-10, hiptmair=hiptmair, coarse="ams")
    K.mult(b, x)
    st = K.stats()
    assert st["converged"], st
    prob0 = HpProblem(ctx, prob.meshes[-1], 0, 2)
    K0, b0, x0 = prob0.pcg_gmg_solver(max_it=200, rel_tol=1e-10, hiptmair=hiptmair, coarse="ams")
    K0.mult(b0, x0)
    st0 = K0.stats()
    assert st0["converged"]
    assert _rel(x.cpu().numpy(), x0.cpu().numpy()) < 1e-7
    # (the h-levels put AMS on a 64x smaller problem; the V-cycle over them is as good a preconditioner as AMS on the fine mesh)
    assert st["iterations"] <= int(1.25 * st0["iterations"]) + 2, (st, st0)
    if hiptmair:
        assert st["iterations"] <= 25, st


def test_cxx_hierarchy_with_h_levels(tmp_path):
    """The C++ face: Mesh::SetRefinementTransforms + FiniteElementSpaceHierarchy::AddLevel across two meshes +
    BilinearForm::Assemble(hierarchy) + KspSolver (examples/cxx_host/solve_hp.cpp on the arrays of dump_problem_hp.py) against
    the same hierarchy through the ctypes mirror (HpProblem): same iteration count, same solution."""
    import os
    import re
    import shutil
    import subprocess
    import sys

    from palace_amd.fem.hproblem import HpProblem

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    sys.path.insert(0, os.path.join(root, "examples", "cxx_host"))
    import dump_problem_hp

    blob, exe = str(tmp_path / "hp.bin"), str(tmp_path / "solve_hp")
    dump_problem_hp.main(blob, 2, 1, 2, 3)
    libdir = os.path.join(root, "palace_amd", "lib")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-std=c++17", "-O2", "-w", "-I" + os.path.join(root, "palace_amd", "csrc"),
                           "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "cxx_host", "solve_hp.cpp"),
                           "-L" + libdir, "-lpalace_amd", "-Wl,-rpath," + libdir, "-o", exe])
    ctx = linalg.Context()
    for aux, coarse in ((0, "pcg"), (1, "pcg"), (1, "ams")):
        out = subprocess.check_output([exe, blob, str(aux), coarse], text=True)
        m = re.search(r"meshes (\d+)\s+levels (\d+)\s+ndofs (\d+)\s+coarsest (\d+) .* iterations (\d+)\s+converged (\d)\s+"
                      r"\|b - A x\| / \|b\| (\S+)\s+sum\(x\) (\S+)", out)
        assert m, out
        nmesh, nlev, n, n0, its, conv = (int(m.group(i)) for i in range(1, 7))
        res, sx = float(m.group(7)), float(m.group(8))
        prob = HpProblem(ctx, ogrid_cylinder(2, 3), 1, 2)
        assert (nmesh, nlev, n, n0) == (2, 3, prob.spaces[-1].ndofs, prob.spaces[0].ndofs), out
        assert conv == 1 and res < 1e-8, out
        K, b, x = prob.pcg_gmg_solver(max_it=400, rel_tol=1e-10, hiptmair=bool(aux), coarse="cg" if coarse == "pcg" else coarse)
        K.mult(b, x)
        st = K.stats()
        assert st["converged"] and abs(st["iterations"] - its) <= 1, (out, st)
        assert abs(float(x.sum()) - sx) < 1e-7 * abs(sx), (out, float(x.sum()))


def test_tet_pcg_with_an_h_level_under_the_p_levels():
    """Tetrahedra: hierarchy [order 1 on the coarse mesh] + [orders 1, 2 on its uniform refinement] (the refinement transfer with one
    local interpolation matrix per child embedding, then the element-matrix p-prolongation), dense MFMA operators assembled per mesh,
    PCG: same solution as the p-levels alone on the fine mesh, not more iterations."""
    from palace_amd import ceed
    from palace_amd.fem import tet
    from palace_amd.fem.tetproblem import TetProblem

    ctx = linalg.Context()
    mc = tet.cube_tet_mesh(3)
    mf = tet.refine_uniform(mc)
    fine = TetProblem(ctx, mf, 2)            # orders 1, 2 on the fine mesh
    coarse = TetProblem(ctx, mc, 1)          # order 1 on the coarse mesh
    # both problems integrate with the rule of the solution order (the fine problem's): rebuild the coarse geometry on it
    coarse.pts, coarse.wts = fine.pts, fine.wts
    coarse.geom = ceed.DenseGeomFactorData(mc.elem_nodes, mc.nodes, mc.attr, mc.geometry_grad_table(fine.pts), fine.wts)
    mass = ceed.coefficient_context(3, attr_mat=[0] * int(mf.attr.max()), mat_coeff=[np.array([2.08])])
    curl = ceed.coefficient_context(3)
    blob = np.concatenate([mass, curl])

    def assemble(prob, s):
        return ceed.Operator(s.ndofs, s.ndofs).add_dense_integrator(prob.geom, prob.nd_block(s), ceed.QF_HDIVMASS_33, blob,
                                                                    ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize()

    f2 = assemble(fine, fine.spaces[1])
    f1 = f2.coarsen_dense(fine.nd_block(fine.spaces[0]))
    c1 = assemble(coarse, coarse.spaces[0])
    spaces = [coarse.spaces[0], fine.spaces[0], fine.spaces[1]]
    ess = [s.ess_dofs() for s in spaces]
    local = [c1, f1, f2]
    A = [linalg.ParOperator(ctx, op, e, linalg.DIAG_ONE) for op, e in zip(local, ess)]
    A[0] = linalg.AssembledParOperator(ctx, c1.full_assemble_device(), ess[0], linalg.DIAG_ONE)
    P = [linalg.RefinementTransfer(ctx, *htransfer.tet_refinement(spaces[0], spaces[1])),
         linalg.DenseInterp(ctx, spaces[1].restriction(), spaces[2].restriction(interp_range=True), tet.nd_tet_transfer_matrix(1, 2))]
    cs = linalg.cg(ctx, A[0], linalg.jacobi(ctx, A[0]), rel_tol=1e-2, max_it=8)
    B = linalg.gmg(ctx, A, P, cs, cheby_order=4)
    K = linalg.cg(ctx, A[-1], B, rel_tol=1e-9, max_it=300)
    n = spaces[-1].ndofs
    b = torch.empty(n, dtype=torch.float64, device="cuda")
    A[-1].mult(torch.ones(n, dtype=torch.float64, device="cuda"), b)
    b[torch.from_numpy(ess[-1].astype(np.int64)).cuda()] = 0.0
    x = torch.zeros_like(b)
    K.mult(b, x)
    st = K.stats()
    assert st["converged"], st
    K0, b0, x0 = fine.pcg_gmg_solver(max_it=300, rel_tol=1e-9, hiptmair=False, coarse="cg")
    K0.mult(b0, x0)
    st0 = K0.stats()
    assert st0["converged"]
    assert _rel(b.cpu().numpy(), b0.cpu().numpy()) < 1e-12
    assert _rel(x.cpu().numpy(), x0.cpu().numpy()) < 1e-6
    assert st["iterations"] <= st0["iterations"] + 3, (st, st0)
