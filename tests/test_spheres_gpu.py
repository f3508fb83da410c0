"""The reference's spheres example on the device (examples/spheres/spheres.json: electrostatics, order-3 H1 on 14 362
cubic tetrahedra): the Maxwell capacitance matrix through the dense MFMA path (f_apply_hcurl_33 on gradients,
isoparametric tet20 geometry data built on the device), ParOperator with the Dirichlet dofs and Jacobi-PCG on the GPU,
gated on test/data/regression/ref/spheres/terminal-C.csv (values in the committed fixture) at 1e-6 relative -- the
reference's own regression gate is 1e-4 (test/unit/regression/cases.cpp:187-195); the CPU oracle reproduces the file to
1.3e-10 (tests/test_oracle_spheres.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_spheres_capacitance_matrix_on_device():
    from palace_amd import ceed, linalg
    from palace_amd.fem import tet

    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "spheres_mesh.npz"))
    nodes, en = d["nodes"], d["elem_nodes"].astype(np.int64)
    used, inv = np.unique(en[:, :4], return_inverse=True)
    mesh = tet.TetMesh(nodes[used], inv.reshape(-1, 4), d["attr"])
    p = 3
    h1 = tet.H1TetSpace(mesh, p)
    pts, wts = tet.default_tet_rule(p)
    interp, grad = h1.elem.tables(pts)
    G = tet.H1TetElement(3).tables(pts)[1]  # cubic geometry basis on the fixture's node order
    geom = ceed.DenseGeomFactorData(en, nodes, mesh.attr, G, wts)
    block = ceed.DenseBlock(ceed.FE_H1, h1.ndofs, h1.offsets, interp, grad)
    K = ceed.Operator(h1.ndofs, h1.ndofs).add_dense_integrator(geom, block, ceed.QF_HCURL_33, ceed.coefficient_context(3),
                                                               ceed.EVAL_GRAD).finalize()
    bt = np.sort(np.searchsorted(used, d["bdr_tris"].astype(np.int64)), axis=1)
    fkey = {tuple(f): i for i, f in enumerate(map(tuple, mesh.face_verts))}
    masks = {}
    for a in (2, 3, 4):  # 2 far field (ground), 3 sphere A, 4 sphere B
        m = np.zeros(mesh.face_verts.shape[0], dtype=bool)
        m[[fkey[tuple(f)] for f in bt[d["bdr_attr"] == a]]] = True
        masks[a] = h1.ess_dofs(m)
    ess = np.unique(np.concatenate(list(masks.values()))).astype(np.int32)
    ctx = linalg.Context()
    A = linalg.ParOperator(ctx, K, ess, linalg.DIAG_ONE)
    solver = linalg.cg(ctx, A, linalg.jacobi(ctx, A), rel_tol=1e-13, max_it=5000)
    n = h1.ndofs
    phi = []
    for a in (3, 4):
        v = torch.zeros(n, dtype=torch.float64, device="cuda")
        v[torch.from_numpy(masks[a].astype(np.int64)).cuda()] = 1.0
        b = torch.zeros_like(v)
        A.eliminate_rhs(v, b)  # b = -K_unconstrained v|ess on the free rows, b[ess] = v[ess]
        x = torch.zeros_like(v)
        solver.mult(b, x)
        assert solver.stats()["converged"], solver.stats()
        phi.append(x)
    eps0 = 1.0 / (1.25663706127e-6 * 299792458.0 ** 2)  # utils/constants.hpp:21-30
    L0 = 1.0e-2  # spheres.json "L0": mesh in cm
    t = torch.empty(n, dtype=torch.float64, device="cuda")
    C = np.zeros((2, 2))
    for i in range(2):
        K.mult(phi[i], t)  # the unconstrained local operator
        for j in range(2):
            C[j, i] = eps0 * L0 * float(phi[j] @ t)
    ref = d["C_F"]
    assert np.abs(C - ref).max() < 1e-6 * np.abs(ref).max(), (C, ref)
    assert abs(C[0, 1] - C[1, 0]) < 1e-10 * abs(C[0, 0]) and C[0, 1] < 0 < C[0, 0] < C[1, 1]


def test_spheres_capacitance_with_p_multigrid_and_native_amg():
    """BASELINE config 4's solver on the same problem: PCG preconditioned by p-multigrid (H1 orders 1, 2, 3 on the curved
    mesh, Chebyshev smoothers) with the native algebraic V-cycle on the assembled order-1 level, where the reference calls
    BoomerAMG (linalg/amg.cpp; ksp.cpp:153-157).  Same capacitance matrix (terminal-C.csv, 1e-6) in a fraction of the
    Jacobi-PCG iterations."""
    from palace_amd import ceed, linalg
    from palace_amd.fem import tet

    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "spheres_mesh.npz"))
    nodes, en = d["nodes"], d["elem_nodes"].astype(np.int64)
    used, inv = np.unique(en[:, :4], return_inverse=True)
    mesh = tet.TetMesh(nodes[used], inv.reshape(-1, 4), d["attr"])
    orders = [1, 2, 3]
    h1s = [tet.H1TetSpace(mesh, q) for q in orders]
    pts, wts = tet.default_tet_rule(orders[-1])
    G = tet.H1TetElement(3).tables(pts)[1]
    geom = ceed.DenseGeomFactorData(en, nodes, mesh.attr, G, wts)
    blocks = []
    for s in h1s:
        interp, grad = s.elem.tables(pts)
        blocks.append(ceed.DenseBlock(ceed.FE_H1, s.ndofs, s.offsets, interp, grad))
    fine = ceed.Operator(h1s[-1].ndofs, h1s[-1].ndofs).add_dense_integrator(geom, blocks[-1], ceed.QF_HCURL_33,
                                                                           ceed.coefficient_context(3), ceed.EVAL_GRAD).finalize()
    local = [fine.coarsen_dense(b) for b in blocks[:-1]] + [fine]
    bt = np.sort(np.searchsorted(used, d["bdr_tris"].astype(np.int64)), axis=1)
    fkey = {tuple(f): i for i, f in enumerate(map(tuple, mesh.face_verts))}
    fm = {}
    for a in (2, 3, 4):
        m = np.zeros(mesh.face_verts.shape[0], dtype=bool)
        m[[fkey[tuple(f)] for f in bt[d["bdr_attr"] == a]]] = True
        fm[a] = m
    all_m = fm[2] | fm[3] | fm[4]
    ess = [s.ess_dofs(all_m).astype(np.int32) for s in h1s]
    ctx = linalg.Context()
    A = [linalg.ParOperator(ctx, op, e, linalg.DIAG_ONE) for op, e in zip(local, ess)]
    csr0 = local[0].full_assemble_device()
    A[0] = linalg.AssembledParOperator(ctx, csr0, ess[0], linalg.DIAG_ONE)
    P = [linalg.DenseInterp(ctx, h1s[l].restriction(), h1s[l + 1].restriction(), tet.h1_tet_transfer_matrix(orders[l], orders[l + 1]))
         for l in range(2)]
    B = linalg.gmg(ctx, A, P, linalg.amg(ctx, csr0, ess[0]), cheby_order=6)
    solver = linalg.cg(ctx, A[-1], B, rel_tol=1e-13, max_it=300)
    jac = linalg.cg(ctx, A[-1], linalg.jacobi(ctx, A[-1]), rel_tol=1e-13, max_it=5000)
    n = h1s[-1].ndofs
    phi, its, its_jac = [], [], []
    for a in (3, 4):
        v = torch.zeros(n, dtype=torch.float64, device="cuda")
        v[torch.from_numpy(h1s[-1].ess_dofs(fm[a]).astype(np.int64)).cuda()] = 1.0
        b = torch.zeros_like(v)
        A[-1].eliminate_rhs(v, b)
        x = torch.zeros_like(v)
        solver.mult(b, x)
        assert solver.stats()["converged"], solver.stats()
        its.append(solver.stats()["iterations"])
        jac.mult(b, torch.zeros_like(v))
        its_jac.append(jac.stats()["iterations"])
        phi.append(x)
    eps0 = 1.0 / (1.25663706127e-6 * 299792458.0 ** 2)
    t = torch.empty(n, dtype=torch.float64, device="cuda")
    C = np.zeros((2, 2))
    for i in range(2):
        fine.mult(phi[i], t)
        for j in range(2):
            C[j, i] = eps0 * 1.0e-2 * float(phi[j] @ t)
    ref = d["C_F"]
    assert np.abs(C - ref).max() < 1e-6 * np.abs(ref).max(), (C, ref)
    assert max(its) <= 40 and max(its) * 8 < min(its_jac), (its, its_jac)
    print(f"spheres p=3: PCG + p-MG + native AMG {its} iterations, Jacobi-PCG {its_jac}")
