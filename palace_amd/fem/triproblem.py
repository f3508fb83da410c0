"""The magnetostatic problem of a plane triangular mesh on the device: the 2-D counterpart of TetProblem for the reference's
examples/cavity2d/cavity2d_magnetostatic.json (BASELINE config 4's magnetostatic half on the reference's own case): singular
curl-curl system, PCG + p-multigrid with plain Chebyshev smoothers (iodata.cpp:533-564), the native AMS in its singular mode on
the assembled order-1 level (linalg/ams.cpp:28-30), inductance from the field energy (drivers/magnetostaticsolver.cpp:132-180:
M_ii = 2 U_i / I_i^2, with a unit surface current I = width of the source)."""
from __future__ import annotations

import numpy as np

from . import tri

MU0 = 1.25663706127e-6  # utils/constants.hpp:26


def magnetostatic_inductance(ctx, mesh: tri.TriMesh, bdr_verts, bdr_attr, source_attr, direction, order=2, rel_tol=1e-8,
                             max_it=100):
    """Returns dict(M11 [H], iterations, converged, rel_residual, x, Kx, spaces, ess, local): every operator apply, the
    p-multigrid cycle, the AMS cycle and the PCG loop run on the device through the C ABI."""
    import torch

    from .. import ceed, linalg

    orders = list(range(1, order + 1))
    spaces = [tri.NDTriSpace(mesh, q) for q in orders]
    pts, wts = tri.tri_quadrature(order + 1)
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
    blocks = []
    for s in spaces:
        interp, curl = s.elem.tables(pts)
        blocks.append(ceed.DenseBlock(ceed.FE_HCURL, s.ndofs, s.offsets, interp, curl, orients=s.orients))
    fine = ceed.Operator(spaces[-1].ndofs, spaces[-1].ndofs).add_dense_integrator(
        geom, blocks[-1], ceed.QF_L2_1, ceed.coefficient_context(1), ceed.EVAL_CURL | ceed.EVAL_WEIGHT).finalize()
    local = [fine.coarsen_dense(b) for b in blocks[:-1]] + [fine]
    b, pec, width = tri.edge_current_load(spaces[-1], bdr_verts, bdr_attr, source_attr, direction)
    ess = [s.ess_dofs(pec) for s in spaces]
    A = [linalg.ParOperator(ctx, op, e, linalg.DIAG_ONE) for op, e in zip(local, ess)]
    A[0] = linalg.AssembledParOperator(ctx, local[0].full_assemble_device(), ess[0], linalg.DIAG_ONE)
    P = [linalg.DenseInterp(ctx, tri.restriction(spaces[l]), tri.restriction(spaces[l + 1]),
                            tri.nd_tri_transfer_matrix(orders[l], orders[l + 1])) for l in range(len(spaces) - 1)]
    h1 = tri.H1TriSpace(mesh, 1)
    coarse = linalg.ams(ctx, A[0].local, ess[0], tri.lowest_order_gradient(h1, spaces[0]), tri.vertex_coordinates(h1),
                        singular=True)
    B = linalg.gmg(ctx, A, P, coarse, cheby_order=max(2 * order, 4)) if len(A) > 1 else coarse
    K = linalg.cg(ctx, A[-1], B, rel_tol=rel_tol, max_it=max_it)
    b[ess[-1]] = 0.0
    bd = torch.from_numpy(b).cuda()
    x = torch.zeros_like(bd)
    K.mult(bd, x)
    st = K.stats()
    kx = torch.empty_like(x)
    A[-1].mult(x, kx)
    energy = float(x @ kx)  # = b^T K^+ b = 2 U / mu0 for the unit surface current
    return dict(M11=MU0 * energy / width ** 2, iterations=st["iterations"], converged=bool(st["converged"]),
                rel_residual=float((kx - bd).norm() / bd.norm()), x=x, Kx=kx, spaces=spaces, ess=ess, width=width,
                ndofs=spaces[-1].ndofs, keep=(geom, blocks, local, A, P, B, K))


def load_cavity2d(npz_path):
    """The reference's examples/cavity2d mesh (tri6) and boundary from the committed fixture (tests/golden/cavity2d_mesh.npz,
    written by tests/golden/make_golden.py from mesh/cavity2d.msh): (TriMesh, boundary vertex pairs, boundary attributes, dict)."""
    M_ = np.load(npz_path)
    en = M_["elem_nodes"].astype(np.int64)
    used, inv = np.unique(en[:, :3], return_inverse=True)
    mesh = tri.TriMesh(M_["nodes"][used], inv.reshape(-1, 3), M_["attr"], elem_nodes=en, nodes=M_["nodes"])
    bv = np.searchsorted(used, M_["bdr_edges"].astype(np.int64))
    return mesh, bv, M_["bdr_attr"], M_
