"""A PEC cavity problem on a hexahedral mesh with the reference's FULL multigrid hierarchy: h-levels (the lowest order of the
p-sequence on every mesh of a uniform-refinement sequence) followed by the p-levels on the finest mesh --
ConstructFiniteElementSpaceHierarchy, fem/multigrid.hpp:77-123; the meshes of the sequence are kept as levels by
utils/geodata.cpp:426-460.  Operators: one partial assembly per mesh (its own geometry data; the quadrature follows the solution
order on every level), the p-levels of a mesh reuse its quadrature data (BilinearForm::Assemble(hierarchy),
bilinearform.cpp:153-201); prolongations: the refinement transfer between meshes (fespace.cpp:246-251), the p-interpolation on
one mesh."""
from __future__ import annotations

import numpy as np

from . import htransfer
from .fespace import H1HexSpace, NDHexSpace, lowest_order_gradient, vertex_coordinates
from .mesh import refine_uniform
from .partition import levels_for


class HpProblem:
    def __init__(self, ctx, coarse_mesh, h_levels: int, p: int):
        from .. import ceed

        self.ctx, self.p, self.h_levels = ctx, p, h_levels
        self.meshes = [coarse_mesh]
        for _ in range(h_levels):
            self.meshes.append(refine_uniform(self.meshes[-1]))
        self.orders_p = levels_for(p)
        # (mesh index, order) of every level, coarsest first
        self.levels = [(m, self.orders_p[0]) for m in range(h_levels)] + [(h_levels, q) for q in self.orders_p]
        self.spaces = [NDHexSpace(self.meshes[m], q) for m, q in self.levels]
        self.ess = [s.ess_dofs() for s in self.spaces]
        self.q1d = p + 1
        self.geoms = [ceed.GeomFactorData(m, self.q1d) for m in self.meshes]
        self._keep = []

    def _operators(self, make_fine, spaces):
        """One operator per level: assembled anew on the first level of every mesh, p-coarsened copies on the others."""
        ops = [None] * len(self.levels)
        for mi in range(len(self.meshes)):
            on_mesh = [l for l, (m, _) in enumerate(self.levels) if m == mi]
            top = make_fine(self.geoms[mi], spaces[on_mesh[-1]])
            ops[on_mesh[-1]] = top
            for l in on_mesh[:-1]:
                ops[l] = top.coarsen(self.geoms[mi], spaces[l])
        return ops

    def _transfers(self, spaces):
        from .. import linalg

        P = []
        for l in range(len(self.levels) - 1):
            if self.levels[l][0] == self.levels[l + 1][0]:
                P.append(linalg.Interp(self.ctx, spaces[l], spaces[l + 1]))
            else:
                P.append(linalg.RefinementTransfer(self.ctx, *htransfer.hex_refinement(spaces[l], spaces[l + 1])))
        return P

    def pcg_gmg_solver(self, max_it=100, rel_tol=1e-8, eps_r=2.08, hiptmair=False, coarse="ams", coarse_tol=1e-2, coarse_max_it=8):
        """PCG on K + M with the V-cycle over ALL levels, configured like SlabProblem.pcg_gmg_solver (iodata.cpp:519-564)."""
        import torch

        from .. import ceed, linalg

        ctx = self.ctx
        mass = ceed.coefficient_context(3, attr_mat=[0] * int(max(m.attr.max() for m in self.meshes)), mat_coeff=[np.array([eps_r])])
        curl = ceed.coefficient_context(3)
        local = self._operators(lambda g, s: ceed.curlcurlmass_operator(g, s, mass, curl), self.spaces)
        A = [linalg.ParOperator(ctx, op, e, linalg.DIAG_ONE) for op, e in zip(local, self.ess)]
        csr0 = local[0].full_assemble_device()
        A[0] = linalg.AssembledParOperator(ctx, csr0, self.ess[0], linalg.DIAG_ONE)
        P = self._transfers(self.spaces)
        aux = {}
        if hiptmair:
            h1s = [H1HexSpace(self.meshes[m], q) for m, q in self.levels]
            loc_h1 = self._operators(lambda g, s: ceed.diffusion_operator(g, s, mass), h1s)
            A_h1 = [linalg.ParOperator(ctx, op, s.ess_dofs(), linalg.DIAG_ONE) for op, s in zip(loc_h1, h1s)]
            G = [linalg.Gradient(ctx, h, n) for h, n in zip(h1s, self.spaces)]
            aux = dict(A_aux=A_h1, G=G)
            self._keep.append((h1s, loc_h1))
        if coarse == "ams":
            assert self.levels[0][1] == 1, "AMS: order-1 coarsest level"
            h1_0 = H1HexSpace(self.meshes[0], 1)
            csolver = linalg.ams(ctx, csr0, self.ess[0], lowest_order_gradient(h1_0, self.spaces[0]), vertex_coordinates(h1_0))
        elif coarse == "chebyshev":
            csolver = linalg.chebyshev(ctx, A[0], 4)
        else:
            csolver = linalg.cg(ctx, A[0], linalg.jacobi(ctx, A[0]), rel_tol=coarse_tol, max_it=coarse_max_it)
        B = linalg.gmg(ctx, A, P, csolver, cheby_order=max(2 * self.p, 4), **aux)
        K = linalg.cg(ctx, A[-1], B, rel_tol=rel_tol, max_it=max_it)
        n = self.spaces[-1].ndofs
        ones = torch.ones(n, dtype=torch.float64, device="cuda")
        b = torch.empty_like(ones)
        A[-1].mult(ones, b)
        b[torch.from_numpy(self.ess[-1].astype(np.int64)).cuda()] = 0.0
        x = torch.zeros_like(b)
        self._keep.append((local, A, P, B, csr0))
        self.A, self.P, self.last_gmg = A, P, B
        return K, b, x
