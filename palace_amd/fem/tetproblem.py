"""A PEC cavity problem on a tetrahedral mesh, all p-levels: spaces, dense-path operators, dense
interpolators — the tetrahedral counterpart of partition.SlabProblem.  With world > 1 the elements are partitioned by
recursive coordinate bisection (rcb.py; the reference: METIS parts of the serial mesh, utils/geodata.cpp:3587-3596) and every
operator, transfer and auxiliary space works on the rank's view with its halo plan."""
from __future__ import annotations

import numpy as np

from . import tet


class TetProblem:
    def __init__(self, ctx, mesh: tet.TetMesh, p: int, orders=None, rank=0, world=1):
        from .. import ceed, linalg

        self.ctx, self.mesh, self.p, self.rank, self.world = ctx, mesh, p, rank, world
        self.orders = list(range(1, p + 1)) if orders is None else list(orders)
        self.spaces = [tet.NDTetSpace(mesh, q) for q in self.orders]
        self.pts, self.wts = tet.default_tet_rule(p)  # every level integrates with the fine rule
        elems = np.arange(mesh.ne)
        self.part = None
        if world > 1:
            from .rcb import PartitionedSpace, rcb

            self.part = rcb(mesh.nodes[mesh.elem_nodes[:, :4]].mean(axis=1), world)
            self.spaces = [PartitionedSpace(s, self.part, rank, world) for s in self.spaces]
            elems = self.spaces[0].elems
        self.geom = ceed.DenseGeomFactorData(mesh.elem_nodes[elems], mesh.nodes, mesh.attr[elems],
                                             mesh.geometry_grad_table(self.pts), self.wts)
        self.ess = [s.ess_dofs() for s in self.spaces]
        self.n_true = [getattr(s, "n_true", s.ndofs) for s in self.spaces]
        self.halos = [linalg.Halo(ctx, s.nbr, s.send, s.recv) if world > 1 else None for s in self.spaces]
        self._keep = []

    def nd_block(self, s):
        from .. import ceed

        interp, curl = s.elem.tables(self.pts)
        kw = dict(orients=s.orients) if s.diagonal_transform else dict(curl_orients=s.curl_orients)
        return ceed.DenseBlock(ceed.FE_HCURL, s.ndofs, s.offsets, interp, curl, **kw)

    def h1_block(self, s):
        from .. import ceed

        interp, grad = s.elem.tables(self.pts)
        return ceed.DenseBlock(ceed.FE_H1, s.ndofs, s.offsets, interp, grad)

    def pcg_gmg_solver(self, max_it=50, rel_tol=0.0, eps_r=2.08, coarse_tol=1e-2, coarse_max_it=8, hiptmair=False,
                       coarse="cg", coarse_assembled=True):
        """Same configuration as SlabProblem.pcg_gmg_solver (reference iodata.cpp:519-564)."""
        import torch

        from .. import ceed, linalg

        ctx = self.ctx
        mass = ceed.coefficient_context(3, attr_mat=[0] * int(self.mesh.attr.max()), mat_coeff=[np.array([eps_r])])
        curl = ceed.coefficient_context(3)
        blocks = [self.nd_block(s) for s in self.spaces]
        fine = ceed.Operator(self.spaces[-1].ndofs, self.spaces[-1].ndofs).add_dense_integrator(
            self.geom, blocks[-1], ceed.QF_HDIVMASS_33, np.concatenate([mass, curl]),
            ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize()
        local = [fine.coarsen_dense(b) for b in blocks[:-1]] + [fine]
        A = [linalg.ParOperator(ctx, op, e, linalg.DIAG_ONE, n_true=nt, halo=h)
             for op, e, nt, h in zip(local, self.ess, self.n_true, self.halos)]
        if coarse_assembled and len(A) > 1:  # coarsest level as a device CSR matrix (rap.cpp:84-152)
            A[0] = linalg.AssembledParOperator(ctx, local[0].full_assemble_device(), self.ess[0], linalg.DIAG_ONE,
                                               n_true=self.n_true[0], halo=self.halos[0])
        P = [linalg.DenseInterp(ctx, self.spaces[l].restriction(), self.spaces[l + 1].restriction(interp_range=True),
                                tet.nd_tet_transfer_matrix(self.orders[l], self.orders[l + 1]), dom_halo=self.halos[l],
                                n_true_dom=self.n_true[l], n_true_rng=self.n_true[l + 1])
             for l in range(len(A) - 1)]
        aux = {}
        if hiptmair:
            h1s = [tet.H1TetSpace(self.mesh, q) for q in self.orders]
            if self.world > 1:
                from .rcb import PartitionedSpace

                h1s = [PartitionedSpace(s, self.part, self.rank, self.world) for s in h1s]
            h1_nt = [getattr(s, "n_true", s.ndofs) for s in h1s]
            h1_halos = [linalg.Halo(ctx, s.nbr, s.send, s.recv) if self.world > 1 else None for s in h1s]
            hb = [self.h1_block(s) for s in h1s]
            fine_h1 = ceed.Operator(h1s[-1].ndofs, h1s[-1].ndofs).add_dense_integrator(
                self.geom, hb[-1], ceed.QF_HCURL_33, mass, ceed.EVAL_GRAD).finalize()
            loc_h1 = [fine_h1.coarsen_dense(b) for b in hb[:-1]] + [fine_h1]
            A_h1 = [linalg.ParOperator(ctx, op, s.ess_dofs(), linalg.DIAG_ONE, n_true=nt, halo=h)
                    for op, s, nt, h in zip(loc_h1, h1s, h1_nt, h1_halos)]
            G = [linalg.DenseInterp(ctx, h.restriction(), n.restriction(interp_range=True), tet.tet_gradient_matrix(q),
                                    dom_halo=hh, n_true_dom=hn, n_true_rng=nn)
                 for h, n, q, hh, hn, nn in zip(h1s, self.spaces, self.orders, h1_halos, h1_nt, self.n_true)]
            aux = dict(A_aux=A_h1, G=G)
            self._keep.append((h1s, loc_h1, hb, h1_halos))
        if len(A) > 1:
            # level 0: the reference calls AMS (HYPRE) here.  Stand-ins: "cg" = a few Jacobi-PCG iterations (needed by
            # the auxiliary-space configuration, where level 0 must really reduce the error), "chebyshev" = a fixed
            # Chebyshev-Jacobi smoother of order 4 (better with the plain smoother on the cylinder: an inexact inner CG is
            # a nonlinear preconditioner and costs the outer PCG 40 % more iterations; scripts/coarse_tune.py)
            if coarse == "chebyshev":
                csolver = linalg.chebyshev(ctx, A[0], 4)
            elif coarse == "ams" and self.world == 1:  # the native auxiliary-space cycle on the assembled order-1 level
                assert coarse_assembled and self.orders[0] == 1
                h1_0 = tet.H1TetSpace(self.mesh, 1)
                csolver = linalg.ams(ctx, A[0].local, self.ess[0], tet.lowest_order_gradient(h1_0, self.spaces[0]),
                                     tet.vertex_coordinates(h1_0))
            elif coarse == "ams":
                # several ranks: the order-1 problem is solved redundantly by every rank (linalg.replicated): the GLOBAL level-0
                # matrix is assembled here from the serial mesh every rank holds, the same AMS solver is built on it everywhere,
                # and the distributed right-hand side is gathered through a halo plan on the global-numbered vector
                assert self.orders[0] == 1
                ps = self.spaces[0]
                g0 = ps.space
                geom_g = ceed.DenseGeomFactorData(self.mesh.elem_nodes, self.mesh.nodes, self.mesh.attr,
                                                  self.mesh.geometry_grad_table(self.pts), self.wts)
                op_g = ceed.Operator(g0.ndofs, g0.ndofs).add_dense_integrator(
                    geom_g, self.nd_block(g0), ceed.QF_HDIVMASS_33, np.concatenate([mass, curl]),
                    ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize()
                csr_g = op_g.full_assemble_device()
                h1_g = tet.H1TetSpace(self.mesh, 1)
                inner = linalg.ams(ctx, csr_g, g0.ess_dofs(), tet.lowest_order_gradient(h1_g, g0), tet.vertex_coordinates(h1_g))
                mine = ps.l2g[: ps.n_true].astype(np.int32)
                others = [q for q in range(self.world) if q != self.rank]
                gather = linalg.Halo(ctx, others, [mine for _ in others],
                                     [np.nonzero(ps.owner == q)[0].astype(np.int32) for q in others])
                csolver = linalg.replicated(ctx, gather, inner, mine, g0.ndofs)
                self._keep.append((geom_g, op_g, csr_g, inner, gather))
            else:
                csolver = linalg.cg(ctx, A[0], linalg.jacobi(ctx, A[0]), rel_tol=coarse_tol, max_it=coarse_max_it)
            B = linalg.gmg(ctx, A, P, csolver, cheby_order=max(2 * self.p, 4), **aux)
        else:
            B = linalg.jacobi(ctx, A[0])
        K = linalg.cg(ctx, A[-1], B, rel_tol=rel_tol, max_it=max_it)
        n = self.n_true[-1]
        ones = torch.ones(n, dtype=torch.float64, device="cuda")
        b = torch.empty_like(ones)
        A[-1].mult(ones, b)
        b[torch.from_numpy(self.ess[-1].astype(np.int64)).cuda()] = 0.0
        x = torch.zeros_like(b)
        self._keep.append((blocks, local, A, P, B))
        self.A = A
        return K, b, x

    def driven_solver(self, pec_faces, k0, eps, tand, coarse="ams", rel_tol=1e-8, max_it=400, restart=100, cheby_order=4,
                      coarse_tol=1e-3, coarse_max_it=200):
        """The driven-type complex system of BASELINE config 3 on this mesh and its solver, configured as the reference does for
        frequency-domain problems (models/spaceoperator.cpp:316-331, linalg/ksp.cpp): A = K - k0^2 eps_r (1 - i tan d) M with
        PEC on the faces `pec_faces` (bool mask over mesh.face_verts), FGMRES preconditioned by the Hiptmair p-multigrid of the
        shifted real matrix K + k0^2 eps_r M applied to both parts; level 0: `coarse` = "ams" (the native auxiliary-space cycle
        on the assembled order-1 matrix, where the reference calls HYPRE's AMS) or "cg" (Jacobi-PCG to coarse_tol).  eps, tand:
        per attribute (1-based attribute a -> entry a - 1).  One rank.  Returns dict(A, solver, ess, n)."""
        from .. import ceed, linalg

        assert self.world == 1
        ctx, mesh = self.ctx, self.mesh
        eps, tand = np.asarray(eps, dtype=np.float64), np.asarray(tand, dtype=np.float64)
        amap = list(range(len(eps)))

        def coef(vals):
            return ceed.coefficient_context(3, attr_mat=amap, mat_coeff=[np.array([v]) for v in vals])

        ident = ceed.coefficient_context(3)
        ess = [s.ess_dofs(pec_faces) for s in self.spaces]
        blocks = [self.nd_block(s) for s in self.spaces]

        def nd_op(block, qf, blob, ops):
            return ceed.Operator(block.lsize, block.lsize).add_dense_integrator(self.geom, block, qf, blob, ops).finalize()

        Kr = nd_op(blocks[-1], ceed.QF_HDIVMASS_33, np.concatenate([coef(-k0 ** 2 * eps), ident]), ceed.EVAL_CURL | ceed.EVAL_INTERP)
        Ki = nd_op(blocks[-1], ceed.QF_HCURL_33, coef(k0 ** 2 * eps * tand), ceed.EVAL_INTERP)
        A = linalg.ComplexParOperator(ctx, Kr, Ki, ess[-1], linalg.DIAG_ONE)
        pfine = nd_op(blocks[-1], ceed.QF_HDIVMASS_33, np.concatenate([coef(k0 ** 2 * eps), ident]), ceed.EVAL_CURL | ceed.EVAL_INTERP)
        ploc = [pfine.coarsen_dense(b) for b in blocks[:-1]] + [pfine]
        Pm = [linalg.ParOperator(ctx, o, es, linalg.DIAG_ONE) for o, es in zip(ploc, ess)]
        h1s = [tet.H1TetSpace(mesh, q) for q in self.orders]
        hb = [self.h1_block(s) for s in h1s]
        hfine = ceed.Operator(h1s[-1].ndofs, h1s[-1].ndofs).add_dense_integrator(self.geom, hb[-1], ceed.QF_HCURL_33,
                                                                                coef(k0 ** 2 * eps), ceed.EVAL_GRAD).finalize()
        hloc = [hfine.coarsen_dense(b) for b in hb[:-1]] + [hfine]
        Ph = [linalg.ParOperator(ctx, o, s.ess_dofs(pec_faces), linalg.DIAG_ONE) for o, s in zip(hloc, h1s)]
        G = [linalg.DenseInterp(ctx, h.restriction(), s.restriction(interp_range=True), tet.tet_gradient_matrix(q))
             for h, s, q in zip(h1s, self.spaces, self.orders)]
        P = [linalg.DenseInterp(ctx, self.spaces[l].restriction(), self.spaces[l + 1].restriction(interp_range=True),
                                tet.nd_tet_transfer_matrix(self.orders[l], self.orders[l + 1])) for l in range(len(Pm) - 1)]
        csr0 = None
        if coarse == "ams":
            assert self.orders[0] == 1
            csr0 = ploc[0].full_assemble_device()
            Pm[0] = linalg.AssembledParOperator(ctx, csr0, ess[0], linalg.DIAG_ONE)
            csolver = linalg.ams(ctx, csr0, ess[0], tet.lowest_order_gradient(h1s[0], self.spaces[0]), tet.vertex_coordinates(h1s[0]))
        else:
            csolver = linalg.cg(ctx, Pm[0], linalg.jacobi(ctx, Pm[0]), rel_tol=coarse_tol, max_it=coarse_max_it)
        B = linalg.gmg(ctx, Pm, P, csolver, cheby_order=cheby_order, A_aux=Ph, G=G) if len(Pm) > 1 else csolver
        S = linalg.ComplexParGmres(ctx, A, B, rel_tol=rel_tol, max_it=max_it, restart=restart, flexible=True)
        self._keep.append((Kr, Ki, blocks, ploc, Pm, h1s, hb, hloc, Ph, G, P, csr0, csolver, B))
        return dict(A=A, solver=S, ess=ess[-1], n=self.spaces[-1].ndofs, Kr=Kr, Ki=Ki, B=B)
