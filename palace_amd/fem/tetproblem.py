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


# ---- BASELINE config 3 as the reference defines it: examples/cpw/cpw_lumped_uniform.json -----------------------------------------
Z0_OHM = 376.730313668  # free-space impedance (utils/constants.hpp: sqrt(mu0 / eps0))
C0_M_S = 299792458.0

# "Domains" / "Boundaries" of examples/cpw/cpw_lumped_uniform.json (:16-131).  The sapphire tensors are given in the material axes
# [[0.8, 0.6, 0], [-0.6, 0.8, 0], [0, 0, 1]] -- a rotation about z of tensors whose x and y entries are equal: the same diagonal
# tensors in the mesh frame.
CPW_LUMPED_UNIFORM = dict(
    L0=1.0e-6,
    materials={1: dict(mu=[1.0, 1.0, 1.0], eps=[1.0, 1.0, 1.0], tand=[0.0, 0.0, 0.0]),
               2: dict(mu=[0.99999975, 0.99999975, 0.99999979], eps=[9.3, 9.3, 11.5], tand=[3.0e-5, 3.0e-5, 8.6e-5])},
    pec=(13,), absorbing=(4,),
    ports={1: dict(R=56.02, elements=[(5, +1.0), (9, -1.0)]), 2: dict(R=56.02, elements=[(6, +1.0), (10, -1.0)]),
           3: dict(R=56.02, elements=[(7, +1.0), (11, -1.0)]), 4: dict(R=56.02, elements=[(8, +1.0), (12, -1.0)])},
    port_axis=1)  # "Direction": "+Y" / "-Y"


class DrivenReferenceSystem:
    """A(omega) = K + i omega C - omega^2 M of a driven simulation as SpaceOperator assembles it (models/spaceoperator.cpp:270-326,
    :786-804) for a configuration like examples/cpw/cpw_lumped_uniform.json, in mesh length units (k0 = omega L0 / c0, impedances in
    units of Z0 -- the reference's nondimensionalisation with Lc = L0):
      K   curl-curl with mu^-1 (tensor)                                          CurlCurlIntegrator, f_apply_hdiv_33
      M   mass with eps (1 - i tan d) (tensors)                                  VectorFEMassIntegrator, f_apply_hcurl_33
      C   surface mass: 1 / Z of the adjacent material on the absorbing boundary (first order,
          farfieldboundaryoperator.cpp:94-106) and 1 / R_s on the lumped-port elements, R_s = R (W / L) n_elements
          (lumpedportoperator.cpp:584-604, lumpedportoperator.hpp:60-63)              f_apply_hcurl_32 on the boundary triangles
      real part  K - k0^2 Re M   (one fused curl-curl + mass operator, f_apply_hdivmass_33)
      imaginary  k0 C + k0^2 eps tan d M-type   (volume mass + surface masses, one ceed::Operator with several sub-operators)
      PEC on `pec` (essential, DIAG_ONE in the real part), natural elsewhere
      right-hand side of excitation e: i k0 int 2 H_inc (e_dir . v) dS over the excited port's elements,
          H_inc = 1 / sqrt(R_s W L n) (lumpedportoperator.cpp:628-662, spaceoperator.cpp:1296-1308)
      S_ij = int H_inc,i (e_dir . E) dS - delta_ij (lumpedportoperator.cpp:176-205 and the driven driver).
    Solver: FGMRES + Hiptmair p-multigrid of K + k0^2 Re(eps) M on both parts + native AMS on the assembled order-1 level, as
    TetProblem.driven_solver."""

    def __init__(self, prob: TetProblem, freq_ghz, cfg=CPW_LUMPED_UNIFORM, coarse="ams", rel_tol=1e-8, max_it=400, restart=None,
                 cheby_order=None, orthogonalization="MGS"):
        import torch

        from .. import ceed, linalg
        from . import tri

        assert prob.world == 1
        ctx, mesh, p = prob.ctx, prob.mesh, prob.p
        self.prob, self.cfg = prob, cfg
        self.k0 = k0 = 2.0 * np.pi * freq_ghz * 1.0e9 * cfg["L0"] / C0_M_S
        nattr = int(mesh.attr.max())
        mats = cfg["materials"]
        amap = [sorted(mats).index(a) if a in mats else -1 for a in range(1, nattr + 1)]
        order = sorted(mats)

        def vol_ctx(fn):
            return ceed.coefficient_context(3, attr_mat=amap, mat_coeff=[np.diag(np.asarray(fn(mats[a]), dtype=np.float64)) for a in order])

        mu_inv = vol_ctx(lambda m: 1.0 / np.asarray(m["mu"]))
        self._vol = dict(mu_inv=[np.diag(1.0 / np.asarray(mats[a]["mu"])) for a in order],
                         eps=[np.diag(np.asarray(mats[a]["eps"], dtype=np.float64)) for a in order],
                         eps_tand=[np.diag(np.asarray(mats[a]["eps"]) * np.asarray(mats[a]["tand"])) for a in order], amap=amap)
        # ---- boundary faces by attribute (internal boundaries included: every boundary triangle is one face of the mesh)
        bt = np.sort(np.asarray(mesh.bdr_tris, dtype=np.int64), axis=1)
        fv = mesh.face_verts
        key = lambda f: (f[:, 0] * mesh.nv + f[:, 1]) * mesh.nv + f[:, 2]  # noqa: E731
        of = np.argsort(key(fv))
        bface = of[np.searchsorted(key(fv)[of], key(bt))]
        battr = np.asarray(mesh.bdr_attr)
        fmask = np.zeros(fv.shape[0], dtype=bool)
        fmask[bface[np.isin(battr, cfg["pec"])]] = True
        self.pec_faces = fmask
        # element attribute next to each face (absorbing faces lie on the outer boundary: one neighbour)
        face_attr = np.zeros(fv.shape[0], dtype=np.int64)
        face_attr[mesh.elem_faces.ravel()] = np.repeat(mesh.attr, 4)
        # surface blocks: attribute 1 .. = absorbing next to material order[0], order[1], ..., then the port elements
        sfaces, sattr, scoef = [], [], []
        for k, a in enumerate(order):
            f = bface[np.isin(battr, cfg["absorbing"]) & (face_attr[bface] == a)]
            sfaces.append(f), sattr.append(np.full(f.size, k + 1))
            scoef.append(np.diag(np.sqrt(np.asarray(mats[a]["eps"]) / np.asarray(mats[a]["mu"]))))  # GetInvImpedance: sqrt(mu^-1 eps)
        self.port_elems = {}
        nabs = len(order)
        for idx, port in sorted(cfg["ports"].items()):
            n_el = len(port["elements"])
            for battr_e, sign in port["elements"]:
                f = bface[battr == battr_e]
                pts = mesh.verts[np.unique(fv[f])]
                ext = pts.max(axis=0) - pts.min(axis=0)
                ax = cfg["port_axis"]
                length = ext[ax]
                width = max(ext[b] for b in range(3) if b != ax)
                Rs = port["R"] / Z0_OHM * (width / length) * n_el
                hinc = 1.0 / np.sqrt(Rs * width * length * n_el)
                k = nabs + len(self.port_elems) + 1
                self.port_elems[(idx, battr_e)] = dict(attr=k, sign=sign, Rs=Rs, hinc=hinc, width=width, length=length)
                sfaces.append(f), sattr.append(np.full(f.size, k)), scoef.append(np.array([1.0 / Rs]))
        sfaces, sattr = np.concatenate(sfaces), np.concatenate(sattr).astype(np.int32)
        self.sfaces, self.sattr, self.scoef = sfaces, sattr, scoef
        nd = prob.spaces[-1]
        self.nd, self.n = nd, nd.ndofs
        self.spts, self.swts = tri.tri_quadrature(p + 1)
        self.sblk = tet.NDTetBoundaryBlock(nd, sfaces, sattr)
        sint, scurl = self.sblk.elem.tables(self.spts)
        self.sgeom = ceed.DenseGeomFactorData(self.sblk.elem_nodes, self.sblk.nodes, self.sblk.attr,
                                              self.sblk.geometry_grad_table(self.spts), self.swts)
        self.sblock = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, self.sblk.offsets, sint, None, orients=self.sblk.orients)
        ns = len(scoef)

        def surf_ctx(vals):  # one "material" per surface attribute
            return ceed.coefficient_context(3, attr_mat=list(range(ns)), mat_coeff=vals)

        # ---- operators
        ess = [s.ess_dofs(fmask) for s in prob.spaces]
        self.ess = ess[-1]
        blocks = [prob.nd_block(s) for s in prob.spaces]

        def nd_op(block, qf, blob, ops):
            return ceed.Operator(block.lsize, block.lsize).add_dense_integrator(prob.geom, block, qf, blob, ops).finalize()

        self.Ar = nd_op(blocks[-1], ceed.QF_HDIVMASS_33, np.concatenate([vol_ctx(lambda m: -k0 ** 2 * np.asarray(m["eps"])), mu_inv]),
                        ceed.EVAL_CURL | ceed.EVAL_INTERP)
        self.Ai = (ceed.Operator(nd.ndofs, nd.ndofs)
                   .add_dense_integrator(prob.geom, blocks[-1], ceed.QF_HCURL_33,
                                         vol_ctx(lambda m: k0 ** 2 * np.asarray(m["eps"]) * np.asarray(m["tand"])), ceed.EVAL_INTERP)
                   .add_dense_integrator(self.sgeom, self.sblock, ceed.QF_HCURL_32, surf_ctx([k0 * c for c in scoef]), ceed.EVAL_INTERP)
                   .finalize())
        self.A = linalg.ComplexParOperator(ctx, self.Ar, self.Ai, ess[-1], linalg.DIAG_ONE)
        # ---- preconditioner: Hiptmair p-multigrid of the real "sum of magnitudes" matrix K + k0 C + k0^2 Re(eps) M the reference
        # assembles for driven problems (GetPreconditionerMatrix, spaceoperator.cpp:786-804 with pc_mat_real: the damping terms --
        # here the surface masses of the absorbing boundary and the ports -- belong to it: on gradient fields they are orders of
        # magnitude larger than k0^2 eps M at GHz frequencies on a mm-sized chip, and without them FGMRES stalls), AMS on level 0.
        # Every level gets its own volume + surface sub-operators (tables of the level's elements at the fine rules); the
        # auxiliary H1 operators likewise (AddAuxIntegrators: diffusion with the mass coefficient in the volume and with the
        # damping coefficient on the surfaces = G^T (.) G of the two Nedelec mass terms).
        cpos = vol_ctx(lambda m: k0 ** 2 * np.asarray(m["eps"]))
        sdamp = surf_ctx([k0 * c for c in scoef])
        ploc = []
        for s_, b_ in zip(prob.spaces, blocks):
            sb = tet.NDTetBoundaryBlock(s_, sfaces, sattr)
            si_, _ = sb.elem.tables(self.spts)
            sblock_l = ceed.DenseBlock(ceed.FE_HCURL, s_.ndofs, sb.offsets, si_, None, orients=sb.orients)
            ploc.append(ceed.Operator(s_.ndofs, s_.ndofs)
                        .add_dense_integrator(prob.geom, b_, ceed.QF_HDIVMASS_33, np.concatenate([cpos, mu_inv]), ceed.EVAL_CURL | ceed.EVAL_INTERP)
                        .add_dense_integrator(self.sgeom, sblock_l, ceed.QF_HCURL_32, sdamp, ceed.EVAL_INTERP).finalize())
        Pm = [linalg.ParOperator(ctx, o, es, linalg.DIAG_ONE) for o, es in zip(ploc, ess)]
        h1s = [tet.H1TetSpace(mesh, q) for q in prob.orders]
        hb = [prob.h1_block(s) for s in h1s]
        hloc = []
        for h_, b_ in zip(h1s, hb):
            sb = tet.H1TetBoundaryBlock(h_, sfaces, sattr)
            hi_, hg_ = sb.elem.tables(self.spts)
            hblock_l = ceed.DenseBlock(ceed.FE_H1, h_.ndofs, sb.offsets, hi_, hg_)
            hloc.append(ceed.Operator(h_.ndofs, h_.ndofs)
                        .add_dense_integrator(prob.geom, b_, ceed.QF_HCURL_33, cpos, ceed.EVAL_GRAD)
                        .add_dense_integrator(self.sgeom, hblock_l, ceed.QF_HCURL_32, sdamp, ceed.EVAL_GRAD).finalize())
        Ph = [linalg.ParOperator(ctx, o, s.ess_dofs(fmask), linalg.DIAG_ONE) for o, s in zip(hloc, h1s)]
        G = [linalg.DenseInterp(ctx, h.restriction(), s.restriction(interp_range=True), tet.tet_gradient_matrix(q))
             for h, s, q in zip(h1s, prob.spaces, prob.orders)]
        P = [linalg.DenseInterp(ctx, prob.spaces[l].restriction(), prob.spaces[l + 1].restriction(interp_range=True),
                                tet.nd_tet_transfer_matrix(prob.orders[l], prob.orders[l + 1])) for l in range(len(Pm) - 1)]
        csr0 = None
        if coarse == "ams" and len(Pm) > 1:
            csr0 = ploc[0].full_assemble_device()
            Pm[0] = linalg.AssembledParOperator(ctx, csr0, ess[0], linalg.DIAG_ONE)
            csolver = linalg.ams(ctx, csr0, ess[0], tet.lowest_order_gradient(h1s[0], prob.spaces[0]), tet.vertex_coordinates(h1s[0]))
        else:
            csolver = linalg.cg(ctx, Pm[0], linalg.jacobi(ctx, Pm[0]), rel_tol=1e-3, max_it=200)
        ko = max(2 * p, 4) if cheby_order is None else cheby_order
        self.B = linalg.gmg(ctx, Pm, P, csolver, cheby_order=ko, A_aux=Ph, G=G) if len(Pm) > 1 else csolver
        self.solver = linalg.ComplexParGmres(ctx, self.A, self.B, rel_tol=rel_tol, max_it=max_it,
                                             restart=max_it if restart is None else restart, flexible=True,
                                             orthogonalization=orthogonalization)
        self._keep = (blocks, ploc, Pm, h1s, hb, hloc, Ph, G, P, csr0, csolver, sdamp)
        # ---- port forms: f_i = int H_inc (e_dir . v) dS = (surface mass with coefficient sign * H_inc on port i's elements) E_dir
        ax = cfg["port_axis"]
        edir = nd.interpolate(lambda X: np.broadcast_to(np.eye(3)[ax], X.shape))
        ed = torch.from_numpy(edir).cuda()
        self.port_form = {}
        for idx in sorted(cfg["ports"]):
            vals = [np.array([0.0])] * ns
            for (pi, _), e in self.port_elems.items():
                if pi == idx:
                    vals[e["attr"] - 1] = np.array([e["sign"] * e["hinc"]])
            op = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(self.sgeom, self.sblock, ceed.QF_HCURL_32, surf_ctx(vals),
                                                                        ceed.EVAL_INTERP).finalize()
            f = torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")
            op.mult(ed, f)
            self.port_form[idx] = f
            del op
        self._ess_t = torch.from_numpy(self.ess.astype(np.int64)).cuda()

    def excitation(self, port_idx):
        """(b_r, b_i) = i k0 * 2 f_port with the essential rows zero (spaceoperator.cpp:1296-1308)."""
        import torch

        bi = 2.0 * self.k0 * self.port_form[port_idx]
        bi[self._ess_t] = 0.0
        return torch.zeros_like(bi), bi

    def s_parameters(self, xr, xi, excited):
        """S_j,excited for every port j from the solution (E_r, E_i)."""
        out = {}
        for j, f in self.port_form.items():
            s = complex(float(f @ xr), float(f @ xi))
            out[j] = s - (1.0 if j == excited else 0.0)
        return out
