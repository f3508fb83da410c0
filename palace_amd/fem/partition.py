"""Element partition of the cylinder across GPUs (one z-slab per rank) and the halo plans that
implement the conforming prolongation P / P^T of every multigrid level.

Reference: Palace partitions elements with METIS (utils/geodata.cpp:3587-3596), each rank's
L-vector holds every dof of its elements, true dofs have one owner and `y = P^T A_local P x`
(linalg/rap.cpp:195-234); P / P^T live in MFEM's ParFiniteElementSpace (not vendored).  Here the
O-grid cylinder is cut into slabs of whole element layers: rank r owns layers [r nz, (r+1) nz) and
the dofs on its top interface plane; the dofs on its bottom plane are ghosts owned by rank r-1.
Local numbering = owned dofs first, ghosts last, so a T-vector is a prefix of the L-vector.

Both sides of an interface enumerate the shared entities in the same order without communicating:
entities in the plane are sorted by the lattice key of their (x, y) midpoint, which both ranks
compute from bit-identical cross-section coordinates; edge direction / face frame on the plane are
fixed by vertex ids whose relative order within a plane is the same on both ranks (see
`_merge_points`: ids follow the (x, y, z) lattice order).
"""
from __future__ import annotations

import numpy as np

from .fespace import H1HexSpace, NDHexSpace
from .mesh import HexMesh, ogrid_cylinder


def slab_shape(dofs_per_rank: float, p: int):
    """(n, nz) of one slab with about dofs_per_rank Nedelec dofs of order p (5 n^2 nz elements)."""
    ne = dofs_per_rank / (3.0 * p**3)
    n = max(1, int(round((ne / (5 * 1.15)) ** (1.0 / 3.0))))
    nz = max(1, int(round(ne / (5 * n * n))))
    return n, nz


def strong_shape(total_dofs: float, p: int):
    """(n, nz) of the one global cylinder of a strong-scaling run: the layer count is a multiple of 8, so that
    1, 2, 4 and 8 ranks cut the very same mesh into equal z-slabs."""
    n, nz = slab_shape(total_dofs, p)
    return n, max(8, 8 * ((nz + 4) // 8))


def _plane_key(xy: np.ndarray, scale: float) -> np.ndarray:
    q = np.round(xy * scale).astype(np.int64) + (1 << 30)
    return (q[:, 0] << 32) | q[:, 1]


class SlabNDSpace(NDHexSpace):
    """Nedelec space on one slab, renumbered owned-first, with its halo lists.

    n_true, ndofs (= n_local); halo: neighbours + send/recv index lists; ess(): essential TRUE dofs
    (PEC wall + the physical bottom/top caps of the whole cylinder)."""

    def __init__(self, mesh: HexMesh, p: int, rank: int, world: int, z_lo: float, z_hi: float, radius: float):
        super().__init__(mesh, p)
        self.rank, self.world = rank, world
        tol = 1e-6 * radius
        vz = mesh.vert_coords[:, 2]
        scale = 1.0 / (1e-6 * radius)

        def plane_dofs(z):
            """Dofs living in the plane z = const, in the canonical cross-rank order."""
            inp = np.abs(vz - z) < tol
            ev, fv = mesh.edge_verts, mesh.face_verts
            e_in = np.nonzero(inp[ev].all(axis=1))[0]
            f_in = np.nonzero(inp[fv].all(axis=1))[0]
            vc = mesh.vert_coords[:, :2]
            e_in = e_in[np.argsort(_plane_key(vc[ev[e_in]].mean(axis=1), scale), kind="stable")]
            f_in = f_in[np.argsort(_plane_key(vc[fv[f_in]].mean(axis=1), scale), kind="stable")]
            n_e, n_f, _ = self.n_per
            de = (self.edge_base + e_in[:, None] * n_e + np.arange(n_e)[None, :]).ravel()
            df = (self.face_base + f_in[:, None] * n_f + np.arange(n_f)[None, :]).ravel()
            return np.concatenate([de, df]).astype(np.int64), f_in

        bottom, f_bot = plane_dofs(z_lo)
        top, f_top = plane_dofs(z_hi)
        n_old = self.ndofs
        ghosts = bottom if rank > 0 else np.zeros(0, dtype=np.int64)
        is_ghost = np.zeros(n_old, dtype=bool)
        is_ghost[ghosts] = True
        perm = np.empty(n_old, dtype=np.int64)
        self.n_true = n_old - ghosts.size
        perm[~is_ghost] = np.arange(self.n_true)
        perm[ghosts] = self.n_true + np.arange(ghosts.size)  # ghost slots in canonical order
        self.elem_dof_lex = perm[self.elem_dof_lex].astype(np.int32)
        self._perm = perm
        # essential dofs: boundary faces minus the interface planes
        fmask = mesh.boundary_face_mask.copy()
        if rank > 0:
            fmask[f_bot] = False
        if rank < world - 1:
            fmask[f_top] = False
        ess = super().ess_dofs(face_mask=fmask)
        self._ess_true = ess[ess < self.n_true]
        # halo plan
        self.nbr, self.send, self.recv = [], [], []
        if rank > 0:
            self.nbr.append(rank - 1)
            self.send.append(np.zeros(0, dtype=np.int32))
            self.recv.append(perm[bottom].astype(np.int32))
        if rank < world - 1:
            self.nbr.append(rank + 1)
            self.send.append(perm[top].astype(np.int32))
            self.recv.append(np.zeros(0, dtype=np.int32))

    def ess_dofs(self, face_mask=None):
        return self._ess_true


class SlabH1Space(H1HexSpace):
    """H1 space on one slab (the auxiliary space of the Hiptmair smoother), renumbered owned-first,
    with its halo lists; essential dofs = all boundary nodes of the whole cylinder."""

    def __init__(self, mesh: HexMesh, p: int, rank: int, world: int, z_lo: float, z_hi: float, radius: float):
        super().__init__(mesh, p)
        self.rank, self.world = rank, world
        tol = 1e-6 * radius
        vc = mesh.vert_coords
        scale = 1.0 / (1e-6 * radius)
        n_e, n_f = p - 1, (p - 1) ** 2

        def plane_dofs(z):
            inp = np.abs(vc[:, 2] - z) < tol
            ev, fv = mesh.edge_verts, mesh.face_verts
            v_in = np.nonzero(inp)[0]
            e_in = np.nonzero(inp[ev].all(axis=1))[0]
            f_in = np.nonzero(inp[fv].all(axis=1))[0]
            v_in = v_in[np.argsort(_plane_key(vc[v_in, :2], scale), kind="stable")]
            e_in = e_in[np.argsort(_plane_key(vc[ev[e_in], :2].mean(axis=1), scale), kind="stable")]
            f_in = f_in[np.argsort(_plane_key(vc[fv[f_in], :2].mean(axis=1), scale), kind="stable")]
            de = (self.edge_base + e_in[:, None] * n_e + np.arange(n_e)[None, :]).ravel()
            df = (self.face_base + f_in[:, None] * n_f + np.arange(n_f)[None, :]).ravel()
            return np.concatenate([self.vert_base + v_in, de, df]).astype(np.int64), f_in

        bottom, f_bot = plane_dofs(z_lo)
        top, f_top = plane_dofs(z_hi)
        n_old = self.ndofs
        ghosts = bottom if rank > 0 else np.zeros(0, dtype=np.int64)
        is_ghost = np.zeros(n_old, dtype=bool)
        is_ghost[ghosts] = True
        perm = np.empty(n_old, dtype=np.int64)
        self.n_true = n_old - ghosts.size
        perm[~is_ghost] = np.arange(self.n_true)
        perm[ghosts] = self.n_true + np.arange(ghosts.size)
        self.elem_dof_lex = perm[self.elem_dof_lex].astype(np.int32)
        fmask = mesh.boundary_face_mask.copy()
        if rank > 0:
            fmask[f_bot] = False
        if rank < world - 1:
            fmask[f_top] = False
        ess = self._ess_with_mask(fmask)
        self._ess_true = ess[ess < self.n_true]
        self.nbr, self.send, self.recv = [], [], []
        if rank > 0:
            self.nbr.append(rank - 1)
            self.send.append(np.zeros(0, dtype=np.int32))
            self.recv.append(perm[bottom].astype(np.int32))
        if rank < world - 1:
            self.nbr.append(rank + 1)
            self.send.append(perm[top].astype(np.int32))
            self.recv.append(np.zeros(0, dtype=np.int32))

    def _ess_with_mask(self, fmask):
        from .mesh import HEX_FACE_AXES

        mesh, p = self.mesh, self.p
        n1 = p + 1
        bmask = fmask[mesh.elem_faces]
        out = []
        for lf, (nax, side, uax, vax) in enumerate(HEX_FACE_AXES):
            el = np.nonzero(bmask[:, lf])[0]
            if el.size == 0:
                continue
            rng = [np.arange(n1)] * 3
            rng[nax] = np.array([side * p])
            K, J, I = np.meshgrid(rng[2], rng[1], rng[0], indexing="ij")
            lex = (I + n1 * (J + n1 * K)).ravel()
            out.append(self.elem_dof_lex[el][:, lex].ravel())
        return np.unique(np.concatenate(out)).astype(np.int32) if out else np.zeros(0, np.int32)

    def ess_dofs(self):
        return self._ess_true


def levels_for(p: int):
    """p-coarsening sequence of the reference (LOGARITHMIC: p -> (p + 1) / 2 down to 1,
    fem/multigrid.hpp:44-69)."""
    out = [p]
    while out[-1] > 1:
        out.append((out[-1] + 1) // 2)
    return out[::-1]


def global_edge_map(slab_space, global_space, slab_height: float, world: int):
    """Order-1 Nedelec dofs (edges) of one slab against those of the whole cylinder: for the slab's TRUE dofs the global dof number
    and the relative orientation (+-1: an edge dof points from the lower to the higher vertex number, and the two meshes number
    their vertices independently), and for every global dof its owning rank (the lowest slab that contains the edge, as in
    SlabNDSpace).  Matching is geometric: rounded coordinates of the edge's end points."""
    assert slab_space.p == 1 and global_space.p == 1
    ms, mg = slab_space.mesh, global_space.mesh
    scale = 1.0e6 / max(1.0, float(np.abs(mg.vert_coords).max()))

    def keys(vc):
        q = np.round(vc * scale).astype(np.int64) + (1 << 20)
        return (q[:, 0] << 42) | (q[:, 1] << 21) | q[:, 2]

    kg, ks = keys(mg.vert_coords), keys(ms.vert_coords)
    order = np.argsort(kg)
    pos = np.searchsorted(kg[order], ks)
    assert (kg[order][pos] == ks).all(), "a slab vertex is not a vertex of the global mesh"
    v_s2g = order[pos]  # slab vertex -> global vertex
    ev_s = ms.edge_verts  # [nedges, 2], lower vertex number first: the direction of the dof
    a, b = v_s2g[ev_s[:, 0]], v_s2g[ev_s[:, 1]]
    nvg = mg.vert_coords.shape[0]
    ekey_g = mg.edge_verts[:, 0].astype(np.int64) * nvg + mg.edge_verts[:, 1]
    eorder = np.argsort(ekey_g)
    ekey_s = np.minimum(a, b) * nvg + np.maximum(a, b)
    epos = np.searchsorted(ekey_g[eorder], ekey_s)
    assert (ekey_g[eorder][epos] == ekey_s).all(), "a slab edge is not an edge of the global mesh"
    e_s2g = eorder[epos]
    sgn = np.where(a < b, 1.0, -1.0)  # same direction as the global dof (lower global vertex first) or against it
    ldof = slab_space._perm[slab_space.edge_base + np.arange(ms.nedges)]  # (one dof per edge at order 1)
    true = ldof < slab_space.n_true
    mine = np.empty(slab_space.n_true, dtype=np.int32)
    sign = np.empty(slab_space.n_true, dtype=np.float64)
    mine[ldof[true]] = global_space.edge_base + e_s2g[true]
    sign[ldof[true]] = sgn[true]
    # owners: the lowest slab whose z-range contains both end points
    z = mg.vert_coords[:, 2]
    tol = 1e-6 * max(1.0, float(np.abs(z).max()))
    zlo = np.minimum(z[mg.edge_verts[:, 0]], z[mg.edge_verts[:, 1]])
    owner = np.clip(np.floor((zlo + tol) / slab_height).astype(np.int64), 0, world - 1)
    # an edge lying in the plane between slabs q - 1 and q (zlo = zhi = q H) belongs to the lower one
    zhi = np.maximum(z[mg.edge_verts[:, 0]], z[mg.edge_verts[:, 1]])
    in_plane = (np.abs(zhi - zlo) < tol) & (np.abs(zlo / slab_height - np.round(zlo / slab_height)) < 1e-6) & (owner > 0)
    on_boundary = np.abs(zlo - owner * slab_height) < tol
    owner[in_plane & on_boundary] -= 1
    full = np.zeros(global_space.ndofs, dtype=np.int64)
    full[global_space.edge_base + np.arange(mg.nedges)] = owner
    return mine, sign, full


class SlabProblem:
    """Everything bench.py / the multi-rank tests need on one rank: the slab mesh, the level spaces,
    device geometry data, operators, halo objects and solvers."""

    def __init__(self, ctx, rank, world, p, dofs_per_rank, levels=True, radius=2.74, shape=None, device=True):
        self.ctx, self.rank, self.world, self.p = ctx, rank, world, p
        n, nz = shape if shape is not None else slab_shape(dofs_per_rank, p)
        self.shape = (n, nz)
        h_layer = 2.0 * radius / max(1, round(1.15 * n))  # roughly isotropic elements
        self.height = nz * h_layer
        z_lo = rank * self.height
        mesh = ogrid_cylinder(n, nz, radius=radius, height=self.height)
        mesh.x = mesh.x.copy()
        mesh.x[:, 2] += z_lo
        self.mesh, self.radius = mesh, radius
        self.orders = levels_for(p) if levels else [p]
        self.spaces = [SlabNDSpace(mesh, q, rank, world, z_lo, z_lo + self.height, radius) for q in self.orders]
        self.n_true = [s.n_true for s in self.spaces]
        self.n_local = [s.ndofs for s in self.spaces]
        self.ess = [s.ess_dofs() for s in self.spaces]
        self.q1d = p + 1
        if device:
            self._device_setup()

    # ---- device objects ---------------------------------------------------------------------
    def _device_setup(self):
        from .. import ceed, linalg

        self.geom = ceed.GeomFactorData(self.mesh, self.q1d)
        self.halos = [linalg.Halo(self.ctx, s.nbr, s.send, s.recv) if self.world > 1 else None for s in self.spaces]
        self.local_curlcurl = ceed.curlcurl_operator(self.geom, self.spaces[-1], ceed.coefficient_context(3))
        self._keep = []

    def global_true_dofs(self):
        if self.world == 1:
            return self.n_true[-1]
        import torch
        import torch.distributed as dist

        t = torch.tensor([self.n_true[-1]], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t)
        return int(t.item())

    def curlcurl_par_operator(self):
        from .. import linalg

        return linalg.ParOperator(self.ctx, self.local_curlcurl, self.ess[-1], linalg.DIAG_ONE,
                                  n_true=self.n_true[-1], halo=self.halos[-1])

    def pcg_gmg_solver(self, max_it=50, rel_tol=0.0, eps_r=2.08, coarse_tol=1e-2, coarse_max_it=8, hiptmair=False,
                       coarse="cg", coarse_assembled=True, singular=False, level_rule="fine"):
        """PCG on (K + M) with the p-multigrid preconditioner configured as the reference does for
        p = 3 (iodata.cpp:533-564: 4th-kind Chebyshev of order max(2p, 4), 1 smoothing step, 1 V-cycle);
        level 0 is solved by Jacobi-PCG (the reference uses AMS from HYPRE there, linalg/ams.cpp).
        level_rule: "fine" -- the p-coarsened levels reuse the fine level's quadrature data, as the reference's
        CeedOperatorCoarsen does (fem/libceed/operator.cpp:528-546; a 54-dof order-2 element then streams 64 points); "own" --
        every level is assembled on the rule of its own order (p_l + 1 points per direction: 27 for order 2).  NOT the
        reference's behaviour: an option, measured beside it (a different, equally symmetric positive definite, preconditioner)."""
        import torch

        from .. import ceed, linalg

        ctx = self.ctx
        mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([eps_r])])
        curl = ceed.coefficient_context(3)
        fine = ceed.curlcurlmass_operator(self.geom, self.spaces[-1], mass, curl)
        if level_rule == "own":
            assert self.world == 1 and not hiptmair, "level_rule='own': one rank, plain smoothers"
            geoms = [ceed.GeomFactorData(self.mesh, s.p + 1) for s in self.spaces[:-1]]
            local = [ceed.curlcurlmass_operator(g, s, mass, curl) for g, s in zip(geoms, self.spaces[:-1])] + [fine]
            self._keep.append(geoms)
        else:
            local = [fine.coarsen(self.geom, s) for s in self.spaces[:-1]] + [fine]
        A = [linalg.ParOperator(ctx, op, e, linalg.DIAG_ONE, n_true=nt, halo=h)
             for op, e, nt, h in zip(local, self.ess, self.n_true, self.halos)]
        if coarse_assembled and len(A) > 1:
            # the coarsest level as a matrix, like the reference (ParOperator::ParallelAssemble, rap.cpp:84-152): at
            # p = 1 the CSR product moves ~3x fewer bytes than the matrix-free apply with the fine quadrature data
            csr0 = local[0].full_assemble_device()
            A[0] = linalg.AssembledParOperator(ctx, csr0, self.ess[0], linalg.DIAG_ONE, n_true=self.n_true[0],
                                               halo=self.halos[0])
        P = [linalg.Interp(ctx, self.spaces[l], self.spaces[l + 1], coarse_halo=self.halos[l],
                           n_true_c=self.n_true[l], n_true_f=self.n_true[l + 1]) for l in range(len(A) - 1)]
        aux = {}
        if hiptmair:
            # auxiliary-space smoothing (reference default for driven / eigenmode problems,
            # iodata.cpp:519-532): H1 diffusion with the mass coefficient, discrete gradients
            z_lo = self.rank * self.height
            h1s = [SlabH1Space(self.mesh, q, self.rank, self.world, z_lo, z_lo + self.height, self.radius)
                   for q in self.orders]
            h1_halos = [linalg.Halo(ctx, s.nbr, s.send, s.recv) if self.world > 1 else None for s in h1s]
            fine_h1 = ceed.diffusion_operator(self.geom, h1s[-1], mass)
            loc_h1 = [fine_h1.coarsen(self.geom, s) for s in h1s[:-1]] + [fine_h1]
            # PEC: the auxiliary potential vanishes on the boundary
            A_h1 = [linalg.ParOperator(ctx, op, s.ess_dofs(), linalg.DIAG_ONE, n_true=s.n_true, halo=h)
                    for op, s, h in zip(loc_h1, h1s, h1_halos)]
            G = [linalg.Gradient(ctx, h, n, h1_halo=hh, n_true_h1=h.n_true, n_true_nd=n.n_true)
                 for h, n, hh in zip(h1s, self.spaces, h1_halos)]
            aux = dict(A_aux=A_h1, G=G)
            self._keep.append((h1s, loc_h1, h1_halos))
        if len(A) > 1:
            # level 0: the reference calls AMS (HYPRE) here.  Stand-ins: "cg" = a few Jacobi-PCG iterations (needed by
            # the auxiliary-space configuration, where level 0 must really reduce the error), "chebyshev" = a fixed
            # Chebyshev-Jacobi smoother of order 4 (better with the plain smoother on the cylinder: an inexact inner CG is
            # a nonlinear preconditioner and costs the outer PCG 40 % more iterations; scripts/coarse_tune.py)
            if coarse == "chebyshev":
                csolver = linalg.chebyshev(ctx, A[0], 4)
            elif coarse == "ams" and self.world == 1:
                # the native auxiliary-space solver on the assembled level (linalg/ams.cpp; ksp.cpp:166-186)
                from .fespace import H1HexSpace, lowest_order_gradient, vertex_coordinates

                assert self.orders[0] == 1 and coarse_assembled, "AMS: assembled order-1 level"
                h1_0 = H1HexSpace(self.mesh, 1)
                csolver = linalg.ams(ctx, csr0, self.ess[0], lowest_order_gradient(h1_0, self.spaces[0]),
                                     vertex_coordinates(h1_0), singular=singular)
            elif coarse == "ams_dist":
                # several ranks, nothing but the ranks' own pieces: the C++ layer numbers the true dofs rank by rank, gathers the
                # local matrices / gradient rows / vertex coordinates over the communicator, and every rank builds the same AMS
                # solver of the global level (ksp.hpp: ReplicatedCoarseSolver -- what KspSolver's LinearSolver::AMS does on a space
                # with a halo; the reference: HYPRE's distributed AMS, linalg/ksp.cpp:129-239)
                from .fespace import vertex_coordinates

                assert self.world > 1 and self.orders[0] == 1, "ams_dist: several ranks, order-1 level"
                z_lo = self.rank * self.height
                h1_0 = SlabH1Space(self.mesh, 1, self.rank, self.world, z_lo, z_lo + self.height, self.radius)
                h1_halo0 = linalg.Halo(ctx, h1_0.nbr, h1_0.send, h1_0.recv)
                G0 = linalg.Gradient(ctx, h1_0, self.spaces[0], h1_halo=h1_halo0, n_true_h1=h1_0.n_true, n_true_nd=self.n_true[0])
                csolver = linalg.replicated_coarse(ctx, A[0], G0, h1_0.n_true, vertex_coordinates(h1_0)[: h1_0.n_true],
                                                   singular=singular)
                self._keep.append((h1_0, h1_halo0, G0))
            elif coarse == "ams":
                # several ranks: the order-1 problem of the WHOLE cylinder is assembled and solved redundantly by every rank
                # (linalg.replicated: the right-hand side gathered through a halo plan on the global-numbered vector)
                from .fespace import H1HexSpace, NDHexSpace, lowest_order_gradient, vertex_coordinates

                assert self.orders[0] == 1, "AMS: order-1 level"
                n, nz = self.shape
                gmesh = ogrid_cylinder(n, nz * self.world, radius=self.radius, height=self.height * self.world)
                g_nd, g_h1 = NDHexSpace(gmesh, 1), H1HexSpace(gmesh, 1)
                geom_g = ceed.GeomFactorData(gmesh, self.q1d)
                op_g = ceed.curlcurlmass_operator(geom_g, g_nd, mass, curl)
                csr_g = op_g.full_assemble_device()
                inner = linalg.ams(ctx, csr_g, g_nd.ess_dofs(), lowest_order_gradient(g_h1, g_nd), vertex_coordinates(g_h1))
                mine, sign, owner = global_edge_map(self.spaces[0], g_nd, self.height, self.world)
                others = [q for q in range(self.world) if q != self.rank]
                gather = linalg.Halo(ctx, others, [np.sort(mine).astype(np.int32) for _ in others],
                                     [np.nonzero(owner == q)[0].astype(np.int32) for q in others])
                csolver = linalg.replicated(ctx, gather, inner, mine, g_nd.ndofs, sign=sign)
                self._keep.append((geom_g, op_g, csr_g, inner, gather))
            else:
                csolver = linalg.cg(ctx, A[0], linalg.jacobi(ctx, A[0]), rel_tol=coarse_tol, max_it=coarse_max_it)
            B = linalg.gmg(ctx, A, P, csolver, cheby_order=max(2 * self.p, 4), **aux)
            self.last_coarse = csolver
        else:
            B = linalg.jacobi(ctx, A[0])
        K = linalg.cg(ctx, A[-1], B, rel_tol=rel_tol, max_it=max_it)
        n = self.n_true[-1]
        ones = torch.ones(n, dtype=torch.float64, device="cuda")
        if singular:  # a right-hand side in the range of the singular operator
            ctx.set_random(ones, 11 + self.rank)
        b = torch.empty_like(ones)
        A[-1].mult(ones, b)
        b[torch.from_numpy(self.ess[-1].astype(np.int64)).cuda()] = 0.0
        x = torch.zeros_like(b)
        self._keep.append((local, A, P, B))
        self.last_gmg = B if len(A) > 1 else None
        self.last_A = A
        return K, b, x


    def h1_pcg_gmg_solver(self, order=2, max_it=50, rel_tol=0.0, eps_r=2.08, coarse="amg"):
        """The electrostatic-type system of BASELINE config 4: H1 order-`order` diffusion (eps grad u, grad v), Dirichlet data
        on the whole boundary, PCG with p-multigrid (levels 1..order, 4th-kind Chebyshev of order max(2p, 4), plain smoothers:
        iodata.cpp:533-564 for the SPD problem types) and on the assembled order-1 level the algebraic V-cycle (`coarse` =
        "amg": where the reference calls BoomerAMG, linalg/amg.cpp; one rank) or Chebyshev-Jacobi of order 4 ("chebyshev")."""
        import torch

        from .. import ceed, linalg

        ctx = self.ctx
        eps = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([eps_r])])
        z_lo = self.rank * self.height
        orders = list(range(1, order + 1))
        h1s = [SlabH1Space(self.mesh, q, self.rank, self.world, z_lo, z_lo + self.height, self.radius) for q in orders]
        halos = [linalg.Halo(ctx, s.nbr, s.send, s.recv) if self.world > 1 else None for s in h1s]
        geom = self.geom if order + 1 == self.geom.q1d else ceed.GeomFactorData(self.mesh, order + 1)
        fine = ceed.diffusion_operator(geom, h1s[-1], eps)
        local = [fine.coarsen(geom, s) for s in h1s[:-1]] + [fine]
        ess = [s.ess_dofs() for s in h1s]
        A = [linalg.ParOperator(ctx, op, e, linalg.DIAG_ONE, n_true=s.n_true, halo=h)
             for op, e, s, h in zip(local, ess, h1s, halos)]
        csr0 = None
        if len(A) > 1:
            csr0 = local[0].full_assemble_device()
            A[0] = linalg.AssembledParOperator(ctx, csr0, ess[0], linalg.DIAG_ONE, n_true=h1s[0].n_true, halo=halos[0])
        P = [linalg.Interp(ctx, h1s[l], h1s[l + 1], coarse_halo=halos[l], n_true_c=h1s[l].n_true, n_true_f=h1s[l + 1].n_true)
             for l in range(len(A) - 1)]
        if len(A) > 1:
            if coarse == "amg":
                assert self.world == 1, "the native AMG cycle works on one rank's matrix"
                csolver = linalg.amg(ctx, csr0, ess[0])
            else:
                csolver = linalg.chebyshev(ctx, A[0], 4)
            B = linalg.gmg(ctx, A, P, csolver, cheby_order=max(2 * order, 4))
        else:
            B = linalg.jacobi(ctx, A[0])
        K = linalg.cg(ctx, A[-1], B, rel_tol=rel_tol, max_it=max_it)
        n = h1s[-1].n_true
        ones = torch.ones(n, dtype=torch.float64, device="cuda")
        b = torch.empty_like(ones)
        A[-1].mult(ones * torch.linspace(0.0, 1.0, n, dtype=torch.float64, device="cuda"), b)
        b[torch.from_numpy(ess[-1].astype(np.int64)).cuda()] = 0.0
        x = torch.zeros_like(b)
        self._keep.append((h1s, halos, local, A, P, B, csr0, geom))
        self.h1_fine = A[-1]
        return K, b, x


# ---- host-side executors of a halo plan over torch.distributed (CPU tests, gloo) -------------

def prolongate_dist(space, lx):
    """lx[ghosts] <- owner values, using torch.distributed point-to-point (any backend)."""
    import torch
    import torch.distributed as dist

    reqs, bufs = [], []
    for nb, s, r in zip(space.nbr, space.send, space.recv):
        if s.size:
            reqs.append(dist.isend(lx[torch.from_numpy(s.astype(np.int64))].contiguous(), nb))
        if r.size:
            buf = torch.empty(r.size, dtype=lx.dtype)
            bufs.append((r, buf))
            reqs.append(dist.irecv(buf, nb))
    for q in reqs:
        q.wait()
    for r, buf in bufs:
        lx[torch.from_numpy(r.astype(np.int64))] = buf
    return lx


def restrict_add_dist(space, ly):
    """ly[owned shared] += ghost contributions of the sharers."""
    import torch
    import torch.distributed as dist

    reqs, bufs = [], []
    for nb, s, r in zip(space.nbr, space.send, space.recv):
        if r.size:
            reqs.append(dist.isend(ly[torch.from_numpy(r.astype(np.int64))].contiguous(), nb))
        if s.size:
            buf = torch.empty(s.size, dtype=ly.dtype)
            bufs.append((s, buf))
            reqs.append(dist.irecv(buf, nb))
    for q in reqs:
        q.wait()
    for s, buf in bufs:
        ly.index_add_(0, torch.from_numpy(s.astype(np.int64)), buf)
    return ly
