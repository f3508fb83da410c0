"""The multi-rank path on CPU: world_size 2, gloo.  Each rank builds its slab, its level spaces and
halo plans (palace_amd/fem/partition.py — the same plans the RCCL path executes on GPUs), applies
y = P^T A_local P x with the oracle as the local operator, and the basis-independent results are
compared with the serial oracle on the undivided cylinder:  ||x||^2, x.Ax, ||Ax||^2 and the number
of true dofs.  A wrong owner, a mis-ordered interface list or an inconsistent edge/face orientation
across the interface changes these numbers at O(1)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SHAPE = (2, 2)  # n, nz per rank
RADIUS = 2.74


def _field(x):
    return np.stack([np.sin(0.7 * x[..., 1]) + 0.3 * x[..., 2], np.cos(0.5 * x[..., 0]) * x[..., 2],
                     0.2 * x[..., 0] * x[..., 1] + np.sin(0.3 * x[..., 2])], axis=-1)


def _potential(x):
    return np.sin(0.4 * x[..., 0]) * np.cos(0.3 * x[..., 1]) + 0.1 * x[..., 2] ** 2


def _worker(rank, world, port, p, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import palace_oracle as po
        from palace_amd.fem.partition import SlabProblem, prolongate_dist, restrict_add_dist
        from tests import util

        prob = SlabProblem(None, rank, world, p, 0, shape=SHAPE, radius=RADIUS, device=False)
        res = {}
        for lvl, sp in enumerate(prob.spaces):
            q1d = p + 1
            geom = util.oracle_geom(prob.mesh, q1d)
            cm, bm = util.make_ctx("scalar")
            cc, bc = util.make_ctx("identity")
            blob = np.concatenate([bm, bc])
            xl = util.nd_interpolate(sp, _field)
            nt = sp.n_true
            # P: owners -> ghosts must reproduce the locally interpolated ghost values exactly
            lx = torch.from_numpy(xl.copy())
            lx[nt:] = 0.0
            prolongate_dist(sp, lx)
            ghost_err = float(np.abs(lx.numpy()[nt:] - xl[nt:]).max()) if sp.ndofs > nt else 0.0
            # essential dofs: zero them as ParOperator does, then y = P^T A P x
            tx = xl[:nt].copy()
            tx[sp.ess_dofs()] = 0.0
            lx = torch.zeros(sp.ndofs, dtype=torch.float64)
            lx[:nt] = torch.from_numpy(tx)
            prolongate_dist(sp, lx)
            ly = util.oracle_apply_c(sp, geom, "hdivmass", blob, lx.numpy(), q1d)
            ly = restrict_add_dist(sp, torch.from_numpy(ly))
            y = ly.numpy()[:nt].copy()
            y[sp.ess_dofs()] = 0.0
            loc = torch.tensor([nt, tx @ tx, tx @ y, y @ y, sp.ess_dofs().size], dtype=torch.float64)
            dist.all_reduce(loc)
            g = torch.tensor([ghost_err], dtype=torch.float64)
            dist.all_reduce(g, op=dist.ReduceOp.MAX)
            res[sp.p] = loc.tolist() + [float(g.item())]
        # H1 auxiliary space (Hiptmair): diffusion operator energy and the discrete gradient
        from palace_amd.fem.partition import SlabH1Space

        z_lo = rank * prob.height
        for sp_nd in prob.spaces:
            q = sp_nd.p
            h1 = SlabH1Space(prob.mesh, q, rank, world, z_lo, z_lo + prob.height, RADIUS)
            q1d = p + 1
            geom = util.oracle_geom(prob.mesh, q1d)
            interp, grad = po.h1_hex_dense_tables(q, q1d)
            A = po.CeedOperatorOracle(h1.ndofs, h1.elem_dof_lex, None, interp, grad, geom, po.QF_HCURL,
                                      po.CoeffCtx(), vector_fe=False)
            phi = util.h1_interpolate(h1, _potential)
            nt = h1.n_true
            tx = phi[:nt].copy()
            tx[h1.ess_dofs()] = 0.0
            lx = torch.zeros(h1.ndofs, dtype=torch.float64)
            lx[:nt] = torch.from_numpy(tx)
            prolongate_dist(h1, lx)
            ly = restrict_add_dist(h1, torch.from_numpy(A.apply_add(lx.numpy(), np.zeros(h1.ndofs))))
            y = ly.numpy()[:nt].copy()
            y[h1.ess_dofs()] = 0.0
            # gradient: G phi on local vectors (H1 ghosts filled by P), owned ND entries only
            Gm = po.InterpOracle(h1.elem_dof_lex, np.ones(h1.elem_dof_lex.shape, dtype=np.int8), sp_nd.elem_dof_lex,
                                 sp_nd.elem_sign_lex, h1.ndofs, sp_nd.ndofs, po.nd_hex_gradient_lex(q))
            lphi = torch.zeros(h1.ndofs, dtype=torch.float64)
            lphi[:nt] = torch.from_numpy(phi[:nt])
            prolongate_dist(h1, lphi)
            gphi = Gm.mult(lphi.numpy())[: sp_nd.n_true]
            loc = torch.tensor([nt, tx @ tx, tx @ y, y @ y, h1.ess_dofs().size, gphi @ gphi], dtype=torch.float64)
            dist.all_reduce(loc)
            res[("h1", q)] = loc.tolist()
        if rank == 0:
            out.put(res)
    finally:
        dist.destroy_process_group()


def _serial(p):
    from palace_amd.fem.fespace import NDHexSpace
    from palace_amd.fem.mesh import ogrid_cylinder
    from palace_amd.fem.partition import levels_for
    from tests import util

    n, nz = SHAPE
    h_layer = 2.0 * RADIUS / max(1, round(1.15 * n))
    mesh = ogrid_cylinder(n, 2 * nz, radius=RADIUS, height=2 * nz * h_layer)
    out = {}
    for q in levels_for(p):
        sp = NDHexSpace(mesh, q)
        q1d = p + 1
        geom = util.oracle_geom(mesh, q1d)
        cm, bm = util.make_ctx("scalar")
        cc, bc = util.make_ctx("identity")
        x = util.nd_interpolate(sp, _field)
        ess = sp.ess_dofs()
        x[ess] = 0.0
        y = util.oracle_apply_c(sp, geom, "hdivmass", np.concatenate([bm, bc]), x, q1d)
        y[ess] = 0.0
        out[q] = [sp.ndofs, x @ x, x @ y, y @ y, ess.size]
        from oracle import palace_oracle as po
        from palace_amd.fem.fespace import H1HexSpace

        h1 = H1HexSpace(mesh, q)
        interp, grad = po.h1_hex_dense_tables(q, q1d)
        A = po.CeedOperatorOracle(h1.ndofs, h1.elem_dof_lex, None, interp, grad, geom, po.QF_HCURL, po.CoeffCtx(),
                                  vector_fe=False)
        phi = util.h1_interpolate(h1, _potential)
        Gm = po.InterpOracle(h1.elem_dof_lex, np.ones(h1.elem_dof_lex.shape, dtype=np.int8), sp.elem_dof_lex,
                             sp.elem_sign_lex, h1.ndofs, sp.ndofs, po.nd_hex_gradient_lex(q))
        gphi = Gm.mult(phi)
        e1 = h1.ess_dofs()
        phi[e1] = 0.0
        yh = A.apply_add(phi, np.zeros(h1.ndofs))
        yh[e1] = 0.0
        out[("h1", q)] = [h1.ndofs, phi @ phi, phi @ yh, yh @ yh, e1.size, gphi @ gphi]
    return out


@pytest.mark.parametrize("p", [2, 3])
def test_two_rank_operator_matches_serial(p):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + p
    procs = [ctx.Process(target=_worker, args=(r, 2, port, p, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = out.get(timeout=300)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    ref = _serial(p)
    for q, r in res.items():
        s = ref[q]
        assert int(r[0]) == s[0], "true dof count"
        assert int(r[4]) == s[4], "essential dof count"
        if isinstance(q, tuple):  # H1 auxiliary space: energies + ||G phi||^2
            for a, b in zip(r[1:4] + r[5:6], s[1:4] + s[5:6]):
                assert abs(a - b) <= 1e-11 * abs(b), (q, r, s)
            continue
        assert r[5] < 1e-12, "ghost values after P differ from the local interpolant"
        for a, b in zip(r[1:4], s[1:4]):
            assert abs(a - b) <= 1e-11 * abs(b), (q, r, s)


@pytest.mark.parametrize("world,rank", [(2, 0), (2, 1), (8, 0), (8, 3), (8, 7)])
def test_ghosts_are_the_contiguous_tail_of_the_local_vector(world, rank):
    """The C++ Halo receives ghosts into, and sends them from, the local vector in place when they are one contiguous range in
    receive order (comm.hip: recv_first_): the slab plans of every level (Nedelec and H1) have that form -- owned dofs first, the
    ghosts of the lower interface last -- and send lists only address owned dofs."""
    from palace_amd.fem.partition import SlabH1Space, SlabProblem

    prob = SlabProblem(None, rank, world, 3, 0, shape=(2, 2), device=False)
    z_lo = rank * prob.height
    h1s = [SlabH1Space(prob.mesh, q, rank, world, z_lo, z_lo + prob.height, prob.radius) for q in prob.orders]
    for s in list(prob.spaces) + h1s:
        recv = np.concatenate(s.recv) if s.recv else np.zeros(0, np.int32)
        send = np.concatenate(s.send) if s.send else np.zeros(0, np.int32)
        assert recv.size == s.ndofs - s.n_true
        if recv.size:
            assert np.array_equal(recv, np.arange(s.n_true, s.ndofs))
        if send.size:
            assert send.min() >= 0 and send.max() < s.n_true and np.unique(send).size == send.size


@pytest.mark.parametrize("world", [2, 4])
def test_slab_edges_against_the_global_order_one_space(world):
    """partition.global_edge_map (the replicated level-0 solve of the slab problems): every global edge dof has exactly one owner
    among the slabs, the slabs' true dofs map onto them one to one, and with the orientation signs the Nedelec interpolant of a
    field on a slab equals the global interpolant entry by entry."""
    from palace_amd.fem.fespace import NDHexSpace
    from palace_amd.fem.mesh import ogrid_cylinder
    from palace_amd.fem.partition import SlabProblem, global_edge_map
    from tests import util

    probs = [SlabProblem(None, r, world, 3, 0, shape=(2, 2), device=False) for r in range(world)]
    gm = ogrid_cylinder(2, 2 * world, radius=probs[0].radius, height=probs[0].height * world)
    g = NDHexSpace(gm, 1)
    ug = util.nd_interpolate(g, _field)
    seen = np.zeros(g.ndofs, dtype=int)
    for r, pr in enumerate(probs):
        s0 = pr.spaces[0]
        mine, sign, owner = global_edge_map(s0, g, pr.height, world)
        assert (owner[mine] == r).all() and set(np.nonzero(owner == r)[0]) == set(mine.tolist())
        seen[mine] += 1
        ul = util.nd_interpolate(s0, _field)[: s0.n_true]
        assert np.abs(sign * ul - ug[mine]).max() < 1e-12
    assert (seen == 1).all()
