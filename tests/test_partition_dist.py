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
        assert r[5] < 1e-12, "ghost values after P differ from the local interpolant"
        for a, b in zip(r[1:4], s[1:4]):
            assert abs(a - b) <= 1e-11 * abs(b), (q, r, s)
