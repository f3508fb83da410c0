"""General element partition (palace_amd/fem/rcb.py; reference: METIS parts of the serial mesh, utils/geodata.cpp:3587-3596,
and MFEM's conforming prolongation behind rap.cpp:195-234) on CPU: the bisection itself, the rank-local views of tetrahedral
Nedelec / H1 spaces (ownership, numbering, halo plans), `y = P^T A_local P x` against the undivided operator -- emulated rank by
rank in one process, and run by two gloo ranks through the same host executors as the slab plans."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _problem(p=2, n=3):
    from oracle import palace_oracle as po
    from palace_amd.fem import tet

    mesh = tet.to_quadratic(tet.cube_tet_mesh(n), warp=lambda x: x + 0.03 * np.sin(2.0 * x[:, [1, 2, 0]]))
    nd = tet.NDTetSpace(mesh, p)
    pts, wts = tet.tet_quadrature(p + 1)
    interp, curl = nd.elem.tables(pts)
    J = mesh.jacobians(pts)
    geom = po.build_geom_factor_33(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9))
    cent = mesh.nodes[mesh.elem_nodes[:, :4]].mean(axis=1)
    return mesh, nd, interp, curl, geom, cent


def _local_apply(view_or_space, elems, interp, curl, geom, x):
    """K + M of the given elements (oracle: dense tables, curl-oriented restriction, hdivmass_33) on a local vector."""
    from oracle import palace_oracle as po

    s = view_or_space
    kw = {} if s.diagonal_transform else dict(curl_orients=s.curl_orients)
    op = po.CeedOperatorOracle(s.ndofs, s.offsets, s.orients if s.diagonal_transform else None, interp, curl, geom[elems],
                               po.QF_HDIVMASS, po.CoeffCtx(attr_mat=[0], mat_coeff=[np.array([2.08])]), po.CoeffCtx(), **kw)
    return op.apply_add(x, np.zeros(s.ndofs))


def test_rcb_is_balanced_deterministic_and_compact():
    from palace_amd.fem.rcb import rcb

    rng = np.random.default_rng(0)
    c = rng.uniform(0, 1, (1000, 3)) * np.array([4.0, 1.0, 1.0])
    for nparts in (1, 2, 3, 5, 8):
        part = rcb(c, nparts)
        sizes = np.bincount(part, minlength=nparts)
        assert sizes.min() >= 1000 // nparts - 1 and sizes.max() <= -(-1000 // nparts) + 1, (nparts, sizes)
        assert np.array_equal(part, rcb(c, nparts))
    # the first cut of an elongated box is across its long axis: the two halves are separated in x
    part = rcb(c, 2)
    assert c[part == 0, 0].max() <= c[part == 1, 0].min()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_views_ownership_plans_and_operator(world):
    from palace_amd.fem.rcb import PartitionedSpace, rcb

    mesh, nd, interp, curl, geom, cent = _problem()
    part = rcb(cent, world)
    views = [PartitionedSpace(nd, part, r, world) for r in range(world)]
    # every dof has exactly one owner; the local numbering is owned-first
    owned = np.concatenate([v.l2g[: v.n_true] for v in views])
    assert owned.size == nd.ndofs and np.array_equal(np.sort(owned), np.arange(nd.ndofs))
    assert sum(v.elems.size for v in views) == mesh.ne
    # plans: what r sends to q is, dof for dof, what q receives from r
    for v in views:
        for k, q in enumerate(v.nbr):
            w = views[q]
            j = w.nbr.index(v.rank)
            assert np.array_equal(v.l2g[v.send[k]], w.l2g[w.recv[j]])
            assert np.all(v.send[k] < v.n_true) and np.all(v.recv[k] >= v.n_true)
        ghosts = np.concatenate(v.recv) if v.recv else np.zeros(0, np.int32)
        assert np.array_equal(np.sort(ghosts), np.arange(v.n_true, v.ndofs))  # every ghost is received exactly once
    # y = P^T A_local P x over the ranks equals the undivided operator
    x = np.random.default_rng(1).uniform(-1, 1, nd.ndofs)
    ref = _local_apply(nd, np.arange(mesh.ne), interp, curl, geom, x)
    y = np.zeros(nd.ndofs)
    for v in views:
        ly = _local_apply(v, v.elems, interp, curl, geom, v.to_local(x))
        np.add.at(y, v.l2g, ly)
    assert np.abs(y - ref).max() < 1e-12 * np.abs(ref).max()
    # essential dofs: the owned part of the global list, each exactly once over the ranks
    ess = np.concatenate([v.l2g[v.ess_dofs()] for v in views])
    assert np.array_equal(np.sort(ess), np.sort(nd.ess_dofs()))


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from palace_amd.fem.partition import prolongate_dist, restrict_add_dist
        from palace_amd.fem.rcb import PartitionedSpace, rcb

        mesh, nd, interp, curl, geom, cent = _problem()
        v = PartitionedSpace(nd, rcb(cent, world), rank, world)
        xg = np.random.default_rng(1).uniform(-1, 1, nd.ndofs)  # the same global vector on every rank
        ess = v.ess_dofs()
        tx = xg[v.l2g[: v.n_true]].copy()
        tx[ess] = 0.0
        lx = torch.zeros(v.ndofs, dtype=torch.float64)
        lx[: v.n_true] = torch.from_numpy(tx)
        prolongate_dist(v, lx)
        # P reproduced the owners' values in the ghost slots (owners zeroed their essential entries)
        want = xg[v.l2g].copy()
        gess = np.zeros(nd.ndofs, dtype=bool)
        gess[nd.ess_dofs()] = True
        want[gess[v.l2g]] = 0.0
        ghost_err = float(np.abs(lx.numpy() - want).max())
        ly = restrict_add_dist(v, torch.from_numpy(_local_apply(v, v.elems, interp, curl, geom, lx.numpy())))
        y = ly.numpy()[: v.n_true].copy()
        y[ess] = 0.0
        loc = torch.tensor([v.n_true, tx @ tx, tx @ y, y @ y, ess.size], dtype=torch.float64)
        dist.all_reduce(loc)
        g = torch.tensor([ghost_err], dtype=torch.float64)
        dist.all_reduce(g, op=dist.ReduceOp.MAX)
        if rank == 0:
            out.put(loc.tolist() + [float(g.item())])
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_match_the_undivided_operator():
    import torch.multiprocessing as mp

    q = mp.get_context("spawn").SimpleQueue()
    mp.spawn(_worker, args=(2, 29631, q), nprocs=2, join=True)
    got = q.get()
    mesh, nd, interp, curl, geom, cent = _problem()
    x = np.random.default_rng(1).uniform(-1, 1, nd.ndofs)
    ess = nd.ess_dofs()
    x[ess] = 0.0
    y = _local_apply(nd, np.arange(mesh.ne), interp, curl, geom, x)
    y[ess] = 0.0
    want = [nd.ndofs, x @ x, x @ y, y @ y, ess.size]
    assert got[5] == 0.0
    for a, b in zip(got[:5], want):
        assert abs(a - b) <= 1e-11 * max(1.0, abs(b)), (got, want)
