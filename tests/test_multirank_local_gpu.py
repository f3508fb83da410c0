"""The multi-rank C++ code paths on ONE GPU: `world` ranks run as threads of this process, each with its own Context / stream and
its slab of the cylinder (palace_amd/fem/partition.py), joined by the in-process communicator (pa_local_group_*, comm.hpp:
LocalGroup -- the same Halo plans, pack / unpack kernels, in-place ghost ranges, ParOperator P / P^T, transfers and discrete
gradients with ghosts, global dots and the device-resident PCG scalars as under RCCL; only the transport differs).  The
basis-independent results -- PCG iteration count, ||b||^2, ||x||^2, x.Ax and the number of true dofs -- must equal those of the
undivided problem on one rank.  (The RCCL transport itself is exercised by tests/test_halo_gpu.py.)"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NZ = 8


def _rank_main(group, rank, world, p, hiptmair, out, errors, coarse="cg"):
    try:
        import torch

        from palace_amd import linalg
        from palace_amd.fem.partition import SlabProblem

        torch.cuda.set_device(0)
        ctx = linalg.Context()
        if world > 1:
            ctx.init_comm_local(group, rank)
        prob = SlabProblem(ctx, rank, world, p, 0, shape=(2, NZ // world))
        K, b, x = prob.pcg_gmg_solver(max_it=200, rel_tol=1e-9, hiptmair=hiptmair, coarse=coarse)
        K.mult(b, x)
        st = K.stats()
        A = prob._keep[-1][1][-1]
        y = torch.zeros_like(x)
        A.mult(x, y)
        # one more application of the fine operator alone, on a vector that is not a solve result
        z = torch.zeros_like(x)
        A.mult(b, z)
        # the complex layer across ranks: (A_r + i A_i)(b + i z) through ComplexParOperator with a halo (copy / mask / P,
        # one-pass local apply, P^T / fix-up), global complex dots
        from palace_amd import ceed

        nd = prob.spaces[-1]
        neg = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([-0.7])])
        cond = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([0.05])])
        Ar = ceed.curlcurlmass_operator(prob.geom, nd, neg, ceed.coefficient_context(3))
        Ai = ceed.ndmass_operator(prob.geom, nd, cond)
        Ac = linalg.ComplexParOperator(ctx, Ar, Ai, prob.ess[-1], linalg.DIAG_ONE, n_true=prob.n_true[-1], halo=prob.halos[-1])
        cr, ci = torch.zeros_like(x), torch.zeros_like(x)
        Ac.mult(b, z, cr, ci)
        out[rank] = dict(st, n=int(prob.n_true[-1]), xx=ctx.dot(x, x), xAx=ctx.dot(x, y), bb=ctx.dot(b, b), bAb=ctx.dot(b, z),
                         zz=ctx.dot(z, z), crcr=ctx.dot(cr, cr), cici=ctx.dot(ci, ci), crci=ctx.dot(cr, ci))
        ctx.synchronize()
    except Exception as e:  # a failing rank must not leave the others waiting at a barrier for ever
        errors.append((rank, repr(e)))
        if group is not None:
            group.abort()
        raise


def _run(world, p, hiptmair, coarse="cg"):
    from palace_amd import linalg

    group = linalg.LocalGroup(world) if world > 1 else None
    out, errors = [None] * world, []
    threads = [threading.Thread(target=_rank_main, args=(group, r, world, p, hiptmair, out, errors, coarse), daemon=True)
               for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank thread is stuck (collective sequence out of step?)"
    res = dict(out[0])
    res["n"] = sum(o["n"] for o in out)
    for o in out[1:]:  # the reductions are global: every rank holds the same values
        for k in ("xx", "xAx", "bb", "bAb", "zz", "crcr", "cici", "crci", "iterations"):
            assert o[k] == out[0][k], (k, o[k], out[0][k])
    return res


@pytest.mark.parametrize("hiptmair", [False, True])
@pytest.mark.parametrize("p", [2, 3])
def test_ranks_as_threads_match_one_rank(p, hiptmair):
    one = _run(1, p, hiptmair)
    assert one["converged"]
    for world in (2, 4, 8):
        many = _run(world, p, hiptmair)
        assert many["converged"] and many["n"] == one["n"], (world, many["n"], one["n"])
        assert abs(many["iterations"] - one["iterations"]) <= 1, (world, many["iterations"], one["iterations"])
        for k in ("bb", "bAb", "zz", "crcr", "cici", "crci"):  # operator applies: rounding only
            assert abs(many[k] - one[k]) < 1e-11 * abs(one[k]), (world, k, many[k], one[k])
        for k in ("xx", "xAx"):  # solves to 1e-9
            assert abs(many[k] - one[k]) < 1e-6 * abs(one[k]), (world, k, many[k], one[k])


def test_slab_ranks_with_the_replicated_ams_coarse_solve():
    """The hexahedral cylinder cut into z-slabs (bench.py's partition) with the native AMS on level 0 across ranks: every rank
    assembles the order-1 problem of the whole cylinder and applies the same solver to the gathered right-hand side
    (partition.global_edge_map gives the global dof and the orientation sign of every slab edge).  Same solve as on one rank."""
    one = _run(1, 3, True, "ams")
    assert one["converged"]
    for world in (2, 4):
        many = _run(world, 3, True, "ams")
        assert many["converged"] and many["n"] == one["n"]
        assert abs(many["iterations"] - one["iterations"]) <= 1, (world, many["iterations"], one["iterations"])
        for k in ("bb", "bAb", "zz"):
            assert abs(many[k] - one[k]) < 1e-11 * abs(one[k]), (world, k, many[k], one[k])
        for k in ("xx", "xAx"):
            assert abs(many[k] - one[k]) < 1e-6 * abs(one[k]), (world, k, many[k], one[k])


def test_ranks_as_threads_with_halo_stream_overlap():
    """The same with PALACE_AMD_OVERLAP=1: ghosts exchanged on the second stream while the interior element batches run."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests.test_multirank_local_gpu import _run\n"
            "one, four = _run(1, 3, True), _run(4, 3, True)\n"
            "assert four['converged'] and four['n'] == one['n'] and abs(four['iterations'] - one['iterations']) <= 1\n"
            "for k in ('bb', 'bAb', 'zz'):\n"
            "    assert abs(four[k] - one[k]) < 1e-11 * abs(one[k]), (k, four[k], one[k])\n"
            "print('OK')\n") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PALACE_AMD_OVERLAP="1"))
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


# ---- tetrahedra under a general (recursive coordinate bisection) element partition -------------------------------------------

def _tet_rank_main(group, rank, world, hiptmair, out, errors, coarse="cg"):
    try:
        import torch

        from palace_amd import linalg
        from palace_amd.fem import tet
        from palace_amd.fem.tetproblem import TetProblem

        torch.cuda.set_device(0)
        ctx = linalg.Context()
        if world > 1:
            ctx.init_comm_local(group, rank)
        mesh = tet.to_quadratic(tet.cube_tet_mesh(4), warp=lambda x: x + 0.02 * np.sin(2.0 * x[:, [1, 2, 0]]))
        prob = TetProblem(ctx, mesh, 2, rank=rank, world=world)
        K, b, x = prob.pcg_gmg_solver(max_it=300, rel_tol=1e-9, hiptmair=hiptmair, coarse=coarse)
        K.mult(b, x)
        st = K.stats()
        A = prob.A[-1]
        y, z = torch.zeros_like(x), torch.zeros_like(x)
        A.mult(x, y)
        A.mult(b, z)
        out[rank] = dict(st, n=int(prob.n_true[-1]), xx=ctx.dot(x, x), xAx=ctx.dot(x, y), bb=ctx.dot(b, b), bAb=ctx.dot(b, z),
                         zz=ctx.dot(z, z))
        ctx.synchronize()
    except Exception as e:
        errors.append((rank, repr(e)))
        if group is not None:
            group.abort()
        raise


def _tet_run(world, hiptmair, coarse="cg"):
    from palace_amd import linalg

    group = linalg.LocalGroup(world) if world > 1 else None
    out, errors = [None] * world, []
    threads = [threading.Thread(target=_tet_rank_main, args=(group, r, world, hiptmair, out, errors, coarse), daemon=True)
               for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank thread is stuck"
    res = dict(out[0])
    res["n"] = sum(o["n"] for o in out)
    for o in out[1:]:
        for k in ("xx", "xAx", "bb", "bAb", "zz", "iterations"):
            assert o[k] == out[0][k], (k, o[k], out[0][k])
    return res


@pytest.mark.parametrize("hiptmair", [False, True])
def test_tet_ranks_under_rcb_partition_match_one_rank(hiptmair):
    """Order-2 Nedelec tetrahedra (curl-oriented restriction, curved tet10 geometry), PCG + p-multigrid with dense-path
    operators, transfers and (hiptmair) auxiliary H1 spaces, the elements cut by recursive coordinate bisection
    (palace_amd/fem/rcb.py) into 2 and 3 parts: the results of the undivided problem."""
    one = _tet_run(1, hiptmair)
    assert one["converged"]
    for world in (2, 3):
        many = _tet_run(world, hiptmair)
        assert many["converged"] and many["n"] == one["n"], (world, many["n"], one["n"])
        # (the plain Chebyshev smoother converges slowly here -- ~100 iterations -- and its eigenvalue estimates start from
        # rank-dependent random vectors: the count moves by a few per cent; with the auxiliary-space smoother it is sharp)
        assert abs(many["iterations"] - one["iterations"]) <= max(1, 0.06 * one["iterations"]), (world, many["iterations"],
                                                                                                one["iterations"])
        for k in ("bb", "bAb", "zz"):
            assert abs(many[k] - one[k]) < 1e-11 * abs(one[k]), (world, k, many[k], one[k])
        for k in ("xx", "xAx"):
            assert abs(many[k] - one[k]) < 1e-6 * abs(one[k]), (world, k, many[k], one[k])


def test_tet_ranks_with_the_replicated_ams_coarse_solve():
    """The native AMS cycle on level 0 across ranks: every rank assembles the global order-1 matrix, builds the same solver and
    applies it to the gathered right-hand side (ReplicatedSolver) -- what HYPRE's distributed AMS is for in the reference.  Same
    solve as on one rank: iteration count (the coarse solve is bit-identical on every rank; the smoothers' eigenvalue estimates
    are not), solution norms; and far fewer iterations than with the Jacobi-PCG stand-in."""
    one = _tet_run(1, True, "ams")
    ref = _tet_run(1, True, "cg")
    assert one["converged"] and one["iterations"] <= ref["iterations"]
    for world in (2, 3):
        many = _tet_run(world, True, "ams")
        assert many["converged"] and many["n"] == one["n"]
        assert abs(many["iterations"] - one["iterations"]) <= 1, (world, many["iterations"], one["iterations"])
        for k in ("bb", "bAb", "zz"):
            assert abs(many[k] - one[k]) < 1e-11 * abs(one[k]), (world, k, many[k], one[k])
        for k in ("xx", "xAx"):
            assert abs(many[k] - one[k]) < 1e-6 * abs(one[k]), (world, k, many[k], one[k])


def test_failing_rank_releases_the_group():
    """A rank thread that throws between two barriers aborts the group: the other ranks leave their all-reduce with an error
    instead of waiting for ever (LocalGroup::Abort)."""
    import torch

    from palace_amd import linalg

    group = linalg.LocalGroup(2)
    seen = []

    def rank_main(rank):
        ctx = linalg.Context()
        ctx.init_comm_local(group, rank)
        x = torch.ones(8, dtype=torch.float64, device="cuda")
        try:
            if rank == 1:
                raise RuntimeError("rank 1 fails before the reduction")
            ctx.dot(x, x)
            seen.append("rank 0 came back")
        except RuntimeError as e:
            if rank == 1:
                group.abort()
            seen.append((rank, str(e)))

    ts = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert all(not t.is_alive() for t in ts), "rank 0 is stuck at the barrier"
    assert any(isinstance(s, tuple) and s[0] == 0 and "aborted" in s[1] for s in seen), seen
