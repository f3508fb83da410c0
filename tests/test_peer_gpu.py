"""The peer transport (palace_amd/csrc/comm.hpp: halo exchanges and global sums as direct stores into the other ranks' device
arenas, plain kernels on the solver's stream) -- between PROCESSES sharing one GPU (hipIpc handles gathered over gloo; RCCL
refuses two ranks on one device).  Same checks as the other multi-rank tests: the basis-independent results of a PCG +
p-multigrid solve equal those of the undivided problem, also when the recorded iteration (HIP graph) is replayed.
(Rank THREADS of one process cannot use this transport: they share the process's legacy null stream, and a rank's wait kernel
would hold up null-stream work another rank has to finish first -- the in-process group keeps its host-barrier copies.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def ctx_dot_vec(ctx, v):
    import torch

    return torch.tensor(ctx.dot(v, v), dtype=torch.float64)


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)  # every rank on the same GPU
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from palace_amd import lib as _lib
        from palace_amd import linalg
        from palace_amd.fem.partition import SlabProblem

        ctx = linalg.Context()
        if world > 1:
            ctx.init_comm_peer_from_torch_distributed()
        prob = SlabProblem(ctx, rank, world, 3, 0, shape=(2, 4 // world))  # (order 3: the four-point kernels, direct form)
        if world > 1:
            assert all(_lib.load().pa_halo_uses_peer(h.handle) for h in prob.halos)
        K, b, x = prob.pcg_gmg_solver(max_it=100, rel_tol=1e-8, hiptmair=True, coarse="cg")
        K.mult(b, x)
        st = K.stats()
        A = prob._keep[-1][1][-1]
        y = torch.zeros_like(x)
        A.mult(x, y)
        z = torch.zeros_like(x)
        A.mult(b, z)
        nt = torch.tensor([prob.n_true[-1]], dtype=torch.int64)
        dist.all_reduce(nt)
        res = dict(st, n=int(nt.item()), xx=ctx.dot(x, x), xAx=ctx.dot(x, y), bb=ctx.dot(b, b), bAb=ctx.dot(b, z))
        ctx.peer_check() if world > 1 else None
        if world > 1:
            # the two forms of the multi-rank Mult (direct: no L-vector copies, ghosts read from the mailbox; L-vector form) give
            # the same bits: the same numbers are summed in the same order
            assert A.direct_form() == 1
            A.set_direct(False)
            assert A.direct_form() == 0
            z2 = torch.zeros_like(z)
            A.mult(b, z2)
            A.set_direct(True)
            res["forms_equal"] = bool(torch.equal(z, z2))
            # the assembled coarsest level (CSR product on split vectors)
            A0 = prob._keep[-1][1][0]
            assert A0.direct_form() == 1
            n0 = prob.n_true[0]
            v = torch.rand(n0, dtype=torch.float64, device="cuda")
            w1, w2 = torch.zeros_like(v), torch.zeros_like(v)
            A0.mult(v, w1)
            A0.set_direct(False)
            A0.mult(v, w2)
            A0.set_direct(True)
            res["forms_equal"] = res["forms_equal"] and bool(torch.allclose(w1, w2, rtol=1e-14, atol=1e-14 * float(w2.abs().max())))
        # round 6: the smoother step consumed where A e_k is produced -- on one rank in the E^T gather, with a halo in the gather
        # (dofs no other rank shares) and in the merged P^T kernel (interface dofs): against the same smoother with the step as a
        # vector kernel, zero and non-zero initial guess; the V-cycle's fused residual rides in the solves above
        S1 = linalg.chebyshev(ctx, A, order=4)
        os.environ["PALACE_AMD_FUSED_STEP"] = "0"
        S0 = linalg.chebyshev(ctx, A, order=4)
        os.environ.pop("PALACE_AMD_FUSED_STEP")
        res["fused"] = (S1.fused_step(), S0.fused_step())
        assert S1.lambda_max() == S0.lambda_max()
        y1, y0 = S1.mult(b, torch.zeros_like(b)), S0.mult(b, torch.zeros_like(b))
        g = 0.3 * z
        g1, g0 = S1.mult(b, g.clone(), initial_guess=True), S0.mult(b, g.clone(), initial_guess=True)
        dd = torch.stack([ctx_dot_vec(ctx, y1 - y0), ctx_dot_vec(ctx, y0), ctx_dot_vec(ctx, g1 - g0), ctx_dot_vec(ctx, g0)])
        res["cheb_rel"] = (float(dd[0] / dd[1]) ** 0.5, float(dd[2] / dd[3]) ** 0.5)
        if world > 1:
            ctx.peer_check()
        # the same solve again: the recorded iteration (HIP graph) replays across ranks
        x.zero_()
        K.mult(b, x)
        res["xx2"] = ctx.dot(x, x)
        if world > 1:
            ctx.peer_check()
        if rank == 0:
            out.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_processes_on_one_gpu_match_one_rank():
    import torch.multiprocessing as mp

    results = {}
    for world, port in ((1, 29621), (2, 29622)):
        q = mp.get_context("spawn").SimpleQueue()
        mp.spawn(_worker, args=(world, port, q), nprocs=world, join=True)
        results[world] = q.get()
    one, two = results[1], results[2]
    assert one["n"] == two["n"] and one["converged"] and two["converged"]
    assert abs(one["iterations"] - two["iterations"]) <= 1
    assert two["forms_equal"]
    # the fused smoother step: taken on one rank and with the halo, equal to the unfused smoother
    assert one["fused"] == (True, False) and two["fused"] == (True, False), (one["fused"], two["fused"])
    assert max(one["cheb_rel"]) < 1e-13 and max(two["cheb_rel"]) < 1e-13, (one["cheb_rel"], two["cheb_rel"])
    for k in ("bb", "bAb"):
        assert abs(one[k] - two[k]) < 1e-11 * abs(one[k]), (k, one[k], two[k])
    for k in ("xx", "xAx", "xx2"):
        assert abs(one[k] - two[k]) < 1e-6 * abs(one[k]), (k, one[k], two[k])


def _tet_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from palace_amd import linalg
        from palace_amd.fem import tet
        from palace_amd.fem.tetproblem import TetProblem

        ctx = linalg.Context()
        if world > 1:
            ctx.init_comm_peer_from_torch_distributed()
        mesh = tet.cube_tet_mesh(4)
        prob = TetProblem(ctx, mesh, 2, rank=rank, world=world)
        K, b, x = prob.pcg_gmg_solver(max_it=200, rel_tol=1e-9, hiptmair=True, coarse="ams")
        res = {}
        for rep in range(3):  # direct run, recording, replay: the gather of the replicated coarse solve is part of the recording
            x.zero_()
            K.mult(b, x)
            st = K.stats()
            res[f"its{rep}"], res[f"xx{rep}"] = st["iterations"], ctx.dot(x, x)
            assert st["converged"]
        if world > 1:
            ctx.peer_check()
        if rank == 0:
            out.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_replicated_ams_over_the_peer_transport():
    """The replicated level-0 solve (every rank applies the native AMS to the right-hand side gathered from all ranks) with the
    gather running over the peer transport between two processes, inside the recorded multigrid cycle."""
    import torch.multiprocessing as mp

    results = {}
    for world, port in ((1, 29641), (2, 29642)):
        q = mp.get_context("spawn").SimpleQueue()
        mp.spawn(_tet_worker, args=(world, port, q), nprocs=world, join=True)
        results[world] = q.get()
    one, two = results[1], results[2]
    for rep in range(3):
        assert abs(one[f"its{rep}"] - two[f"its{rep}"]) <= 1, (one, two)
        assert abs(one[f"xx{rep}"] - two[f"xx{rep}"]) < 1e-6 * one[f"xx{rep}"], (one, two)
    assert two["its0"] == two["its1"] == two["its2"] and two["xx1"] == two["xx2"]


def _stress_worker(rank, world, port, rounds, out, fenced=False):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if fenced:
        os.environ["PALACE_AMD_PEER_FENCE"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from palace_amd import linalg

        ctx = linalg.Context()
        ctx.init_comm_peer_from_torch_distributed()  # (runs the start-up self-test: 3 x 1 000 rounds)
        res = {"bring_up": ctx.transport_report, "fenced": bool(linalg._L().pa_comm_peer_fenced())}
        ring = ctx._ring_plan(4096)
        # both buffer parities (odd and even round counts leave the next call starting on the other buffer), inside and
        # outside graph replay, the L-vector and the direct form interleaved on ONE plan
        for name, r, direct, graph in (("lvector", rounds, False, False), ("direct", rounds + 1, True, False),
                                       ("lvector_graph", rounds // 2, False, True), ("direct_graph", rounds // 2 + 1, True, True),
                                       ("lvector_again", 7, False, False)):
            res[name] = ctx.peer_stress(r, 4096, direct=direct, graph=graph, ring=ring)
        ctx.peer_check()
        # plans come and go: the descriptor slots and arena blocks of destroyed plans are used again (more plans than slots)
        for _ in range(560):
            h = ctx._ring_plan(64)
            del h
        res["after_churn"] = ctx.peer_stress(50, 4096, direct=True, graph=False)
        if rank == 0:
            out.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_flag_protocol_stress(world):
    """1.5e5 rounds (3e5 P / P^T exchanges + 1.5e5 all-reduces) between two processes on one GPU -- through the local vector and in the
    direct form, outside and inside graph replay, odd and even counts (both mailbox buffers) --, payloads that change every round,
    every received value verified on the device: a reordering of data and flag stores would show up as a wrong value."""
    import torch.multiprocessing as mp

    rounds = 50000 if world == 2 else 5000
    q = mp.get_context("spawn").SimpleQueue()
    mp.spawn(_stress_worker, args=(world, 29660 + world, rounds, q), nprocs=world, join=True)
    res = q.get()
    assert res["bring_up"]["transport"] == "peer" and res["bring_up"]["ordering"] == "relaxed", res
    for k in ("lvector", "direct", "lvector_graph", "direct_graph", "lvector_again", "after_churn"):
        assert res[k] == 0, res


def _dist_coarse_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from palace_amd import linalg
        from palace_amd.fem.partition import SlabProblem

        ctx = linalg.Context()
        if world > 1:
            ctx.init_comm_peer_from_torch_distributed()
        prob = SlabProblem(ctx, rank, world, 2, 0, shape=(2, 4 // world))
        res = {}
        for kind in (("ams",) if world == 1 else ("ams", "ams_dist")):
            K, b, x = prob.pcg_gmg_solver(max_it=100, rel_tol=1e-9, hiptmair=True, coarse=kind)
            K.mult(b, x)
            st = K.stats()
            assert st["converged"], (kind, st)
            res[kind] = (st["iterations"], ctx.dot(x, x))
            if kind == "ams_dist":  # (round 5: the C++ layer's solver applies its rows of every algebraic level, amg_dist.hpp)
                res["ams_dist_form"] = (bool(prob.last_coarse.distributed), int(prob.last_coarse.algebraic_levels))
            x.zero_()
            K.mult(b, x)  # (the recorded iteration replays: the gather of the replicated solve is part of it)
            res[kind + "_again"] = (K.stats()["iterations"], ctx.dot(x, x))
        if world > 1:
            ctx.peer_check()
        if rank == 0:
            out.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_replicated_coarse_solver_assembled_from_the_ranks_pieces():
    """LinearSolver::AMS on a space with a halo (ksp.hpp: ReplicatedCoarseSolver): the C++ layer builds the global level-0 problem
    from the ranks' local matrices, gradient rows and vertex coordinates alone.  Against the replicated solver built from the
    global mesh (two processes) and the one-rank solve: iterations +- 1, the same solution."""
    import torch.multiprocessing as mp

    results = {}
    for world, port in ((1, 29671), (2, 29672)):
        q = mp.get_context("spawn").SimpleQueue()
        mp.spawn(_dist_coarse_worker, args=(world, port, q), nprocs=world, join=True)
        results[world] = q.get()
    one, two = results[1], results[2]
    assert two["ams_dist_form"][0] is True and two["ams_dist_form"][1] >= 1, two
    for kind in ("ams", "ams_dist"):
        assert abs(two[kind][0] - one["ams"][0]) <= 1, (kind, one, two)
        assert abs(two[kind][1] - one["ams"][1]) < 1e-6 * one["ams"][1], (kind, one, two)
        assert two[kind + "_again"][0] == two[kind][0]


def test_flag_protocol_stress_with_system_scope_fences():
    """The fall-back tier of the transport (PALACE_AMD_PEER_FENCE=1 / pa_comm_peer_set_fenced: system-scope release / acquire fences
    around the flag stores and waits instead of relaxed atomics + s_waitcnt) runs the same stress test: it must be there when the
    relaxed protocol fails its self-test on some machine."""
    import torch.multiprocessing as mp

    q = mp.get_context("spawn").SimpleQueue()
    mp.spawn(_stress_worker, args=(2, 29669, 4000, q, True), nprocs=2, join=True)
    res = q.get()
    assert res["fenced"] and res["bring_up"]["transport"] == "peer", res
    for k in ("lvector", "direct", "lvector_graph", "direct_graph", "lvector_again", "after_churn"):
        assert res[k] == 0, res
