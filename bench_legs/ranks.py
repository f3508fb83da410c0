"""Legs that run with several ranks (partition report, per-rank measurements).  Part of bench.py (split in round 6; `python bench.py` is the entry point)."""
import json  # noqa: F401
import os  # noqa: F401
import sys  # noqa: F401
import time  # noqa: F401

import numpy as np  # noqa: F401

from .common import HBM_PEAK_GBS, ROOT, _rel, host_cores, oracle_hex_data  # noqa: F401


def partition_report(space, halo_space_name="ND"):
    """Quality of the element partition as this rank sees it (surface / volume, neighbour counts)."""
    nbr = list(getattr(space, "nbr", []))
    ns = int(sum(len(q) for q in getattr(space, "send", [])))
    nr = int(sum(len(q) for q in getattr(space, "recv", [])))
    nt = int(getattr(space, "n_true", space.ndofs))
    return {"space": halo_space_name, "neighbours": len(nbr), "true_dofs": nt, "ghost_dofs": nr, "owned_dofs_sent": ns,
            "surface_to_volume": (ns + nr) / max(1, nt)}


def nranks_legs(ctx, rank, world, args, barrier, max_over_ranks):
    """N > 1: the other two element families of BASELINE's configs on the same N ranks -- order-4 hexahedra (config 5) as z-slabs
    of the strong-scaling cylinder, and order-`--order` Nedelec tetrahedra (configs 3 / 4 shape) cut by recursive coordinate
    bisection -- `ParOperator::Mult` throughput of the whole job and PCG + p-multigrid iterations/s, with the partition quality
    of each.  Every timed region is bracketed by barriers and the maximum over ranks is reported, like the headline."""
    import torch

    from palace_amd.fem.partition import SlabProblem, strong_shape

    def timed(fn, reps, warm):
        for _ in range(warm):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        return max_over_ranks(time.perf_counter() - t0) / reps

    out = {}
    try:
        n_cross, nz = strong_shape(args.dofs, 4)
        if nz % world:
            raise ValueError(f"{nz} layers do not divide into {world} slabs")
        prob = SlabProblem(ctx, rank, world, 4, args.dofs, levels=True, shape=(n_cross, nz // world))
        K = prob.curlcurl_par_operator()
        n = prob.n_true[-1]
        x = torch.rand(n, dtype=torch.float64, device="cuda")
        y = torch.empty_like(x)
        ng = prob.global_true_dofs()
        sec = timed(lambda: K.mult(x, y), 200, 30)
        e = {"workload": f"ND p=4 hexahedra, strong z-slabs x{world}, {ng} true dofs total", "global_true_dofs": ng,
             "direct_form": K.direct_form(), "mult_ms": 1e3 * sec, "dof_per_s": ng / sec,
             "partition": partition_report(prob.spaces[-1], "ND p=4, z-slabs")}
        solver, b, xs = prob.pcg_gmg_solver(max_it=20, hiptmair=False, coarse="chebyshev")
        solver.mult(b, xs)
        barrier()
        t0 = time.perf_counter()
        solver.mult(b, xs)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        st = solver.stats()
        e["pcg_chebyshev"] = {"iters_per_s": st["iterations"] / dt, "iterations": st["iterations"], "seconds": dt,
                              "levels": ",".join(str(q) for q in prob.orders)}
        prob._keep.clear()
        del prob, K, solver
        out["p4"] = e
    except Exception as exc:  # noqa: BLE001 -- reported in the line
        out["p4"] = {"error": f"{type(exc).__name__}: {exc}"}
    try:
        from palace_amd.fem import tet
        from palace_amd.fem.tetproblem import TetProblem

        mesh = tet.cube_tet_mesh(args.tet_n)
        prob = TetProblem(ctx, mesh, args.order, rank=rank, world=world)
        solver, b, xs = prob.pcg_gmg_solver(max_it=400, rel_tol=1e-8, hiptmair=True, coarse="ams")
        A = prob.A[-1]
        n = prob.n_true[-1]
        ngt = torch.tensor([n], dtype=torch.int64)
        import torch.distributed as dist

        if dist.get_backend() == "nccl":
            ngt = ngt.cuda()
        dist.all_reduce(ngt)
        ng = int(ngt.item())
        x = torch.rand(n, dtype=torch.float64, device="cuda")
        y = torch.empty_like(x)
        sec = timed(lambda: A.mult(x, y), 50, 10)
        e = {"workload": f"ND p={args.order} tetrahedra (dense MFMA path), {mesh.ne} tets cut into {world} parts by recursive "
                         f"coordinate bisection, {ng} true dofs total; K + M ParOperator::Mult and PCG + Hiptmair p-multigrid "
                         "with the replicated native AMS on level 0",
             "global_true_dofs": ng, "direct_form": A.direct_form(), "mult_ms": 1e3 * sec, "dof_per_s": ng / sec,
             "partition": partition_report(prob.spaces[-1], f"ND p={args.order} tets, RCB")}
        solver.mult(b, xs)  # warm-up (records the iteration)
        xs.zero_()
        barrier()
        t0 = time.perf_counter()
        solver.mult(b, xs)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        st = solver.stats()
        e["pcg_hiptmair_ams"] = {"iterations_to_1e-8": st["iterations"], "seconds": dt, "iters_per_s": st["iterations"] / dt,
                                 "converged": st["converged"]}
        prob._keep.clear()
        out["tets"] = e
    except Exception as exc:  # noqa: BLE001
        out["tets"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out
