#!/usr/bin/env python
"""bench.py — the hot path on MI355X: Nedelec p=3 curl-curl `ParOperator::Mult` throughput and PCG
iterations/s on a ~10M-dof cylinder cavity (BASELINE.json metric), one JSON line on rank 0.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL over xGMI)

A "step" is one `ParOperator::Mult` (BC masking + P + fused E-B-D-B^T-E^T kernel + P^T) of the
curl-curl operator over the whole vector.  `value` = true dofs processed per second by all ranks.
Scaling is weak: every rank owns a z-slab of the cylinder with the same number of elements
(~10M dofs per GPU); `config.scaling_mode` says so.  Extra keys: `roofline` (fused apply kernel,
algorithmic bytes of SURVEY.md 8(d) / HIP-event kernel time), `cpu_baseline` (the oracle's C port of
the reference's dense-table CPU path on a bounded sample), `pcg` (iterations/s of PCG + p-multigrid).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--order", type=int, default=3)
    ap.add_argument("--dofs", type=float, default=10.0e6, help="target true dofs per GPU")
    ap.add_argument("--pcg-iters", type=int, default=50, help="fixed PCG iterations for iterations/s (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-dofs", type=float, default=1.0e6, help="size of the CPU baseline sample")
    ap.add_argument("--no-tets", action="store_true", help="skip the tetrahedral (dense MFMA path) leg")
    ap.add_argument("--tet-n", type=int, default=36, help="cubes per direction of the Kuhn-split tet mesh")
    ap.add_argument("--force-comm", action="store_true",
                    help="create the RCCL communicator even with one rank (exercises the multi-GPU code path)")
    return ap.parse_args()


def cpu_baseline(order, target_dofs):
    """The oracle's C restatement of the reference CPU path (dense [3Q x P] tables, libCEED-style
    blocked E/B/D/B^T/E^T, OpenMP over element ranges) timed on this host's cores, on a smaller
    cylinder of the same family (bounded sample: ~10-30 s of CPU work)."""
    from oracle import capi
    from oracle import palace_oracle as po
    from palace_amd.fem.fespace import NDHexSpace
    from palace_amd.fem.mesh import cylinder_for_dofs
    from tests import util

    capi.build(ref=False)
    mesh = cylinder_for_dofs(target_dofs, order)
    nd = NDHexSpace(mesh, order)
    q1d = order + 1
    geom = util.oracle_geom(mesh, q1d)
    off, ori = nd.native_restriction()
    interp, curl = po.nd_hex_dense_tables(order, q1d, nd.dof_map_native())
    blob = po.CoeffCtx().pack()
    x = np.random.default_rng(1).uniform(0, 1, nd.ndofs)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 64)  # the element loop stops scaling beyond a socket's worth of threads
    y = np.zeros(nd.ndofs)
    capi.apply_add(off, ori, interp, curl, geom, capi.QF_HDIV, blob, x, y, threads=cores)  # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        y[:] = 0.0
        capi.apply_add(off, ori, interp, curl, geom, capi.QF_HDIV, blob, x, y, threads=cores)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 10.0:
            break
    return {"value": nd.ndofs * reps / dt, "unit": "DOF/s", "cores": cores, "kind": "port",
            "sample": f"curl-curl apply, ND p={order}, {mesh.ne} hex27 elements, {nd.ndofs} dofs, {reps} applies "
                      f"in {dt:.1f} s; oracle/oracle_c.c (dense-table libCEED-style CPU path restated), OpenMP"}


def tets_leg(order, n, reps=20):
    """The non-tensor path (dense tables on the FP64 matrix cores): Nedelec tets of the same order on a
    Kuhn-split cube, curl-curl and curl-curl+mass `ceed::Operator::Mult`, order-2p symmetric quadrature
    (the reference's default rule size).  Reported beside the headline, N = 1 only."""
    import torch

    from palace_amd import ceed
    from palace_amd.fem import tet

    mesh = tet.cube_tet_mesh(n)
    nd = tet.NDTetSpace(mesh, order)
    pts, wts = tet.default_tet_rule(order)
    interp, curl = nd.elem.tables(pts)
    geom = ceed.DenseGeomFactorData(mesh.elem_nodes, mesh.nodes, mesh.attr, mesh.geometry_grad_table(pts), wts)
    kw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
    block = ceed.DenseBlock(ceed.FE_HCURL, nd.ndofs, nd.offsets, interp, curl, **kw)
    ident = ceed.coefficient_context(3)
    mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
    ops = {"curlcurl": (ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(
                            geom, block, ceed.QF_HDIV_33, ident, ceed.EVAL_CURL).finalize(), 3),
           "curlcurl_mass": (ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(
                                 geom, block, ceed.QF_HDIVMASS_33, np.concatenate([mass, ident]),
                                 ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize(), 6)}
    x = torch.rand(nd.ndofs, dtype=torch.float64, device="cuda")
    y = torch.zeros_like(x)
    out = {"workload": f"ND p={order} tetrahedra (curl-oriented restriction), {mesh.ne} tets, {nd.ndofs} dofs, "
                       f"P={nd.P}, Q={len(wts)}; dense [3Q x P] tables on v_mfma_f64_16x16x4", "dofs": nd.ndofs}
    for name, (op, nct) in ops.items():
        for _ in range(3):
            op.mult(x, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            op.mult(x, y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        alg = op.algorithmic_bytes()
        out[name] = {"ms": ms, "dof_per_s": nd.ndofs / (ms * 1e-3), "algorithmic_GBps": alg / ms / 1e6,
                     "hbm_frac": alg / ms / 1e6 / HBM_PEAK_GBS,
                     "table_TFLOPs": mesh.ne * (2 * 2 * nct * len(wts) * nd.P) / ms / 1e9}
    # PCG + p-multigrid (p = 1..order) with the auxiliary-space smoother on the same mesh
    from palace_amd import linalg
    from palace_amd.fem.tetproblem import TetProblem

    prob = TetProblem(linalg.Context(), mesh, order)
    solver, b, xs = prob.pcg_gmg_solver(max_it=400, rel_tol=1e-8, hiptmair=True)
    solver.mult(b, xs)  # warm-up
    xs.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.mult(b, xs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = solver.stats()
    out["pcg_hiptmair"] = {"iterations_to_1e-8": st["iterations"], "seconds": dt, "iters_per_s": st["iterations"] / dt,
                           "converged": st["converged"]}
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge

    if not os.path.exists(os.path.join(ROOT, "palace_amd", "lib", "libpalace_amd.so")):
        if rank == 0:
            ge.build()
        if world > 1:
            dist.barrier()

    from palace_amd import ceed, linalg
    from palace_amd.fem.partition import SlabProblem

    p = args.order
    ctx = linalg.Context()
    if world > 1:
        ctx.init_comm_from_torch_distributed()
    elif args.force_comm:
        ctx.init_comm_single()

    # ---- set-up (not timed): mesh slab, spaces, geometry factors, operators ---------------------
    t_setup = time.perf_counter()
    prob = SlabProblem(ctx, rank, world, p, args.dofs, levels=True)
    K = prob.curlcurl_par_operator()  # ParOperator(curl-curl, mu^-1 = 1), PEC essential dofs, DIAG_ONE
    n_true = prob.n_true[-1]
    n_global = prob.global_true_dofs()
    x = torch.empty(n_true, dtype=torch.float64, device="cuda")
    y = torch.empty(n_true, dtype=torch.float64, device="cuda")
    ctx.set_random(x, 1 + rank)
    x.add_(1.0).mul_(0.5)  # uniform [0, 1)
    x[torch.from_numpy(prob.ess[-1].astype(np.int64)).cuda()] = 0.0
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- M1: ParOperator::Mult throughput --------------------------------------------------------
    for _ in range(args.warmup):
        K.mult(x, y)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        K.mult(x, y)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = n_global * args.steps / elapsed

    # ---- roofline of the dominant kernel (fused apply), HIP events on the launch stream -----------
    local_op = prob.local_curlcurl
    lx = torch.zeros(prob.n_local[-1], dtype=torch.float64, device="cuda")
    ly = torch.zeros(prob.n_local[-1], dtype=torch.float64, device="cuda")
    lx[:n_true] = x
    for _ in range(3):
        local_op.mult(lx, ly)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nk = max(10, min(args.steps, 100))
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(nk):
        local_op.mult(lx, ly)  # the same two kernels ParOperator::Mult launches (overwrite form)
    ev1.record()
    torch.cuda.synchronize()
    kernel_ms = ev0.elapsed_time(ev1) / nk
    alg_bytes = local_op.algorithmic_bytes()
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    # HBM traffic of the same launch from the PMC passes (collected by scripts/profile_round.sh in separate
    # rocprofv3 --pmc runs, summary committed under profiles/): raw FETCH_SIZE + WRITE_SIZE bytes
    traffic, traffic_note = None, "no PMC summary under profiles/"
    pmc_file = os.path.join(ROOT, "profiles", "r01_apply_pmc.json")
    if os.path.exists(pmc_file) and abs(args.dofs - 10.0e6) < 1 and p == 3:
        pmc = json.load(open(pmc_file))
        pb = pmc["per_apply_bytes"]
        traffic = pb.get("traffic_corrected") or pb["traffic_raw"]
        traffic_note = ("FETCH_SIZE + WRITE_SIZE per apply from profiles/r01_apply_pmc.json (separate --pmc passes of the "
                        "same kernels at this size), each divided by the fraction the same counters report on a known "
                        "stream in the same run; " + pmc.get("calibration", "uncalibrated"))
    # measured peaks of this GPU in the same run: a streaming y = a x + b y over 2 x 0.5 GB
    # (16 B read + 8 B written per entry) and the FP64 matrix-core micro-kernel; `peak` stays the guide's figure
    def _event_ms(fn, reps):
        fn()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a0.record()
        for _ in range(reps):
            fn()
        a1.record()
        torch.cuda.synchronize()
        return a0.elapsed_time(a1) / reps
    ns = 1 << 26
    sx = torch.ones(ns, dtype=torch.float64, device="cuda")
    sy = torch.ones(ns, dtype=torch.float64, device="cuda")
    stream_ms = _event_ms(lambda: ctx.axpby(0.5, sx, 0.5, sy), 20)
    measured_stream = 24.0 * ns / (stream_ms * 1e-3) / 1e9
    mf = {}
    mfma_ms = _event_ms(lambda: mf.__setitem__("flops", ctx.bench_mfma_f64(4096, 4096, sx)), 5)
    measured_mfma = mf["flops"] / (mfma_ms * 1e-3) / 1e12
    del sx, sy
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "measured_stream_GBps": measured_stream, "frac_of_measured_stream": achieved / measured_stream,
                "measured_mfma_f64_TFLOPs": measured_mfma,
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                "kernel": "pa::nd_hex_apply_kernel<3,4,curl,qdata> + pa::et_gather_kernel (E^T)", "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_formula": "NE*(Q*11*8 + P*5) + 16*N_L (SURVEY.md 8d, G=11)"}

    # ---- M2: PCG + p-multigrid on (K + M) x = b ---------------------------------------------------
    # iterations/s over a fixed number of iterations (rel_tol = 0, so the count is the same on any
    # device), for the two smoother configurations the reference uses, plus iterations-to-1e-8.
    pcg = None
    if args.pcg_iters > 0:
        pcg = {}
        for name, hip in (("chebyshev", False), ("hiptmair", True)):
            coarse = "cg" if hip else "chebyshev"
            solver, b, xs = prob.pcg_gmg_solver(max_it=args.pcg_iters, hiptmair=hip, coarse=coarse)
            solver.mult(b, xs)  # warm-up solve (also first-touch of all work vectors)
            barrier()
            t0 = time.perf_counter()
            solver.mult(b, xs)
            barrier()
            dt = time.perf_counter() - t0
            st = solver.stats()
            entry = {"iters_per_s": st["iterations"] / dt, "iterations": st["iterations"], "seconds": dt,
                     "final_rel_res": st["final_res"] / st["initial_res"]}
            solver, b, xs = prob.pcg_gmg_solver(max_it=400, rel_tol=1e-8, hiptmair=hip, coarse=coarse)
            barrier()
            t0 = time.perf_counter()
            solver.mult(b, xs)
            barrier()
            st = solver.stats()
            entry.update({"iterations_to_1e-8": st["iterations"], "seconds_to_1e-8": time.perf_counter() - t0,
                          "converged": st["converged"]})
            pcg[name] = entry
            prob._keep.clear()
        pcg["config"] = (f"PCG on K+M (eps_r=2.08), p-multigrid levels p={','.join(str(q) for q in prob.orders)}, "
                         f"4th-kind Chebyshev order {max(2 * p, 4)}, 1 V-cycle "
                         "per iteration; 'chebyshev' = plain smoother (reference default for magnetostatics), "
                         "'hiptmair' = auxiliary-space smoother (reference default for driven/eigenmode); level 0 assembled to a device CSR matrix like the reference's coarsest level (stand-in for AMS on it): "
                         "Chebyshev-Jacobi order 4 with the plain smoother, 8 Jacobi-PCG iterations with the auxiliary-space one")

    tets = None
    if rank == 0 and world == 1 and not args.no_tets:
        tets = tets_leg(p, args.tet_n)
        for key in ("curlcurl", "curlcurl_mass"):
            if tets and key in tets:
                tets[key]["frac_of_measured_mfma_f64"] = tets[key]["table_TFLOPs"] / measured_mfma
                tets[key]["frac_of_measured_stream"] = tets[key]["algorithmic_GBps"] / measured_stream

    cpu = None
    if rank == 0 and not args.no_cpu:
        cpu = cpu_baseline(p, args.cpu_dofs)
    if world > 1:
        dist.barrier()

    if rank == 0:
        out = {
            "metric": "curl-curl Mult DOF/s + PCG iters/s, p=3 H(curl) 10M DOF @1/2/4/8 GPU",
            "value": value, "unit": "DOF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"ND p={p} curl-curl ParOperator::Mult, O-grid cylinder cavity (hex27, PEC), "
                                   f"{n_global} true dofs total", "order": p, "elements_per_gpu": prob.mesh.ne,
                       "true_dofs_per_gpu": n_true, "global_true_dofs": n_global, "q1d": p + 1,
                       "scaling_mode": "weak: one z-slab of the cylinder per GPU, same element count per GPU",
                       "parallelism": f"element partition x{world}, RCCL halo (P / P^T) + allreduce dots"},
            "roofline": roofline, "cpu_baseline": cpu, "pcg": pcg, "tets_mfma": tets, "setup_s": t_setup,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
