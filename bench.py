#!/usr/bin/env python
"""bench.py — the hot path on MI355X: Nedelec p=3 curl-curl `ParOperator::Mult` throughput and PCG
iterations/s on a ~10M-dof cylinder cavity (BASELINE.json metric), one JSON line on rank 0.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL over xGMI)

A "step" is one `ParOperator::Mult` (BC masking + P + fused E-B-D-B^T-E^T kernel + P^T) of the
curl-curl operator over the whole vector.  `value` = true dofs processed per second by all ranks.
Scaling is strong by default (BASELINE.json: "10M DOF @1/2/4/8 GPU"): the same ~10M-dof cylinder is cut
into N z-slabs, one per GPU; `--scaling weak` gives every rank its own ~10M-dof slab instead.  Extra
keys: `roofline` (element kernel + E^T run gather, algorithmic bytes of SURVEY.md 8(d) / HIP-event
time), `cpu_baseline` (the oracle's C port of the reference's dense-table CPU path on a bounded sample,
apply and PCG + p-multigrid), `parity` (this run's device results against the oracle), `pcg`
(iterations/s of PCG + p-multigrid), `p4` (the order-4 operator of BASELINE config 5 at the same size).
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
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--pre-warm", type=int, default=300,
                    help="untimed applies at the end of the set-up (outside the W warm-up steps): the first ~50 ms of a cold GPU "
                         "run 5 %% slower (clock ramp), which would dominate a 100-step measurement")
    ap.add_argument("--order", type=int, default=3)
    ap.add_argument("--dofs", type=float, default=10.0e6,
                    help="target true dofs: of the whole job (strong scaling) or per GPU (weak)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--pcg-iters", type=int, default=50, help="fixed PCG iterations for iterations/s (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-dofs", type=float, default=1.0e6, help="size of the CPU baseline sample")
    ap.add_argument("--cpu-pcg-dofs", type=float, default=2.5e5, help="size of the CPU PCG + p-multigrid sample")
    ap.add_argument("--cpu-pcg-iters", type=int, default=5)
    ap.add_argument("--cpu-pcg-full-iters", type=int, default=1, help="oracle PCG + p-multigrid iterations ON THE BENCH MESH (M2's CPU "
                    "baseline at size; 0 = skip)")
    ap.add_argument("--no-p4", action="store_true", help="skip the order-4 leg")
    ap.add_argument("--no-tets", action="store_true", help="skip the tetrahedral (dense MFMA path) leg")
    ap.add_argument("--tet-n", type=int, default=36, help="cubes per direction of the Kuhn-split tet mesh")
    ap.add_argument("--force-comm", action="store_true",
                    help="create the RCCL communicator even with one rank (exercises the multi-GPU code path)")
    ap.add_argument("--rehearse", type=int, default=0, metavar="N",
                    help="rehearsal of the N-GPU run on ONE GPU: N processes on device 0, torch.distributed over gloo, halo "
                         "exchanges and sums over the peer transport (RCCL refuses several ranks per device); runs the very "
                         "code of `--gpus N` -- partition, halo plans, direct form, every N > 1 leg -- and prints its line "
                         "with \"rehearsal\": true.  Timings are those of N ranks sharing one GPU, not a scaling result.")
    ap.add_argument("--no-nranks-legs", action="store_true", help="N > 1: skip the order-4 and tetrahedral legs")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child processes that measure roofline.traffic")
    return ap.parse_args()


def rehearse(args):
    """Parent of a rehearsal: starts the N rank processes (the same launch contract as torch.distributed.run: RANK,
    LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT in the environment), passes rank 0's line through."""
    import socket
    import subprocess

    n = args.rehearse
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:]]
    for i, a in enumerate(argv):  # drop --rehearse N / --rehearse=N, force --gpus N
        if a == "--rehearse":
            argv[i:i + 2] = []
            break
        if a.startswith("--rehearse="):
            argv[i:i + 1] = []
            break
    for i, a in enumerate(argv):
        if a == "--gpus":
            argv[i:i + 2] = []
            break
        if a.startswith("--gpus="):
            argv[i:i + 1] = []
            break
    argv += ["--gpus", str(n)]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PALACE_AMD_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit(f"rehearsal: rank exit codes {rcs}")


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


_ORACLE_CACHE = {}


def oracle_hex_data(prob, order):
    """The C oracle's inputs for the finest space of a SlabProblem (geometry data, restriction, dense tables), built once per
    problem: several legs check their device results against it at the full size."""
    key = (id(prob), order)
    if key not in _ORACLE_CACHE:
        from oracle import capi
        from oracle import palace_oracle as po
        from tests import util

        capi.build(ref=False)
        nd = prob.spaces[-1]
        off, ori = nd.native_restriction()
        interp, curl = po.nd_hex_dense_tables(order, order + 1, nd.dof_map_native())
        _ORACLE_CACHE.clear()  # (one problem at a time: the geometry data of the 10M-dof mesh is 0.7 GB)
        _ORACLE_CACHE[key] = dict(geom=util.oracle_geom(prob.mesh, order + 1), off=off, ori=ori, interp=interp, curl=curl)
    return _ORACLE_CACHE[key]


def host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return min(cores, 64)  # the element loop stops scaling beyond a socket's worth of threads


def cpu_leg(ctx, prob, order, args):
    """CPU baseline + parity (rank 0, N = 1).  The oracle is the checker and the thing timed as the CPU baseline,
    never part of the device path.

    cpu_baseline: the oracle's C restatement of the reference CPU path (dense [3Q x P] tables, libCEED-style blocked
    E/B/D/B^T/E^T, OpenMP over element ranges) timed on this host's cores on a smaller cylinder of the same family
    (bounded sample, ~10 s), and the oracle PCG + p-multigrid on a yet smaller one (M2's CPU figure).
    parity: the device results of the same inputs against the oracle: curl-curl apply on the sample mesh and on the
    full bench mesh, and the PCG + p-multigrid iterate after a fixed number of iterations."""
    import torch

    from oracle import capi
    from oracle import palace_oracle as po
    from palace_amd import ceed
    from palace_amd.fem.fespace import NDHexSpace
    from palace_amd.fem.mesh import cylinder_for_dofs
    from palace_amd.fem.partition import SlabProblem
    from tests import util

    capi.build(ref=False)
    q1d = order + 1
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 64)  # the element loop stops scaling beyond a socket's worth of threads
    blob = po.CoeffCtx().pack()
    parity = {"tolerance": "operator 1e-12, PCG iterate 1e-8 (relative l2; tests/ hold the same bounds)"}

    def dev_apply(mesh, nd, x):
        g = ceed.GeomFactorData(mesh, q1d)
        op = ceed.curlcurl_operator(g, nd, ceed.coefficient_context(3))
        y = torch.zeros(nd.ndofs, dtype=torch.float64, device="cuda")
        op.mult(torch.from_numpy(x).cuda(), y)
        return y.cpu().numpy()

    # ---- apply: timing on the sample + parity of the device apply on the same mesh and vector
    mesh = cylinder_for_dofs(args.cpu_dofs, order)
    nd = NDHexSpace(mesh, order)
    geom = util.oracle_geom(mesh, q1d)
    off, ori = nd.native_restriction()
    interp, curl = po.nd_hex_dense_tables(order, q1d, nd.dof_map_native())
    x = np.random.default_rng(1).uniform(0, 1, nd.ndofs)
    y = np.zeros(nd.ndofs)
    capi.apply_add(off, ori, interp, curl, geom, capi.QF_HDIV, blob, x, y, threads=cores)  # warm-up
    parity["rel_l2_y"] = _rel(dev_apply(mesh, nd, x), y)
    parity["rel_l2_y_sample"] = f"{nd.ndofs} dofs, {mesh.ne} elements"
    reps, t0 = 0, time.perf_counter()
    while True:
        y[:] = 0.0
        capi.apply_add(off, ori, interp, curl, geom, capi.QF_HDIV, blob, x, y, threads=cores)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 10.0:
            break
    cpu = {"value": nd.ndofs * reps / dt, "unit": "DOF/s", "cores": cores, "kind": "port",
           "sample": f"curl-curl apply, ND p={order}, {mesh.ne} hex27 elements, {nd.ndofs} dofs, {reps} applies "
                     f"in {dt:.1f} s; oracle/oracle_c.c (dense-table libCEED-style CPU path restated), OpenMP"}
    del geom

    # ---- the CPU baseline of SURVEY.md 8(d) proper: the same oracle apply ON THE BENCH MESH (the very operator and mesh of the timed
    # loop), >= 3 timed applies after one warm-up; the first one is also the full-size parity check
    t0 = time.perf_counter()
    fnd = prob.spaces[-1]
    od = oracle_hex_data(prob, order)
    fgeom, foff, fori = od["geom"], od["off"], od["ori"]
    fx = np.random.default_rng(2).uniform(0, 1, fnd.ndofs)
    fy = np.zeros(fnd.ndofs)
    capi.apply_add(foff, fori, interp, curl, fgeom, capi.QF_HDIV, blob, fx, fy, threads=cores)
    dy = torch.zeros(fnd.ndofs, dtype=torch.float64, device="cuda")
    prob.local_curlcurl.mult(torch.from_numpy(fx).cuda(), dy)
    parity["rel_l2_y_full"] = _rel(dy.cpu().numpy(), fy)
    parity["rel_l2_y_full_size"] = f"{fnd.ndofs} dofs, {prob.mesh.ne} elements ({time.perf_counter() - t0:.1f} s of oracle work)"
    freps, t0 = 0, time.perf_counter()
    while True:
        fy[:] = 0.0
        capi.apply_add(foff, fori, interp, curl, fgeom, capi.QF_HDIV, blob, fx, fy, threads=cores)
        freps += 1
        fdt = time.perf_counter() - t0
        if freps >= 3 and fdt > 3.0 or fdt > 20.0:
            break
    cpu["sample_1M"] = {"value": cpu["value"], "sample": cpu["sample"]}
    cpu["value"] = fnd.ndofs * freps / fdt
    cpu["omp_threads"] = cores
    cpu["sample"] = (f"curl-curl apply on the BENCH mesh itself: ND p={order}, {prob.mesh.ne} hex27 elements, {fnd.ndofs} dofs, {freps} applies in "
                     f"{fdt:.1f} s on {cores} OpenMP threads (OMP_NUM_THREADS is set by the call; host has {os.cpu_count()} logical cores); "
                     "oracle/oracle_c.c (dense-table libCEED-style CPU path restated); the cache-resident 1M-dof sample of rounds 1-4 "
                     "is kept as sample_1M")
    del fgeom, fx, fy, dy

    # ---- M2 on the CPU: oracle PCG + p-multigrid (plain Chebyshev, Jacobi-PCG(8) on level 0), timed, and the
    # device iterate of the same configuration (same eigenvalue estimates) against it
    if args.cpu_pcg_iters > 0:
        its = args.cpu_pcg_iters
        sp = SlabProblem(ctx, 0, 1, order, args.cpu_pcg_dofs)
        solver, b, xs = sp.pcg_gmg_solver(max_it=its, hiptmair=False, coarse="cg")
        solver.mult(b, xs)
        xd = xs.cpu().numpy()
        gmg = sp.last_gmg
        nl = len(sp.spaces)
        ogeom = util.oracle_geom(sp.mesh, q1d)
        cm, bm = util.make_ctx("scalar")
        cc, bc = util.make_ctx("identity")
        blob2 = np.concatenate([bm, bc])
        oA = [util.FastParOperatorOracle(sx, ogeom, "hdivmass", blob2, sx.ess_dofs(), q1d, cm, cc) for sx in sp.spaces]
        oP = [po.InterpOracle(c.elem_dof_lex, c.elem_sign_lex, f.elem_dof_lex, f.elem_sign_lex, c.ndofs, f.ndofs,
                              po.nd_hex_interp_lex(c.p, f.p)) for c, f in zip(sp.spaces[:-1], sp.spaces[1:])]
        ko = max(2 * order, 4)
        sm = [None] + [po.ChebyshevOracle(oA[l], ko, lambda_max=gmg.gmg_lambda_max(l)) for l in range(1, nl)]
        d0 = 1.0 / oA[0].diagonal()
        coarse = lambda r: po.pcg(oA[0].mult, r, lambda v: d0 * v, rel_tol=1e-2, max_it=8)[0]  # noqa: E731
        oB = po.GMGOracle(oA, [(q.mult, q.mult_transpose) for q in oP], sm, coarse, [sx.ess_dofs() for sx in sp.spaces])
        n = sp.spaces[-1].ndofs
        ob = oA[-1].mult(np.ones(n))
        ob[sp.spaces[-1].ess_dofs()] = 0.0
        t0 = time.perf_counter()
        xo, it, hist = po.pcg(oA[-1].mult, ob, oB.mult, rel_tol=0.0, max_it=its)
        dt = time.perf_counter() - t0
        parity["rel_l2_pcg"] = _rel(xd, xo)
        parity["rel_l2_pcg_sample"] = (f"{n} dofs, iterate after {it} PCG + p-multigrid iterations (plain Chebyshev order {ko}, "
                                       "Jacobi-PCG(8) on level 0), device eigenvalue estimates handed to the oracle")
        cpu["pcg_iters_per_s"] = it / dt
        cpu["pcg_sample"] = (f"oracle PCG + p-multigrid on K+M, {n} dofs, {sp.mesh.ne} elements, {it} iterations in {dt:.1f} s "
                             "(local applies oracle/oracle_c.c with OpenMP, the rest numpy)")
        sp._keep.clear()

    # ---- M2's CPU figure AT THE BENCH SIZE (round 6): the same oracle loop on the bench mesh itself, --cpu-pcg-full-iters
    # iterations (default 1: one iteration is ~17 fine-level and ~14 order-2 applies of the C oracle on `cores` OpenMP threads;
    # the vector work is numpy, single-threaded).  The operator diagonals and eigenvalue estimates the smoothers need are taken
    # from the device objects (set-up, outside the timed region: the numpy diagonal of 125k dense element matrices takes
    # minutes); the device iterate after the same number of iterations is checked against the oracle's.
    if args.cpu_pcg_full_iters > 0:
        its = args.cpu_pcg_full_iters
        solver, b, xs = prob.pcg_gmg_solver(max_it=its, hiptmair=False, coarse="cg")
        solver.mult(b, xs)
        xd = xs.cpu().numpy()
        gmg, dA = prob.last_gmg, prob.last_A
        nl = len(prob.spaces)
        cm, bm = util.make_ctx("scalar")
        cc, bc = util.make_ctx("identity")
        blob2 = np.concatenate([bm, bc])
        ogeom = oracle_hex_data(prob, order)["geom"]

        class _Level:  # ParOperatorOracle's Mult with the C apply on `cores` threads; diagonal handed over from the device
            def __init__(self, sx, dev):
                self.off, self.ori = sx.native_restriction()
                self.off = np.ascontiguousarray(self.off, dtype=np.int32)
                self.tab = po.nd_hex_dense_tables(sx.p, q1d, sx.dof_map_native())
                self.ess, self.n = sx.ess_dofs().astype(np.int64), sx.ndofs
                d = torch.empty(sx.ndofs, dtype=torch.float64, device="cuda")
                dev.assemble_diagonal(d)
                self._diag = d.cpu().numpy()

            def mult(self, v):
                tv = v.copy()
                tv[self.ess] = 0.0
                out = np.zeros(self.n)
                capi.apply_add(self.off, self.ori, self.tab[0], self.tab[1], ogeom, capi.QF_HDIVMASS, blob2, tv, out, threads=cores)
                out[self.ess] = v[self.ess]
                return out

            def diagonal(self):
                return self._diag

        oA = [_Level(sx, da) for sx, da in zip(prob.spaces, dA)]
        oP = [po.InterpOracle(c.elem_dof_lex, c.elem_sign_lex, f.elem_dof_lex, f.elem_sign_lex, c.ndofs, f.ndofs,
                              po.nd_hex_interp_lex(c.p, f.p)) for c, f in zip(prob.spaces[:-1], prob.spaces[1:])]
        ko = max(2 * order, 4)
        sm = [None] + [po.ChebyshevOracle(oA[l], ko, lambda_max=gmg.gmg_lambda_max(l)) for l in range(1, nl)]
        d0 = 1.0 / oA[0].diagonal()
        coarse = lambda r: po.pcg(oA[0].mult, r, lambda v: d0 * v, rel_tol=1e-2, max_it=8)[0]  # noqa: E731
        oB = po.GMGOracle(oA, [(q.mult, q.mult_transpose) for q in oP], sm, coarse, [sx.ess_dofs() for sx in prob.spaces])
        n = prob.spaces[-1].ndofs
        ob = oA[-1].mult(np.ones(n))
        ob[prob.spaces[-1].ess_dofs()] = 0.0
        t0 = time.perf_counter()
        xo, it, hist = po.pcg(oA[-1].mult, ob, oB.mult, rel_tol=0.0, max_it=its)
        dt = time.perf_counter() - t0
        parity["rel_l2_pcg_full"] = _rel(xd, xo)
        parity["rel_l2_pcg_full_size"] = f"{n} dofs (the bench mesh), iterate after {it} PCG + p-multigrid iteration(s)"
        cpu["pcg_iters_per_s_sample"] = cpu.get("pcg_iters_per_s")
        cpu["pcg_iters_per_s"] = it / dt
        cpu["pcg_full_size_sample"] = (f"oracle PCG + p-multigrid on K+M ON THE BENCH MESH: {n} dofs, {prob.mesh.ne} elements, {it} iteration(s) in "
                                       f"{dt:.1f} s; local applies oracle/oracle_c.c on {cores} OpenMP threads, vector work and transfers numpy "
                                       "(1 thread); operator diagonals and eigenvalue estimates handed over from the device (set-up, untimed)")
        prob._keep.clear()
    return cpu, parity


def p4_leg(ctx, dofs, reps=200, pcg_iters=20, parity=True, big_dofs=40.0e6):
    """Order 4 (BASELINE config 5's element) on a cylinder of the same size, N = 1: `ParOperator::Mult` of curl-curl (PEC rows
    fused) and `ceed::Operator::Mult` of curl-curl + mass through the five-point streaming kernel (pa_nd_hex_stream5.hip), the
    device result against the C oracle at this size, and PCG + p-multigrid (p = 1..4, plain Chebyshev) iterations/s."""
    import torch

    from palace_amd import ceed
    from palace_amd.fem.partition import SlabProblem

    p = 4
    prob = SlabProblem(ctx, 0, 1, p, dofs, levels=True)
    nd, mesh, geom = prob.spaces[-1], prob.mesh, prob.geom
    mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
    ident = ceed.coefficient_context(3)
    K = prob.curlcurl_par_operator()
    KM = ceed.curlcurlmass_operator(geom, nd, mass, ident)
    x = torch.rand(nd.ndofs, dtype=torch.float64, device="cuda")
    y = torch.zeros_like(x)
    lib = ceed._lib.load()
    out = {"workload": f"ND p=4 hexahedra, {mesh.ne} elements, {nd.ndofs} dofs, P=300, Q=125", "dofs": nd.ndofs,
           "streaming_kernel": bool(lib.pa_op_streams(prob.local_curlcurl.handle)) and bool(lib.pa_op_streams(KM.handle)),
           "bytes_formula": "NE*(Q*11*8 + P*5) + 16*N_L (SURVEY.md 8d, G=11): 12 500 B / element"}
    for name, fn, op in (("curlcurl", lambda: K.mult(x, y), prob.local_curlcurl), ("curlcurl_mass", lambda: KM.mult(x, y), KM)):
        with torch.cuda.stream(ctx.torch_stream):
            for _ in range(30):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        alg = op.algorithmic_bytes()
        out[name] = {"ms": ms, "dof_per_s": nd.ndofs / (ms * 1e-3), "algorithmic_GBps": alg / ms / 1e6,
                     "hbm_frac": alg / ms / 1e6 / HBM_PEAK_GBS}
    if parity:  # the local curl-curl apply at this size against the C oracle (dense [3Q x P] tables), one oracle apply
        from oracle import capi
        from oracle import palace_oracle as po
        from tests import util

        capi.build(ref=False)
        t0 = time.perf_counter()
        cores = host_cores()
        od = oracle_hex_data(prob, p)
        og, off, ori, interp, curl = od["geom"], od["off"], od["ori"], od["interp"], od["curl"]
        hx = np.random.default_rng(4).uniform(0, 1, nd.ndofs)
        hy = np.zeros(nd.ndofs)
        capi.apply_add(off, ori, interp, curl, og, capi.QF_HDIV, po.CoeffCtx().pack(), hx, hy, threads=cores)
        dy = torch.empty(nd.ndofs, dtype=torch.float64, device="cuda")
        prob.local_curlcurl.mult(torch.from_numpy(hx).cuda(), dy)
        out["parity"] = {"rel_l2_y_full": _rel(dy.cpu().numpy(), hy), "tolerance": 1e-12,
                         "size": f"{nd.ndofs} dofs, {mesh.ne} elements ({time.perf_counter() - t0:.1f} s of oracle work)"}
        del hx, hy, dy
    cl = complex_leg(ctx, prob, parity=parity)  # the complex form of the five-point kernel
    out["complex"] = {k: cl[k] for k in ("one_pass", "ms", "complex_dof_per_s", "hbm_frac", "parity") if k in cl}
    if pcg_iters > 0:
        solver, b, xs = prob.pcg_gmg_solver(max_it=pcg_iters, hiptmair=False, coarse="chebyshev")
        solver.mult(b, xs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.mult(b, xs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = solver.stats()
        out["pcg_chebyshev"] = {"iters_per_s": st["iterations"] / dt, "iterations": st["iterations"], "seconds": dt,
                                "levels": ",".join(str(q) for q in prob.orders),
                                "final_rel_res": st["final_res"] / st["initial_res"]}
        prob._keep.clear()
        solver, b, xs = prob.pcg_gmg_solver(max_it=400, rel_tol=1e-8, hiptmair=False, coarse="chebyshev")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.mult(b, xs)
        torch.cuda.synchronize()
        st = solver.stats()
        out["pcg_chebyshev"].update({"iterations_to_1e-8": st["iterations"], "seconds_to_1e-8": time.perf_counter() - t0,
                                     "converged": st["converged"]})
        prob._keep.clear()
    if big_dofs:
        # BASELINE config 5's SIZE on one GPU (the N = 1 anchor of the 8-GPU configuration): ~40M dofs at order 4 fit 288 GB
        # many times over; curl-curl ParOperator::Mult only, same kernels, same byte formula
        del K, KM, x, y, prob
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        big = SlabProblem(ctx, 0, 1, p, big_dofs, levels=False)
        Kb = big.curlcurl_par_operator()
        nb = big.n_true[-1]
        xb = torch.rand(nb, dtype=torch.float64, device="cuda")
        yb = torch.zeros_like(xb)
        setup_s = time.perf_counter() - t0
        with torch.cuda.stream(ctx.torch_stream):
            for _ in range(20):
                Kb.mult(xb, yb)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                Kb.mult(xb, yb)
            e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        alg = big.local_curlcurl.algorithmic_bytes()
        out["config5_size_one_gpu"] = {"workload": f"ND p=4, {big.mesh.ne} hex27 elements, {nb} true dofs (BASELINE config 5's size on ONE GPU)",
                                       "ms": ms, "dof_per_s": nb / (ms * 1e-3), "algorithmic_GBps": alg / ms / 1e6,
                                       "hbm_frac": alg / ms / 1e6 / HBM_PEAK_GBS, "host_setup_s": setup_s}
        del Kb, xb, yb, big
        torch.cuda.empty_cache()
    return out


def complex_leg(ctx, prob, reps=50, parity=True, aniso=False):
    """BASELINE config 3's operator shape on the bench mesh, N = 1: y = (K - w^2 eps M + i w sigma M) x through
    ComplexParOperator::Mult -- both parts in one pass over the element data (pa_op_mult_complex).
    hbm_frac: the algorithmic bytes of ONE pass over the element data (SURVEY.md 8d with G = 11) plus the second part of x and
    y, over the measured time; parity: the device result against the C oracle's four real applies at this size.
    aniso: the materials of the reference's driven example (examples/cpw/cpw_lumped_uniform.json:24-28, sapphire: permittivity
    [9.3, 9.3, 11.5], loss tangent [3.0e-5, 3.0e-5, 8.6e-5]) rotated out of the mesh axes -- the packed-D form of the complex kernel
    (two operators' symmetric D at every point: 12 + 6 doubles instead of the metric form's 7)."""
    import torch

    from palace_amd import ceed, linalg

    nd = prob.spaces[-1]
    mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([-2.08 * 0.3])])
    cond = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([0.05])])
    if aniso:
        c, s_ = np.cos(0.3), np.sin(0.3)
        R = np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]]) @ np.array([[1.0, 0.0, 0.0], [0.0, c, -s_], [0.0, s_, c]])
        eps = R @ np.diag([9.3, 9.3, 11.5]) @ R.T
        loss = R @ np.diag([9.3 * 3.0e-5, 9.3 * 3.0e-5, 11.5 * 8.6e-5]) @ R.T
        eps, loss = 0.5 * (eps + eps.T), 0.5 * (loss + loss.T)  # (exactly symmetric: the packed form is chosen on an exact test)
        mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[-0.3 * eps])
        cond = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[0.3 * loss])
    Ar = ceed.curlcurlmass_operator(prob.geom, nd, mass, ceed.coefficient_context(3))
    Ai = ceed.ndmass_operator(prob.geom, nd, cond)
    A = linalg.ComplexParOperator(ctx, Ar, Ai, prob.ess[-1], linalg.DIAG_ONE)
    n = nd.ndofs
    xr, xi = (torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(2))
    yr, yi = torch.empty_like(xr), torch.empty_like(xr)
    for _ in range(10):
        A.mult(xr, xi, yr, yi)
    with torch.cuda.stream(ctx.torch_stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            A.mult(xr, xi, yr, yi)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fused = bool(ceed._lib.load().pa_op_complex_fused(Ar.handle, Ai.handle))
    alg = Ar.algorithmic_bytes() + 16.0 * n
    out = {"workload": f"ComplexParOperator::Mult, A = (K - w^2 eps M) + i w sigma M, ND p={nd.p}, {n} complex dofs" +
                       (", anisotropic eps and sigma (sapphire tensors, rotated)" if aniso else ""),
           "one_pass": fused, "ms": ms, "complex_dof_per_s": n / (ms * 1e-3), "algorithmic_GBps": alg / ms / 1e6,
           "hbm_frac": alg / ms / 1e6 / HBM_PEAK_GBS,
           "bytes_formula": "NE*(Q*11*8 + P*5) + 32*N_L: one pass over the element data, both parts of x and y"}
    if parity:
        from oracle import capi

        t0 = time.perf_counter()
        od = oracle_hex_data(prob, nd.p)
        cores = host_cores()
        ess = prob.ess[-1].astype(np.int64)
        hr, hi = xr.cpu().numpy(), xi.cpu().numpy()
        mr, mi = hr.copy(), hi.copy()
        mr[ess] = 0.0
        mi[ess] = 0.0
        blob_r = np.concatenate([mass, ceed.coefficient_context(3)])

        def oapply(qf, blob, v):
            w = np.zeros(n)
            capi.apply_add(od["off"], od["ori"], od["interp"], od["curl"], od["geom"], qf, blob, v, w, threads=cores)
            return w

        wr = oapply(capi.QF_HDIVMASS, blob_r, mr) - oapply(capi.QF_HCURL, cond, mi)
        wi = oapply(capi.QF_HDIVMASS, blob_r, mi) + oapply(capi.QF_HCURL, cond, mr)
        wr[ess], wi[ess] = hr[ess], hi[ess]  # DIAG_ONE (rap.cpp:450-457)
        A.mult(xr, xi, yr, yi)
        d = np.concatenate([yr.cpu().numpy() - wr, yi.cpu().numpy() - wi])
        out["parity"] = {"rel_l2_y_full": float(np.linalg.norm(d) / np.linalg.norm(np.concatenate([wr, wi]))), "tolerance": 1e-12,
                         "size": f"{n} complex dofs ({time.perf_counter() - t0:.1f} s of oracle work: four real applies of the C oracle)"}
    return out


def h1_leg(ctx, prob, order=2, reps=200, pcg_iters=50):
    """BASELINE config 4's system on the same cylinder, N = 1: H1 order-2 diffusion (eps grad u, grad v) -- `ParOperator::Mult`
    and PCG + p-multigrid (levels 1, 2; plain Chebyshev smoothers) with the native algebraic V-cycle on the assembled order-1
    level, where the reference calls BoomerAMG."""
    import torch

    out = {}
    for coarse in ("amg", "chebyshev"):
        solver, b, xs = prob.h1_pcg_gmg_solver(order=order, max_it=pcg_iters, coarse=coarse)
        A = prob.h1_fine
        n = b.numel()
        if "apply" not in out:
            xx, yy = torch.rand_like(b), torch.empty_like(b)
            with torch.cuda.stream(ctx.torch_stream):
                for _ in range(30):
                    A.mult(xx, yy)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    A.mult(xx, yy)
                e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            out["workload"] = f"H1 p={order} hexahedra, {prob.mesh.ne} elements, {n} dofs, diffusion (eps_r = 2.08), Dirichlet boundary"
            out["dofs"] = n
            alg = A.local.algorithmic_bytes()
            out["apply"] = {"ms": ms, "dof_per_s": n / (ms * 1e-3), "algorithmic_GBps": alg / ms / 1e6,
                            "hbm_frac": alg / ms / 1e6 / HBM_PEAK_GBS,
                            "bytes_formula": "NE*(Q*11*8 + P*4) + 16*N_L (SURVEY.md 8d, G = 11; Q = 27 at order 2)"}
            # the local diffusion apply at this size against the numpy oracle (dense [3Q x P] gradient table, f_apply_hcurl_33)
            from oracle import palace_oracle as po
            from tests import util

            t0 = time.perf_counter()
            h1 = prob._keep[-1][0][-1]
            q1 = order + 1
            interp, grad = po.h1_hex_dense_tables(order, q1)
            orc = po.CeedOperatorOracle(h1.ndofs, h1.elem_dof_lex, None, interp, grad, util.oracle_geom(prob.mesh, q1), po.QF_HCURL,
                                        po.CoeffCtx(attr_mat=[0], mat_coeff=[np.array([2.08])]), None, vector_fe=False)
            hx = np.random.default_rng(8).uniform(-1, 1, h1.ndofs)
            hy = orc.apply_add(hx, np.zeros(h1.ndofs))
            dy = torch.empty(h1.ndofs, dtype=torch.float64, device="cuda")
            A.local.mult(torch.from_numpy(hx).cuda(), dy)
            out["parity"] = {"rel_l2_y_full": _rel(dy.cpu().numpy(), hy), "tolerance": 1e-12,
                             "size": f"{h1.ndofs} dofs, {prob.mesh.ne} elements ({time.perf_counter() - t0:.1f} s of oracle work)"}
            del orc, hx, hy, dy
        solver.mult(b, xs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.mult(b, xs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = solver.stats()
        entry = {"iters_per_s": st["iterations"] / dt, "iterations": st["iterations"], "seconds": dt}
        solver, b, xs = prob.h1_pcg_gmg_solver(order=order, max_it=400, rel_tol=1e-8, coarse=coarse)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.mult(b, xs)
        torch.cuda.synchronize()
        st = solver.stats()
        entry.update({"iterations_to_1e-8": st["iterations"], "seconds_to_1e-8": time.perf_counter() - t0,
                      "converged": st["converged"]})
        out["pcg_" + coarse] = entry
        prob._keep.clear()
    return out


def cpw_iso_leg(order=3, refine=1, reps=20, ab=False):
    """(rounds 3-4's form of the leg, kept for continuity: surrogate isotropic materials, white-noise right-hand side)
    BASELINE config 3 on the reference's own mesh: examples/cpw/mesh/cpw_lumped_0.msh (committed as tests/golden/cpw_mesh.npz,
    14 628 tetrahedra) uniformly refined `refine` times, order-3 Nedelec tetrahedra, the driven-type complex system
    A = K - k0^2 eps_r (1 - i tan d) M at 16 GHz, FGMRES + Hiptmair p-multigrid (p = 1, 2, 3) with the native AMS cycle on the
    assembled order-1 level: complex applies/s, iterations to 1e-8 and iterations/s; the real part against the numpy oracle."""
    import torch

    from palace_amd import linalg
    from palace_amd.fem import tet
    from palace_amd.fem.tetproblem import TetProblem

    d = np.load(os.path.join(ROOT, "tests", "golden", "cpw_mesh.npz"))
    mesh = tet.TetMesh(d["verts"], d["tets"], d["attr"], bdr_tris=d["bdr_tris"], bdr_attr=d["bdr_attr"])
    for _ in range(refine):
        mesh = tet.refine_uniform(mesh)
    ctx = linalg.Context()
    prob = TetProblem(ctx, mesh, order)
    bt = np.sort(np.asarray(mesh.bdr_tris, dtype=np.int64), axis=1)
    pec = bt[np.isin(mesh.bdr_attr, (4, 13))]  # far field and the metal trace; the port faces stay natural
    fv = mesh.face_verts
    nvt = mesh.nv
    key = lambda f: (f[:, 0] * nvt + f[:, 1]) * nvt + f[:, 2]
    order_f = np.argsort(key(fv))
    fmask = np.zeros(fv.shape[0], dtype=bool)
    fmask[order_f[np.searchsorted(key(fv)[order_f], key(pec))]] = True
    k0 = 2 * np.pi * 16.0e9 * 1.0e-6 / 299792458.0
    # reference defaults: Chebyshev order max(2p, 4), no restart before max_it (iodata.cpp:533-564: max_size = max_it)
    sys_ = prob.driven_solver(fmask, k0, eps=[1.0, 11.7], tand=[0.0, 0.05], coarse="ams", cheby_order=max(2 * order, 4),
                              max_it=600, restart=600)
    A, S, ess, n = sys_["A"], sys_["solver"], sys_["ess"], sys_["n"]
    rng = np.random.default_rng(4)
    b = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    b[ess] = 0.0
    br, bi = torch.from_numpy(b.real.copy()).cuda(), torch.from_numpy(b.imag.copy()).cuda()
    out = {"workload": f"examples/cpw mesh refined x{refine}: {mesh.ne} tetrahedra, ND p={order}, {n} complex dofs, 16 GHz, "
                       "eps_r = (1, 11.7), tan d = (0, 0.05), white-noise right-hand side; FGMRES (no restart) + Hiptmair p-multigrid (p = 1..3, "
                       "Chebyshev order 6) + native AMS on level 0 (with the Jacobi-PCG stand-in there the solve does not converge in 600 "
                       "iterations on this mesh: scripts/cpw_explore.py)",
           "complex_dofs": n}
    yr, yi = torch.empty_like(br), torch.empty_like(br)
    with torch.cuda.stream(ctx.torch_stream):
        for _ in range(5):
            A.mult(br, bi, yr, yi)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            A.mult(br, bi, yr, yi)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = sys_["Kr"].algorithmic_bytes() + 16.0 * n
    out["complex_apply"] = {"ms": ms, "complex_dof_per_s": n / (ms * 1e-3), "algorithmic_GBps": alg / ms / 1e6,
                            "hbm_frac": alg / ms / 1e6 / HBM_PEAK_GBS,
                            "bytes_formula": "NE*(Q*11*8 + P*7) + 32*N_L: one pass over the element data, both parts of x and y"}
    xr, xi = torch.zeros_like(br), torch.zeros_like(br)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    S.mult(br, bi, xr, xi)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = S.stats()
    A.mult(xr, xi, yr, yi)
    res = float(torch.sqrt(((yr - br) ** 2 + (yi - bi) ** 2).sum()) / torch.sqrt((br ** 2 + bi ** 2).sum()))
    out["fgmres"] = {"iterations_to_1e-8": st["iterations"], "seconds": dt, "iters_per_s": st["iterations"] / dt,
                     "converged": st["converged"], "true_rel_residual": res, "orthogonalization": "MGS (the reference's default), coefficients on the device: one host synchronisation per column (orthog.hip)"}
    # the same solve with the batched orthogonalisation (OrthogonalizeColumnCGS2, linalg/orthog.hpp:57-89: two reductions per step
    # instead of j + 1): same preconditioner object
    try:
        if not ab:
            raise StopIteration
        S2 = linalg.ComplexParGmres(ctx, A, sys_["B"], rel_tol=1e-8, max_it=600, restart=600, flexible=True, orthogonalization="CGS2")
        xr2, xi2 = torch.zeros_like(br), torch.zeros_like(br)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        S2.mult(br, bi, xr2, xi2)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        st2 = S2.stats()
        dx = float(torch.sqrt(((xr2 - xr) ** 2 + (xi2 - xi) ** 2).sum()) / torch.sqrt((xr ** 2 + xi ** 2).sum()))
        out["fgmres_cgs2"] = {"iterations_to_1e-8": st2["iterations"], "seconds": dt2, "iters_per_s": st2["iterations"] / dt2,
                              "converged": st2["converged"], "rel_diff_of_the_solution_from_the_MGS_solve": dx}
        del S2, xr2, xi2
    except StopIteration:
        pass
    except Exception as exc:  # noqa: BLE001
        out["fgmres_cgs2"] = {"error": f"{type(exc).__name__}: {exc}"}
    # A / B: the same MGS solve with the host driving every inner product (rounds 1-4: one synchronisation per basis vector)
    try:
        if not ab:
            raise StopIteration
        linalg.Context.set_device_orthogonalization(False)
        xr3, xi3 = torch.zeros_like(br), torch.zeros_like(br)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        S.mult(br, bi, xr3, xi3)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t0
        st3 = S.stats()
        dx3 = float(torch.sqrt(((xr3 - xr) ** 2 + (xi3 - xi) ** 2).sum()) / torch.sqrt((xr ** 2 + xi ** 2).sum()))
        out["fgmres_host_driven_mgs"] = {"iterations_to_1e-8": st3["iterations"], "seconds": dt3, "iters_per_s": st3["iterations"] / dt3,
                                         "rel_diff_of_the_solution_from_the_device_chained_solve": dx3}
        # ... and the device-chained form once more on the same solver object: like the host-driven solve above it finds the basis
        # vectors allocated (the first solve of a solver allocates them on the way) -- the like-for-like pair of the A / B
        linalg.Context.set_device_orthogonalization(True)
        xr3.zero_(), xi3.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        S.mult(br, bi, xr3, xi3)
        torch.cuda.synchronize()
        dt4 = time.perf_counter() - t0
        out["fgmres_second_solve"] = {"iterations_to_1e-8": S.stats()["iterations"], "seconds": dt4, "iters_per_s": S.stats()["iterations"] / dt4,
                                      "note": "device-chained MGS, basis vectors already allocated: compare with fgmres_host_driven_mgs"}
        del xr3, xi3
    except StopIteration:
        pass
    except Exception as exc:  # noqa: BLE001
        out["fgmres_host_driven_mgs"] = {"error": f"{type(exc).__name__}: {exc}"}
    finally:
        linalg.Context.set_device_orthogonalization(True)
    # the real-part operator at this size against the numpy oracle (one oracle apply)
    from oracle import palace_oracle as po

    t0 = time.perf_counter()
    nd = prob.spaces[-1]
    interp, curl = nd.elem.tables(prob.pts)
    J = mesh.jacobians(prob.pts)
    og = po.build_geom_factor_33(mesh.attr.astype(np.float64), prob.wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9))
    okw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
    oc = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[np.array([-k0 ** 2 * 1.0]), np.array([-k0 ** 2 * 11.7])])
    orc = po.CeedOperatorOracle(n, nd.offsets, None, interp, curl, og, po.QF_HDIVMASS, oc, po.CoeffCtx(), **okw)
    hx = rng.uniform(0, 1, n)
    hy = orc.apply_add(hx, np.zeros(n))
    dy = torch.empty_like(br)
    sys_["Kr"].mult(torch.from_numpy(hx).cuda(), dy)
    out["parity"] = {"rel_l2_y_full": _rel(dy.cpu().numpy(), hy), "tolerance": 1e-12,
                     "size": f"{n} dofs, {mesh.ne} tets ({time.perf_counter() - t0:.1f} s of oracle work)"}
    # the COMPLEX operator: ComplexParOperator::Mult (one pass on the device) against the oracle's real and imaginary operators
    # applied to both parts, essential rows as rap.cpp:450-457; and the FGMRES solution in the ORACLE's operator: the residual the
    # reference's own arithmetic assigns to the device's answer
    t0 = time.perf_counter()
    oi = po.CoeffCtx(attr_mat=[0, 1], mat_coeff=[np.array([k0 ** 2 * 1.0 * 0.0]), np.array([k0 ** 2 * 11.7 * 0.05])])
    orci = po.CeedOperatorOracle(n, nd.offsets, None, interp, curl, og, po.QF_HCURL, oi, **okw)

    def o_complex(vr, vi):
        mr, mi = vr.copy(), vi.copy()
        mr[ess], mi[ess] = 0.0, 0.0
        z = np.zeros(n)
        wr = orc.apply_add(mr, z.copy()) - orci.apply_add(mi, z.copy())
        wi = orc.apply_add(mi, z.copy()) + orci.apply_add(mr, z.copy())
        wr[ess], wi[ess] = vr[ess], vi[ess]
        return wr, wi

    cr, ci = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    wr, wi = o_complex(cr, ci)
    A.mult(torch.from_numpy(cr).cuda(), torch.from_numpy(ci).cuda(), yr, yi)
    dd = np.concatenate([yr.cpu().numpy() - wr, yi.cpu().numpy() - wi])
    out["parity"]["complex_apply_rel_l2"] = float(np.linalg.norm(dd) / np.linalg.norm(np.concatenate([wr, wi])))
    sr, si = o_complex(xr.cpu().numpy(), xi.cpu().numpy())
    rr = np.concatenate([sr - b.real, si - b.imag])
    out["parity"]["fgmres_solution_rel_residual_in_the_oracle_operator"] = float(np.linalg.norm(rr) / np.linalg.norm(np.concatenate([b.real, b.imag])))
    out["parity"]["complex_size"] = f"{n} complex dofs ({time.perf_counter() - t0:.1f} s of oracle work: eight real applies of the numpy oracle)"
    return out


def cpw_leg(order=3, refine=1, reps=20, freq_ghz=17.0):
    """BASELINE config 3 AS THE REFERENCE DEFINES IT (round 5): examples/cpw/cpw_lumped_uniform.json on its own mesh
    (cpw_lumped_0.msh, committed as tests/golden/cpw_mesh.npz) uniformly refined `refine` times, order-3 Nedelec tetrahedra:
    sapphire tensors (eps, mu, tan d), first-order absorbing boundary and four resistive lumped ports as surface f_apply_hcurl_32
    terms of the imaginary part, PEC trace, uniform excitation of port 1, the 17 GHz point of the reference's sweep.  FGMRES (no
    restart) + Hiptmair p-multigrid + native AMS; complex applies/s, iterations to 1e-8 and iterations/s, the A / B of the
    orthogonalisation forms, the complex operator against the oracle at this size, S[j][1] of the device solution beside the
    reference's regression values (which belong to the unrefined order-2 discretisation: tests/test_cpw_gpu.py checks those)."""
    import torch

    from palace_amd import linalg
    from palace_amd.fem import tet
    from palace_amd.fem.tetproblem import CPW_LUMPED_UNIFORM, DrivenReferenceSystem, TetProblem

    d = np.load(os.path.join(ROOT, "tests", "golden", "cpw_mesh.npz"))
    mesh = tet.TetMesh(d["verts"], d["tets"], d["attr"], bdr_tris=d["bdr_tris"], bdr_attr=d["bdr_attr"])
    for _ in range(refine):
        mesh = tet.refine_uniform(mesh)
    t0 = time.perf_counter()
    ctx = linalg.Context()
    prob = TetProblem(ctx, mesh, order)
    ds = DrivenReferenceSystem(prob, freq_ghz, CPW_LUMPED_UNIFORM, rel_tol=1e-8, max_it=600)
    n, A, S = ds.n, ds.A, ds.solver
    br, bi = ds.excitation(1)
    out = {"materials": "reference", "workload": f"examples/cpw/cpw_lumped_uniform.json: mesh refined x{refine} = {mesh.ne} tetrahedra, ND p={order}, {n} complex "
                       f"dofs, {freq_ghz} GHz; sapphire eps = (9.3, 9.3, 11.5), tan d = (3, 3, 8.6)e-5, mu = (0.99999975, 0.99999975, 0.99999979); "
                       f"first-order absorbing boundary ({int((ds.sattr <= 2).sum())} faces), 4 lumped ports of 56.02 Ohm ({int((ds.sattr > 2).sum())} faces), "
                       "PEC trace; excitation: port 1 (uniform); FGMRES (no restart) + Hiptmair p-multigrid (p = 1..3, Chebyshev order 6) + native AMS on level 0",
           "complex_dofs": n, "setup_s": time.perf_counter() - t0}
    xr, xi = torch.rand(n, dtype=torch.float64, device="cuda"), torch.rand(n, dtype=torch.float64, device="cuda")
    yr, yi = torch.empty_like(xr), torch.empty_like(xr)
    with torch.cuda.stream(ctx.torch_stream):
        for _ in range(5):
            A.mult(xr, xi, yr, yi)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            A.mult(xr, xi, yr, yi)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = ds.Ar.algorithmic_bytes() + 16.0 * n
    out["complex_apply"] = {"ms": ms, "complex_dof_per_s": n / (ms * 1e-3), "algorithmic_GBps": alg / ms / 1e6,
                            "hbm_frac": alg / ms / 1e6 / HBM_PEAK_GBS, "one_pass": int(A.fused()) if hasattr(A, "fused") else None,
                            "bytes_formula": "NE*(Q*11*8 + P*7) + 32*N_L (volume elements; the surface blocks are 0.4 % of the faces)"}

    def solve(label, device_gs=True, solver=None):
        sv = S if solver is None else solver
        linalg.Context.set_device_orthogonalization(device_gs)
        try:
            sr, si = torch.zeros_like(br), torch.zeros_like(br)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sv.mult(br, bi, sr, si)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            st = sv.stats()
            out[label] = {"iterations_to_1e-8": st["iterations"], "seconds": dt, "iters_per_s": st["iterations"] / dt, "converged": st["converged"]}
            return sr, si
        finally:
            linalg.Context.set_device_orthogonalization(True)

    sr, si = solve("fgmres")
    out["fgmres"]["orthogonalization"] = "MGS (the reference's default), coefficients on the device: one host synchronisation per column (orthog.hip)"
    A.mult(sr, si, yr, yi)
    out["fgmres"]["true_rel_residual"] = float(torch.sqrt(((yr - br) ** 2 + (yi - bi) ** 2).sum()) / torch.sqrt((br ** 2 + bi ** 2).sum()))
    Sp = ds.s_parameters(sr, si, excited=1)
    ref17 = {1: (-1.810712845683e+01, -1.101540146601e+02), 2: (-6.999449711910e-02, +1.590219070071e+02),
             3: (-5.569003089997e+01, +7.122016852697e+01), 4: (-6.183742235531e+01, -1.302920651568e+02)}
    out["s_parameters"] = {f"S[{j}][1]": {"dB": float(20 * np.log10(abs(v))), "deg": float(np.degrees(np.angle(v)))} for j, v in Sp.items()}
    if abs(freq_ghz - 17.0) < 1e-12:
        out["s_parameters"]["reference_port-S.csv_17GHz_unrefined_p2"] = {f"S[{j}][1]": {"dB": a, "deg": b} for j, (a, b) in ref17.items()}
    # A / B of the orthogonalisation on the same solver object (its basis vectors are allocated now): host-driven MGS, then the
    # device-chained form again; and CGS2 on a second solver
    try:
        s2r, s2i = solve("fgmres_host_driven_mgs", device_gs=False)
        out["fgmres_host_driven_mgs"]["rel_diff_of_the_solution"] = float(torch.sqrt(((s2r - sr) ** 2 + (s2i - si) ** 2).sum()) / torch.sqrt((sr ** 2 + si ** 2).sum()))
        solve("fgmres_second_solve")
        out["fgmres_second_solve"]["note"] = "device-chained MGS, basis vectors already allocated: the like-for-like partner of fgmres_host_driven_mgs"
        S2 = linalg.ComplexParGmres(ctx, A, ds.B, rel_tol=1e-8, max_it=600, restart=600, flexible=True, orthogonalization="CGS2")
        s3r, s3i = solve("fgmres_cgs2", solver=S2)
        out["fgmres_cgs2"]["rel_diff_of_the_solution_from_the_MGS_solve"] = float(torch.sqrt(((s3r - sr) ** 2 + (s3i - si) ** 2).sum()) / torch.sqrt((sr ** 2 + si ** 2).sum()))
        del S2, s2r, s2i, s3r, s3i
    except Exception as exc:  # noqa: BLE001
        out["fgmres_ab_error"] = f"{type(exc).__name__}: {exc}"
    # the complex operator at this size against the oracle: volume operators with the tensor coefficients + the surface mass
    from oracle import palace_oracle as po

    t0 = time.perf_counter()
    nd, k0, v = ds.nd, ds.k0, ds._vol
    interp, curl = nd.elem.tables(prob.pts)
    J = mesh.jacobians(prob.pts)
    og = po.build_geom_factor_33(mesh.attr.astype(np.float64), prob.wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9))
    okw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
    o_r = po.CeedOperatorOracle(n, nd.offsets, None, interp, curl, og, po.QF_HDIVMASS,
                                po.CoeffCtx(attr_mat=v["amap"], mat_coeff=[-k0 ** 2 * m for m in v["eps"]]),
                                po.CoeffCtx(attr_mat=v["amap"], mat_coeff=v["mu_inv"]), **okw)
    o_iv = po.CeedOperatorOracle(n, nd.offsets, None, interp, curl, og, po.QF_HCURL,
                                 po.CoeffCtx(attr_mat=v["amap"], mat_coeff=[k0 ** 2 * m for m in v["eps_tand"]]), **okw)
    sint, scurl = ds.sblk.elem.tables(ds.spts)
    Js = ds.sblk.jacobians(ds.spts)
    ogs = po.build_geom_factor_32(ds.sblk.attr.astype(np.float64), ds.swts, np.transpose(Js, (0, 1, 3, 2)).reshape(ds.sblk.ne, -1, 6))
    o_is = po.CeedOperatorOracle(n, ds.sblk.offsets, ds.sblk.orients, sint, scurl, ogs, po.QF_HCURL_32,
                                 po.CoeffCtx(attr_mat=list(range(len(ds.scoef))), mat_coeff=[k0 * c for c in ds.scoef]))
    ess = ds.ess

    def o_complex(vr, vi):
        mr, mi = vr.copy(), vi.copy()
        mr[ess], mi[ess] = 0.0, 0.0
        z = np.zeros(n)
        ai = lambda w: o_iv.apply_add(w, z.copy()) + o_is.apply_add(w, z.copy())  # noqa: E731
        wr = o_r.apply_add(mr, z.copy()) - ai(mi)
        wi = o_r.apply_add(mi, z.copy()) + ai(mr)
        wr[ess], wi[ess] = vr[ess], vi[ess]
        return wr, wi

    wr, wi = o_complex(xr.cpu().numpy(), xi.cpu().numpy())
    A.mult(xr, xi, yr, yi)
    dd = np.concatenate([yr.cpu().numpy() - wr, yi.cpu().numpy() - wi])
    out["parity"] = {"complex_apply_rel_l2": float(np.linalg.norm(dd) / np.linalg.norm(np.concatenate([wr, wi]))), "tolerance": 1e-12}
    ar, ai_ = o_complex(sr.cpu().numpy(), si.cpu().numpy())
    rr = np.concatenate([ar - br.cpu().numpy(), ai_ - bi.cpu().numpy()])
    out["parity"]["fgmres_solution_rel_residual_in_the_oracle_operator"] = float(np.linalg.norm(rr) / float(torch.sqrt(bi @ bi)))
    out["parity"]["size"] = f"{n} complex dofs ({time.perf_counter() - t0:.1f} s of oracle work: eight volume + eight surface applies of the numpy oracle)"
    return out


def eigen_leg(order=3, dofs=1.0e6, steps=30):
    """BASELINE config 2's shape on the device: the cylinder cavity (radius 2.74 cm, height 5.48 cm, eps_r = 2.08, PEC) at ~1M dofs,
    p = 3, shift-and-invert about the reference's target 2.0 GHz: each outer step is (K - sigma^2 M)^-1 M x by FGMRES + Hiptmair
    p-multigrid + native AMS (positive-shift preconditioner), M-orthogonalisation on the device; the outer iteration is a plain
    Lanczos loop on the host (palace_amd/fem/eigen.py; ARPACK / SLEPc are out of scope).  Reported: inner iterations/s, seconds per
    outer step, the lowest distinct frequencies against the analytic values of docs/src/examples/cylinder.md:113-123."""
    from palace_amd import linalg
    from palace_amd.fem.eigen import HexEigenSystem
    from palace_amd.fem.mesh import cylinder_for_dofs

    t0 = time.perf_counter()
    mesh = cylinder_for_dofs(dofs, order)
    ctx = linalg.Context()
    es = HexEigenSystem(ctx, mesh, order, 2.0, eps_r=2.08, L0=1.0e-2, tol=1.0e-8, max_it=200)
    setup = time.perf_counter() - t0
    res = es.lanczos(steps, nev=4, res_tol=1.0e-8)
    f = [float(v) for v in res["frequencies_ghz"]]
    distinct = []
    for v in f:
        if not distinct or abs(v - distinct[-1]) > 1e-4 * v:
            distinct.append(v)
    analytic = {"TM010": 2.903605, "TE111": 2.922212, "TM011": 3.468149}
    out = {"workload": f"cylinder cavity, {mesh.ne} hex27 elements, ND p={order}, {es.n} dofs, target 2.0 GHz, inner FGMRES to 1e-8 "
                       "(Hiptmair p-multigrid 1..p + native AMS on K + sigma^2 M), divergence-free start vector",
           "dofs": es.n, "setup_s": setup, "outer_steps": res["steps"], "seconds": res["seconds"],
           "seconds_per_outer_step": res["seconds"] / max(1, res["steps"]),
           "inner_iterations": res["inner_iterations"], "inner_iterations_per_solve": res["inner_iterations"] / max(1, res["inner_solves"]),
           "inner_iters_per_s": res["inner_iterations"] / max(1e-9, res["inner_seconds"]),
           "divfree_pcg_iterations": res["divfree_pcg_iterations"],
           "frequencies_ghz": f[:8], "residual_estimates": [float(v) for v in res["residual_estimates"][:8]],
           "lowest_distinct_ghz": distinct[:3],
           "analytic_ghz": analytic,
           "rel_err_vs_analytic": [abs(a - b) / b for a, b in zip(distinct[:3], analytic.values())],
           "rayleigh_quotient_0_rel_diff": abs(res.get("rayleigh_quotient_0", float("nan")) - res["lambda"][0]) / res["lambda"][0]}
    return out


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
    # the curl-curl apply at this size against the numpy oracle (CeedOperatorOracle: restriction with the tridiagonal dof
    # transformation, dense tables, the qfunction, and back), one oracle apply
    from oracle import palace_oracle as po

    t0 = time.perf_counter()
    J = mesh.jacobians(pts)
    og = po.build_geom_factor_33(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9))
    orc = po.CeedOperatorOracle(nd.ndofs, nd.offsets, nd.orients if nd.diagonal_transform else None, interp, curl, og,
                                po.QF_HDIV, po.CoeffCtx(), curl_orients=None if nd.diagonal_transform else nd.curl_orients)
    hx = np.random.default_rng(6).uniform(0, 1, nd.ndofs)
    hy = orc.apply_add(hx, np.zeros(nd.ndofs))
    dy = torch.empty_like(x)
    ops["curlcurl"][0].mult(torch.from_numpy(hx).cuda(), dy)
    out["parity"] = {"rel_l2_y_full": _rel(dy.cpu().numpy(), hy), "tolerance": 1e-12,
                     "size": f"{nd.ndofs} dofs, {mesh.ne} tets ({time.perf_counter() - t0:.1f} s of oracle work)"}
    del J, og, orc, hx, hy, dy
    # complex apply (BASELINE config 3's shape): (K - w^2 eps M) + i w sigma M in one pass (pa_op_mult_complex, dense form)
    from palace_amd import linalg

    cctx = linalg.Context()
    neg = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([-2.08 * 0.3])])
    cond = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([0.05])])
    Ar = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, ceed.QF_HDIVMASS_33, np.concatenate([neg, ident]),
                                                               ceed.EVAL_CURL | ceed.EVAL_INTERP).finalize()
    Ai = ceed.Operator(nd.ndofs, nd.ndofs).add_dense_integrator(geom, block, ceed.QF_HCURL_33, cond, ceed.EVAL_INTERP).finalize()
    Ac = linalg.ComplexParOperator(cctx, Ar, Ai)
    xi, yi = torch.rand_like(x), torch.zeros_like(x)
    for _ in range(5):
        Ac.mult(x, xi, y, yi)
    with torch.cuda.stream(cctx.torch_stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            Ac.mult(x, xi, y, yi)
        e1.record()
        torch.cuda.synchronize()
    cms = e0.elapsed_time(e1) / reps
    calg = Ar.algorithmic_bytes() + 16.0 * nd.ndofs
    out["complex"] = {"one_pass": int(ceed._lib.load().pa_op_complex_fused(Ar.handle, Ai.handle)), "ms": cms,
                      "complex_dof_per_s": nd.ndofs / (cms * 1e-3), "algorithmic_GBps": calg / cms / 1e6,
                      "hbm_frac": calg / cms / 1e6 / HBM_PEAK_GBS,
                      "bytes_formula": "NE*(Q*11*8 + P*7) + 32*N_L: one pass over the element data, both parts of x and y"}
    # at-size parity of the complex apply: four real applies of the numpy oracle (real and imaginary operator on both parts)
    t0 = time.perf_counter()
    J = mesh.jacobians(pts)
    og = po.build_geom_factor_33(mesh.attr.astype(np.float64), wts, np.transpose(J, (0, 1, 3, 2)).reshape(mesh.ne, -1, 9))
    okw = dict(orients=nd.orients) if nd.diagonal_transform else dict(curl_orients=nd.curl_orients)
    o_r = po.CeedOperatorOracle(nd.ndofs, nd.offsets, None, interp, curl, og, po.QF_HDIVMASS,
                                po.CoeffCtx(attr_mat=[0], mat_coeff=[np.array([-2.08 * 0.3])]), po.CoeffCtx(), **okw)
    o_i = po.CeedOperatorOracle(nd.ndofs, nd.offsets, None, interp, curl, og, po.QF_HCURL,
                                po.CoeffCtx(attr_mat=[0], mat_coeff=[np.array([0.05])]), **okw)
    hr, hi = x.cpu().numpy(), xi.cpu().numpy()
    z = np.zeros(nd.ndofs)
    wr = o_r.apply_add(hr, z.copy()) - o_i.apply_add(hi, z.copy())
    wi = o_r.apply_add(hi, z.copy()) + o_i.apply_add(hr, z.copy())
    Ac.mult(x, xi, y, yi)
    torch.cuda.synchronize()
    dd = np.concatenate([y.cpu().numpy() - wr, yi.cpu().numpy() - wi])
    out["complex"]["parity"] = {"rel_l2": float(np.linalg.norm(dd) / np.linalg.norm(np.concatenate([wr, wi]))), "tolerance": 1e-12,
                                "size": f"{nd.ndofs} complex dofs ({time.perf_counter() - t0:.1f} s of oracle work: four real applies)"}
    del Ac, Ar, Ai, J, og, o_r, o_i, wr, wi, dd
    # PCG + p-multigrid (p = 1..order) with the auxiliary-space smoother on the same mesh
    from palace_amd.fem.tetproblem import TetProblem

    for name, coarse in (("pcg_hiptmair", "cg"), ("pcg_hiptmair_ams", "ams")):
        prob = TetProblem(linalg.Context(), mesh, order)
        solver, b, xs = prob.pcg_gmg_solver(max_it=400, rel_tol=1e-8, hiptmair=True, coarse=coarse)
        solver.mult(b, xs)  # warm-up
        xs.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.mult(b, xs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = solver.stats()
        out[name] = {"iterations_to_1e-8": st["iterations"], "seconds": dt, "iters_per_s": st["iterations"] / dt,
                     "converged": st["converged"]}
        del prob, solver
    return out


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


def spheres_leg(orders=(2, 3), reps=50):
    """BASELINE config 4 on the reference's own mesh: examples/spheres/mesh/spheres.msh (14 362 cubic tetrahedra, committed as
    tests/golden/spheres_mesh.npz), electrostatics: H1 order-p diffusion through the dense MFMA path, PCG + p-multigrid
    (levels 1..p, Chebyshev smoothers) with the native algebraic V-cycle on the assembled order-1 level where the reference calls
    BoomerAMG, one solve per terminal, the Maxwell capacitance matrix from the two potentials -- checked in-line against
    test/data/regression/ref/spheres/terminal-C.csv (the reference runs the example at order 3; its own gate is 1e-4)."""
    import torch

    from palace_amd import ceed, linalg
    from palace_amd.fem import tet

    d = np.load(os.path.join(ROOT, "tests", "golden", "spheres_mesh.npz"))
    nodes, en = d["nodes"], d["elem_nodes"].astype(np.int64)
    used, inv = np.unique(en[:, :4], return_inverse=True)
    mesh = tet.TetMesh(nodes[used], inv.reshape(-1, 4), d["attr"])
    bt = np.sort(np.searchsorted(used, d["bdr_tris"].astype(np.int64)), axis=1)
    fkey = {tuple(fv): i for i, fv in enumerate(map(tuple, mesh.face_verts))}
    fm = {}
    for a in (2, 3, 4):  # 2 far field (ground), 3 sphere A, 4 sphere B
        m = np.zeros(mesh.face_verts.shape[0], dtype=bool)
        m[[fkey[tuple(fv)] for fv in bt[d["bdr_attr"] == a]]] = True
        fm[a] = m
    all_m = fm[2] | fm[3] | fm[4]
    ref = d["C_F"]
    eps0 = 1.0 / (1.25663706127e-6 * 299792458.0 ** 2)  # utils/constants.hpp:21-30; the mesh is in cm (L0 = 1e-2)
    out = {"workload": f"examples/spheres mesh: {mesh.ne} cubic tetrahedra, electrostatics (H1 diffusion, three Dirichlet boundaries), "
                       "PCG + p-multigrid + native AMG on level 0, capacitance matrix against ref/spheres/terminal-C.csv",
           "terminal_C_reference_F": ref.tolist()}
    for p in orders:
        levels = list(range(1, p + 1))
        h1s = [tet.H1TetSpace(mesh, q) for q in levels]
        pts, wts = tet.default_tet_rule(p)
        G = tet.H1TetElement(3).tables(pts)[1]  # cubic geometry basis on the fixture's node order
        geom = ceed.DenseGeomFactorData(en, nodes, mesh.attr, G, wts)
        blocks = []
        for sp in h1s:
            interp, grad = sp.elem.tables(pts)
            blocks.append(ceed.DenseBlock(ceed.FE_H1, sp.ndofs, sp.offsets, interp, grad))
        fine = ceed.Operator(h1s[-1].ndofs, h1s[-1].ndofs).add_dense_integrator(geom, blocks[-1], ceed.QF_HCURL_33,
                                                                               ceed.coefficient_context(3), ceed.EVAL_GRAD).finalize()
        local = [fine.coarsen_dense(b) for b in blocks[:-1]] + [fine]
        ess = [sp.ess_dofs(all_m).astype(np.int32) for sp in h1s]
        ctx = linalg.Context()
        A = [linalg.ParOperator(ctx, op, e, linalg.DIAG_ONE) for op, e in zip(local, ess)]
        csr0 = local[0].full_assemble_device()
        A[0] = linalg.AssembledParOperator(ctx, csr0, ess[0], linalg.DIAG_ONE)
        P = [linalg.DenseInterp(ctx, h1s[l].restriction(), h1s[l + 1].restriction(),
                                tet.h1_tet_transfer_matrix(levels[l], levels[l + 1])) for l in range(len(levels) - 1)]
        B = linalg.gmg(ctx, A, P, linalg.amg(ctx, csr0, ess[0]), cheby_order=max(2 * p, 4))
        solver = linalg.cg(ctx, A[-1], B, rel_tol=1e-12, max_it=300)
        n = h1s[-1].ndofs
        x = torch.rand(n, dtype=torch.float64, device="cuda")
        y = torch.empty_like(x)
        with torch.cuda.stream(ctx.torch_stream):
            for _ in range(10):
                fine.mult(x, y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fine.mult(x, y)
            e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        alg = fine.algorithmic_bytes()
        phi, its, secs = [], [], []
        for a in (3, 4):
            v = torch.zeros(n, dtype=torch.float64, device="cuda")
            v[torch.from_numpy(h1s[-1].ess_dofs(fm[a]).astype(np.int64)).cuda()] = 1.0
            b = torch.zeros_like(v)
            A[-1].eliminate_rhs(v, b)
            xs = torch.zeros_like(v)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            solver.mult(b, xs)
            torch.cuda.synchronize()
            secs.append(time.perf_counter() - t0)
            its.append(solver.stats()["iterations"])
            phi.append(xs)
        t = torch.empty(n, dtype=torch.float64, device="cuda")
        Cm = np.zeros((2, 2))
        for i in range(2):
            fine.mult(phi[i], t)
            for j in range(2):
                Cm[j, i] = eps0 * 1.0e-2 * float(phi[j] @ t)
        out[f"p{p}"] = {"dofs": n, "levels": ",".join(str(q) for q in levels),
                        "apply": {"ms": ms, "dof_per_s": n / (ms * 1e-3), "hbm_frac": alg / ms / 1e6 / HBM_PEAK_GBS,
                                  "note": "14 362 elements: a launch of 56 us cannot fill the GPU; the number is here for completeness"},
                        "pcg_iterations_to_1e-12": its, "pcg_seconds": secs, "iters_per_s": sum(its) / sum(secs),
                        "terminal_C_F": Cm.tolist(), "rel_dev_from_terminal_C_csv": float(np.abs(Cm - ref).max() / np.abs(ref).max()),
                        "gate": "order 3 (the order of the reference's regression run): 1e-6; order 2: discretisation difference only"}
        del solver, B, A, P, local, fine, geom
    return out


def hlevels_leg(ctx, prob, order):
    """The reference's FULL hierarchy at the bench size (fem/multigrid.hpp:103-123, utils/geodata.cpp:426-460:
    the meshes of a uniform-refinement sequence are multigrid levels): a cylinder with 1/8 of the bench mesh's elements refined
    once, hierarchy = [order 1 on the coarse mesh] + [orders 1 .. p on the fine mesh], PCG on K + M with the auxiliary-space
    smoothers and the native AMS on the coarsest level -- which is now 8x smaller than with the p-levels alone -- against the
    same fine problem with the p-levels only.  Iterations to 1e-8 and iterations/s."""
    import torch

    from palace_amd.fem.hproblem import HpProblem
    from palace_amd.fem.mesh import ogrid_cylinder

    n, nz = prob.shape
    coarse = ogrid_cylinder(max(1, n // 2), max(1, nz // 2))
    out = {}
    hp = HpProblem(ctx, coarse, 1, order)
    for name, pr in (("h_and_p_levels", hp), ("p_levels_only", None)):
        if pr is None:
            pr = HpProblem(ctx, hp.meshes[-1], 0, order)
        K, b, x = pr.pcg_gmg_solver(max_it=400, rel_tol=1e-8, hiptmair=True, coarse="ams")
        K.mult(b, x)  # (first solve: work vectors, graph recording)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K.mult(b, x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = K.stats()
        out[name] = {"levels": [f"mesh {m} ({pr.meshes[m].ne} elements), order {q}" for m, q in pr.levels],
                     "dofs_per_level": [s.ndofs for s in pr.spaces], "iterations_to_1e-8": st["iterations"], "converged": bool(st["converged"]),
                     "seconds": dt, "iters_per_s": st["iterations"] / dt}
        if name == "h_and_p_levels":
            xs = x.clone()
        else:
            out["rel_diff_of_the_two_solutions"] = float((x - xs).norm() / x.norm())
        pr._keep.clear()
    out["workload"] = (f"PCG on K + M (eps_r = 2.08), ND p={order}, {hp.spaces[-1].ndofs} dofs on {hp.meshes[-1].ne} hex27 elements (a once-refined "
                       f"{hp.meshes[0].ne}-element cylinder), Hiptmair smoothers, AMS on the coarsest level")
    return out


def magnetostatic_leg(ctx, prob, iters=400):
    """The singular magnetostatic system on the bench cylinder: curl-curl alone (no mass term), PCG + p-multigrid with plain
    Chebyshev smoothers (the reference's configuration for magnetostatics, iodata.cpp:533-564) and the native AMS on level 0 in
    its singular mode (ams_singular_op: no gradient-space correction, linalg/ams.cpp:28-30, :149-152); the right-hand side is in
    the range of K (K times a random vector), iterations to 1e-8 in the preconditioned residual."""
    import torch

    solver, b, xs = prob.pcg_gmg_solver(max_it=iters, rel_tol=1e-8, hiptmair=False, coarse="ams", eps_r=0.0, singular=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.mult(b, xs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = solver.stats()
    prob._keep.clear()
    out = {"workload": f"K x = b (curl-curl only, singular), ND p={prob.p}, {b.numel()} dofs, b = K (random)", "iterations_to_1e-8": st["iterations"],
           "seconds": dt, "iters_per_s": st["iterations"] / dt, "converged": st["converged"],
           "final_rel_res": st["final_res"] / st["initial_res"]}
    # the reference's own magnetostatic case (examples/cavity2d/cavity2d_magnetostatic.json) through the same device solver
    # stack, against its regression value (test/data/regression/ref/cavity2d/magnetostatic/terminal-M.csv)
    try:
        from palace_amd.fem import triproblem

        mesh, bv, battr, M_ = triproblem.load_cavity2d(os.path.join(ROOT, "tests", "golden", "cavity2d_mesh.npz"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = triproblem.magnetostatic_inductance(ctx, mesh, bv, battr, 2, [1.0, 0.0], order=2, rel_tol=1e-8, max_it=100)
        torch.cuda.synchronize()
        ref = float(M_["M11_H"])
        out["cavity2d"] = {"case": "examples/cavity2d/cavity2d_magnetostatic.json: order 2, 2-D curl-curl (dense MFMA path), PCG + p-multigrid + "
                                   "singular AMS on the device", "dofs": r["ndofs"], "iterations_to_1e-8": r["iterations"], "converged": r["converged"],
                           "M11_H": r["M11"], "terminal_M_csv_H": ref, "rel_dev_from_terminal_M_csv": abs(r["M11"] - ref) / ref,
                           "seconds_setup_and_solve": time.perf_counter() - t0}
    except Exception as exc:  # noqa: BLE001
        out["cavity2d"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def measure_traffic(dofs, timeout=240):
    """HBM-side traffic of ONE curl-curl apply (element kernel + run gather), measured in this run: two `rocprofv3 --pmc` child
    processes (FETCH_SIZE, WRITE_SIZE: separate passes, counters only with --kernel-trace, as MI355X_MICROARCH.md prescribes)
    over scripts/profile_apply.py -- the bench mesh, 10 applies, then a calibration stream y = a x + b y with known bytes in the
    same process -- each counter divided by the fraction it reports of that known stream (the gfx950 FETCH_SIZE halving of
    16-byte-lane loads included).  Returns (bytes per apply or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals, n_cal = {}, None
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pa_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT, OP="curl", REPS="10", DOFS=str(dofs))
        try:
            p = subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.join(ROOT, "scripts", "profile_apply.py")], cwd="/tmp", env=env, capture_output=True, timeout=timeout)
            for ln in p.stdout.decode(errors="replace").splitlines():
                if ln.startswith("done"):
                    n_cal = int(ln.split()[1])
            acc = {}
            for fcsv in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(fcsv)):
                    k = row["Kernel_Name"]
                    tag = ("elem" if "nd_hex_stream_kernel" in k else "gather" if "et_run_gather_kernel" in k else
                           "cal" if ("OpAxpby" in k and ("k_ew<2" in k or "k_ew<(int)2" in k)) else None)
                    if tag and row["Counter_Name"] == ctr:
                        sm, ids = acc.get(tag, (0.0, set()))
                        ids.add(row["Dispatch_Id"])
                        acc[tag] = (sm + float(row["Counter_Value"]), ids)
            for tag, (sm, ids) in acc.items():
                vals[(tag, ctr)] = sm / max(1, len(ids)) * 1024.0  # (KiB per dispatch)
        except Exception as exc:  # noqa: BLE001
            return None, f"rocprofv3 --pmc {ctr} failed: {type(exc).__name__}: {exc}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    need = [("elem", "FETCH_SIZE"), ("gather", "FETCH_SIZE"), ("cal", "FETCH_SIZE"), ("elem", "WRITE_SIZE"), ("gather", "WRITE_SIZE"),
            ("cal", "WRITE_SIZE")]
    if n_cal is None or any(k not in vals for k in need):
        return None, "counter output incomplete: " + ", ".join(f"{a}.{b}" for a, b in need if (a, b) not in vals)
    rf = vals[("cal", "FETCH_SIZE")] / (16.0 * n_cal)
    rw = vals[("cal", "WRITE_SIZE")] / (8.0 * n_cal)
    traffic = (vals[("elem", "FETCH_SIZE")] + vals[("gather", "FETCH_SIZE")]) / rf + (vals[("elem", "WRITE_SIZE")] + vals[("gather", "WRITE_SIZE")]) / rw
    note = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes) over 10 applies on the bench mesh; per apply: "
            f"element kernel {vals[('elem', 'FETCH_SIZE')] / 1e6:.1f} MB fetched (raw) + {vals[('elem', 'WRITE_SIZE')] / 1e6:.1f} MB written, run gather "
            f"{vals[('gather', 'FETCH_SIZE')] / 1e6:.1f} + {vals[('gather', 'WRITE_SIZE')] / 1e6:.1f}; calibration on y = a x + b y over {n_cal} doubles "
            f"in the same process: FETCH_SIZE reports {rf:.3f} of the known read bytes, WRITE_SIZE {rw:.3f} of the written ones; raw counters divided by those")
    return traffic, note


def main():
    args = parse()
    if args.rehearse > 1:
        return rehearse(args)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL across processes)
    # stdout carries the one JSON line and nothing else: whatever libraries print there (RCCL's version banner at communicator
    # creation) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    # rehearsal (bench.py --rehearse N): every rank on device 0, gloo instead of RCCL -- everything below is the same code
    rehearsal = os.environ.get("PALACE_AMD_BENCH_REHEARSE") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    cdev = "cpu" if rehearsal else "cuda"  # where the tensors of torch.distributed's own collectives live

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
        if rehearsal:
            ctx.init_comm_peer_from_torch_distributed()
        else:
            ctx.init_comm_from_torch_distributed()
    elif args.force_comm:
        ctx.init_comm_single()

    # ---- set-up (not timed): mesh slab, spaces, geometry factors, operators ---------------------
    t_setup = time.perf_counter()
    if args.scaling == "strong":
        # one global mesh for every N: the layer count is rounded to a multiple of 8 so that 1, 2, 4 and 8 ranks cut
        # the very same cylinder into equal z-slabs
        from palace_amd.fem.partition import strong_shape
        n_cross, nz = strong_shape(args.dofs, p)
        if nz % world:
            raise SystemExit(f"--scaling strong needs a rank count dividing {nz} layers")
        prob = SlabProblem(ctx, rank, world, p, args.dofs, levels=True, shape=(n_cross, nz // world))
    else:
        prob = SlabProblem(ctx, rank, world, p, args.dofs, levels=True)
    K = prob.curlcurl_par_operator()  # ParOperator(curl-curl, mu^-1 = 1), PEC essential dofs, DIAG_ONE
    n_true = prob.n_true[-1]
    n_global = prob.global_true_dofs()
    x = torch.empty(n_true, dtype=torch.float64, device="cuda")
    y = torch.empty(n_true, dtype=torch.float64, device="cuda")
    ctx.set_random(x, 1 + rank)
    x.add_(1.0).mul_(0.5)  # uniform [0, 1)
    x[torch.from_numpy(prob.ess[-1].astype(np.int64)).cuda()] = 0.0
    # multi-rank: which halo transport runs, and a cross-check of the two forms of the peer transport on the bench operator itself
    # (direct form: the element kernel reads the neighbours' stores from the mailbox with plain loads; L-vector form: they are
    # copied out with system-scope loads first -- the form the start-up self-test of the transport exercises).  A mismatch
    # keeps the L-vector form for everything that follows and is reported in the line.
    halo_info = None
    if world > 1:
        halo_info = {"transport": "peer (direct stores over IPC-mapped arenas)" if ctx.peer_ready() else "rccl send / receive groups",
                     "bring_up": getattr(ctx, "transport_report", None), "direct_form": K.direct_form(),
                     "partition": partition_report(prob.spaces[-1], f"ND p={p}, z-slabs")}
        df = torch.tensor([K.direct_form()], dtype=torch.int64, device=cdev)
        dist.all_reduce(df, op=dist.ReduceOp.MIN)  # (the check below is collective: every rank or none)
        if int(df.item()) == 1:
            y2 = torch.empty_like(y)
            K.mult(x, y)
            K.set_direct(False)
            K.mult(x, y2)
            err = torch.tensor([float((y - y2).abs().max()), float(y2.abs().max())], dtype=torch.float64, device=cdev)
            dist.all_reduce(err, op=dist.ReduceOp.MAX)
            rel = float(err[0] / err[1]) if float(err[1]) > 0 else float("inf")
            halo_info["direct_vs_lvector_rel_err"] = rel
            if rel < 1e-13:
                K.set_direct(True)
            else:
                os.environ["PALACE_AMD_HALO_DIRECT"] = "0"  # (read once per process: before any other ParOperator is made)
                halo_info["direct_form"] = 0
            del y2
    for _ in range(args.pre_warm):  # bring the clocks to their steady state (part of the set-up, not of the measurement)
        K.mult(x, y)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- M1: ParOperator::Mult throughput --------------------------------------------------------
    # W untimed warm-up steps, then blocks of EXACTLY K steps between barrier + synchronize; a K-step block shorter than
    # 100 ms (the driver's --steps 20 is 3 ms of device work) is repeated back to back inside one bracket until the timed
    # window is >= 100 ms, so that `value` is not a statement about one clock state (round-4 review): value = dofs x
    # (blocks x K) / elapsed, ms_per_step = elapsed / (blocks x K); `timed_blocks` is in the line.
    for _ in range(args.warmup):
        K.mult(x, y)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        K.mult(x, y)
    barrier()
    probe = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([probe], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        probe = float(t.item())
    blocks = 1 if probe >= 0.1 else int(min(10000, np.ceil(0.1 / max(probe, 1e-6))))
    barrier()
    t0 = time.perf_counter()
    for _ in range(blocks):
        for _ in range(args.steps):
            K.mult(x, y)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / (args.steps * blocks)
    value = n_global * args.steps * blocks / elapsed
    first_block_ms_per_step = 1e3 * probe / args.steps

    # ---- roofline of the dominant kernel (fused apply), HIP events on the launch stream -----------
    local_op = prob.local_curlcurl
    lx = torch.zeros(prob.n_local[-1], dtype=torch.float64, device="cuda")
    ly = torch.zeros(prob.n_local[-1], dtype=torch.float64, device="cuda")
    lx[:n_true] = x
    for _ in range(3):
        local_op.mult(lx, ly)
    # events on the stream the kernels are launched on (the context's own stream; local_op.mult above goes to PyTorch's
    # current stream): one ParOperator::Mult = the element kernel + the E^T run gather with the essential rows fused.
    # The leg has its own warm-up and repetition count, independent of --steps: the few applies of a short driver run,
    # timed right after a synchronisation and host work, measure the clock ramp, not the kernel (round-2 review).
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nk = max(500, min(args.steps, 1000))  # (the first ~50 ms of kernel time after host work run ~5 % slow: long warm-up, >= 500 timed applies)

    def _roofline_apply():
        if world == 1:
            K.mult(x, y)
        else:  # (multi-rank: the local operator without the halo exchange)
            local_op.mult(lx, ly)

    with torch.cuda.stream(ctx.torch_stream if world == 1 else torch.cuda.current_stream()):
        for _ in range(300):
            _roofline_apply()
        ev0.record()
        for _ in range(nk):
            _roofline_apply()
        ev1.record()
    torch.cuda.synchronize()
    kernel_ms = ev0.elapsed_time(ev1) / nk
    alg_bytes = local_op.algorithmic_bytes()
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    # HBM traffic of the same launch from the PMC passes (collected by scripts/profile_round.sh in separate
    # rocprofv3 --pmc runs, summary committed under profiles/): raw FETCH_SIZE + WRITE_SIZE bytes
    traffic, traffic_note = None, "no PMC summary under profiles/"
    pmc_file = os.path.join(ROOT, "profiles", "r04_apply_pmc.json")  # (fall-back when the in-run measurement below is not available)
    if not os.path.exists(pmc_file):
        pmc_file = os.path.join(ROOT, "profiles", "r03_apply_pmc.json")
    if os.path.exists(pmc_file) and abs(args.dofs - 10.0e6) < 1 and p == 3 and world == 1 and args.scaling == "strong":
        pmc = json.load(open(pmc_file))
        pb = pmc["per_apply_bytes"]
        traffic = pb.get("traffic_corrected") or pb["traffic_raw"]
        traffic_note = ("FETCH_SIZE + WRITE_SIZE per apply from profiles/" + os.path.basename(pmc_file) + " (separate --pmc passes of the "
                        "same kernels on this mesh; collected at commit " + str(pmc.get("commit", "?")) + "), each divided by "
                        "the fraction the same counters report on a known stream in the same run; "
                        + pmc.get("calibration", "uncalibrated"))
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
    design_floor = prob.mesh.ne * ((p + 1) ** 3 * 6 * 8 + 384) + 16 * prob.n_local[-1] if p == 3 else None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "measured_stream_GBps": measured_stream, "frac_of_measured_stream": achieved / measured_stream,
                "measured_mfma_f64_TFLOPs": measured_mfma,
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                "kernel": "pa::nd_hex_stream_kernel<P1=3, packed q-data> (E, B, D, B^T, signed E-vector / exclusive dofs) + "
                          "pa::et_run_gather_kernel (E^T over shared-dof runs)", "kernel_ms": kernel_ms,
                "kernel_reps": nk, "kernel_warmup": 300,
                # the two clocks of this line must tell the same story (round-2 review): events around nk applies on the launch
                # stream against the wall clock around --steps applies
                "consistent_with_ms_per_step": bool(world > 1 or kernel_ms <= 1.05 * ms_per_step),
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_formula": "NE*(Q*11*8 + P*5) + 16*N_L (SURVEY.md 8d, G=11)",
                # what this design has to move at least: packed q-data 6 doubles / point + 384 B of index per element + one pass
                # over x and y (the E-vector round trip of the shared dofs comes on top)
                "design_floor_bytes": design_floor,
                "traffic_over_design_floor": (traffic / design_floor) if (traffic and design_floor) else None}
    # The "second line" of SURVEY.md 8(d): the same apply with the geometry recomputed from the 27 nodes of every element (648 B per
    # element instead of 3 072 B of packed D; PALACE_AMD_STREAM_GEOM=nodes, read at every launch).  It moves fewer bytes and is
    # SLOWER on this chip (the 27 partial sums per lane cost a wave per SIMD): reported, not used for `value`.
    if world == 1 and p == 3:
        try:
            K.mult(x, y)
            y_packed = y.clone()
            os.environ["PALACE_AMD_STREAM_GEOM"] = "nodes"
            K.mult(x, y)
            reld = float((y - y_packed).norm() / y_packed.norm())
            gms = _event_ms(lambda: [K.mult(x, y) for _ in range(50)], 6) / 50
            roofline["geometry_from_nodes"] = {"ms": gms, "dof_per_s": n_global / (gms * 1e-3), "rel_diff_from_the_packed_form": reld,
                                               "bytes_per_element": "27 x 3 x 8 = 648 (nodes) instead of 64 x 6 x 8 = 3 072 (packed D)",
                                               "used_for_value": False}
            del y_packed
        except Exception as exc:  # noqa: BLE001
            roofline["geometry_from_nodes"] = {"error": f"{type(exc).__name__}: {exc}"}
        finally:
            os.environ.pop("PALACE_AMD_STREAM_GEOM", None)
    try:  # affine batches of the streaming kernel (round 6): how many elements read the compact per-element D
        ne_, na_, nc_ = prob.local_curlcurl.stream_affine()
        roofline["affine_elements"] = {"elements": ne_, "affine": na_, "compressed_in_batches_of_4": nc_, "fraction_compressed": nc_ / max(1, ne_),
                                       "note": "elements with a constant Jacobian (the central block of the O-grid) read 6 numbers per element "
                                               "instead of per point; every other element is unchanged (PALACE_AMD_STREAM_AFFINE=0 switches the form off)"}
    except Exception as exc:  # noqa: BLE001
        roofline["affine_elements"] = {"error": str(exc)}
    if world == 1 and not roofline["consistent_with_ms_per_step"]:
        print(f"bench.py: roofline leg {kernel_ms:.4f} ms per apply against {ms_per_step:.4f} ms per step", file=sys.stderr)

    # ---- M2: PCG + p-multigrid on (K + M) x = b ---------------------------------------------------
    # iterations/s over a fixed number of iterations (rel_tol = 0, so the count is the same on any
    # device), for the two smoother configurations the reference uses, plus iterations-to-1e-8.
    pcg = None
    if args.pcg_iters > 0:
        pcg = {}
        # level 0: the stand-ins of rounds 1-2 (comparable numbers), and the native auxiliary-space solver (AMS) where the
        # reference calls HYPRE's (one rank: it works on the rank's own matrix)
        legs = [("chebyshev", False, "chebyshev"), ("hiptmair", True, "cg")]
        if world == 1:
            legs += [("chebyshev_ams", False, "ams"), ("hiptmair_ams", True, "ams")]
        else:  # several ranks: level 0 solved redundantly by every rank (ReplicatedSolver around the native AMS)
            legs += [("hiptmair_ams", True, "ams")]
            # round 5: the same cycle with its SOLVE distributed (amg_dist.hpp: every rank its rows of every algebraic level;
            # assembled by the C++ layer from the ranks' own pieces)
            legs += [("hiptmair_ams_distributed", True, "ams_dist")]
        for name, hip, coarse in legs:
            try:  # a failing secondary leg is reported in the line, it does not take the headline measurement with it
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
                if not hip:  # (plain Chebyshev levels: are their steps evaluated inside the operator's E^T gather?)
                    try:
                        entry["chebyshev_steps_fused_into_the_gather"] = [bool(prob.last_gmg.fused_step(l)) for l in range(1, len(prob.spaces))]
                    except Exception:  # noqa: BLE001
                        pass
                solver, b, xs = prob.pcg_gmg_solver(max_it=400, rel_tol=1e-8, hiptmair=hip, coarse=coarse)
                barrier()
                t0 = time.perf_counter()
                solver.mult(b, xs)
                barrier()
                st = solver.stats()
                entry.update({"iterations_to_1e-8": st["iterations"], "seconds_to_1e-8": time.perf_counter() - t0,
                              "converged": st["converged"]})
                if world > 1 and coarse in ("ams", "ams_dist"):
                    # what the replicated level-0 solve costs per application (gather over the transport + the AMS cycle of
                    # the whole cylinder's order-1 problem on every rank) against one PCG iteration: the part that does not scale
                    cs, n0 = prob.last_coarse, prob.n_true[0]
                    r0 = torch.rand(n0, dtype=torch.float64, device="cuda")
                    z0 = torch.zeros_like(r0)
                    for _ in range(3):
                        cs.mult(r0, z0)
                    barrier()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        cs.mult(r0, z0)
                    barrier()
                    ms0 = 1e3 * (time.perf_counter() - t0) / 20
                    entry["replicated_level0" if coarse == "ams" else "distributed_level0"] = {
                        "ms_per_application": ms0, "share_of_iteration": ms0 * 1e-3 * entry["iters_per_s"]}
                    if coarse == "ams_dist":  # (what the C++ layer built: amg_dist.hpp unless PALACE_AMD_COARSE_SOLVE=replicated)
                        entry["distributed_level0"].update({"distributed": bool(getattr(cs, "distributed", False)),
                                                            "algebraic_levels": int(getattr(cs, "algebraic_levels", 0))})
                pcg[name] = entry
                prob._keep.clear()
            except Exception as exc:  # noqa: BLE001
                pcg[name] = {"error": f"{type(exc).__name__}: {exc}"}
                prob._keep.clear()
        pcg["config"] = (f"PCG on K+M (eps_r=2.08), p-multigrid levels p={','.join(str(q) for q in prob.orders)}, "
                         f"4th-kind Chebyshev order {max(2 * p, 4)}, 1 V-cycle "
                         "per iteration; 'chebyshev' = plain smoother (reference default for magnetostatics), "
                         "'hiptmair' = auxiliary-space smoother (reference default for driven/eigenmode); level 0 assembled to a device CSR matrix like the reference's coarsest level (stand-in for AMS on it): "
                         "Chebyshev-Jacobi order 4 with the plain smoother, 8 Jacobi-PCG iterations with the auxiliary-space one; '*_ams' = the "
                         "native auxiliary-space solver on level 0 (amg_solver.hip: smoothed-aggregation AMG on G^T A G and on the "
                         "three Pi_c^T A Pi_c, HYPRE AMS cycle 14, where the reference calls HYPRE's AMS)")

    tets = None
    def _leg(fn, *a):
        try:
            return fn(*a)
        except Exception as exc:  # noqa: BLE001 -- reported in the line
            return {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0 and world == 1 and not args.no_tets:
        tets = _leg(tets_leg, p, args.tet_n)
        for key in ("curlcurl", "curlcurl_mass"):
            if tets and key in tets:
                tets[key]["frac_of_measured_mfma_f64"] = tets[key]["table_TFLOPs"] / measured_mfma
                tets[key]["frac_of_measured_stream"] = tets[key]["algorithmic_GBps"] / measured_stream

    p4 = None
    if rank == 0 and world == 1 and not args.no_p4:
        p4 = _leg(p4_leg, ctx, args.dofs)
    cplx = cplx_aniso = h1 = None
    if rank == 0 and world == 1 and not args.no_p4:
        cplx = _leg(complex_leg, ctx, prob)
        cplx_aniso = _leg(complex_leg, ctx, prob, 50, True, True)
        h1 = _leg(h1_leg, ctx, prob)
    eig = None
    if rank == 0 and world == 1 and not args.no_p4:
        eig = _leg(eigen_leg, p)
    cpw = sph = mag = None
    cpw_iso = None
    if rank == 0 and world == 1 and not args.no_tets:
        cpw = _leg(cpw_leg, p)
        cpw_iso = _leg(cpw_iso_leg, p)
        sph = _leg(spheres_leg)
    hlev = None
    if rank == 0 and world == 1 and not args.no_p4:
        mag = _leg(magnetostatic_leg, ctx, prob)
        hlev = _leg(hlevels_leg, ctx, prob, p)

    nranks = None
    if world > 1 and not args.no_nranks_legs:
        def max_over_ranks(v):
            t = torch.tensor([v], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        nranks = nranks_legs(ctx, rank, world, args, barrier, max_over_ranks)
        try:  # whatever the transport noted while the legs ran
            ctx.peer_check()
            nranks["peer_check"] = "ok"
        except Exception as exc:  # noqa: BLE001
            nranks["peer_check"] = str(exc)

    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            cpu, parity = cpu_leg(ctx, prob, p, args)
        except Exception as exc:  # noqa: BLE001
            cpu, parity = {"error": f"{type(exc).__name__}: {exc}"}, None
    if rank == 0 and world == 1 and not args.no_traffic and abs(args.dofs - 10.0e6) < 1 and p == 3 and args.scaling == "strong":
        tr, note = measure_traffic(args.dofs)
        if tr is not None:
            roofline["traffic"], roofline["traffic_note"] = tr, note
            roofline["traffic_over_design_floor"] = (tr / design_floor) if design_floor else None
            roofline["traffic_over_algorithmic"] = tr / alg_bytes
        else:
            roofline["traffic_note"] = f"in-run measurement unavailable ({note}); " + roofline["traffic_note"]
    if world > 1:
        dist.barrier()

    if rank == 0:
        out = {
            "metric": "curl-curl Mult DOF/s + PCG iters/s, p=3 H(curl) 10M DOF @1/2/4/8 GPU",
            "value": value, "unit": "DOF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "timed_blocks": blocks, "first_block_ms_per_step": first_block_ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"ND p={p} curl-curl ParOperator::Mult, O-grid cylinder cavity (hex27, PEC), "
                                   f"{n_global} true dofs total", "order": p, "elements_per_gpu": prob.mesh.ne,
                       "true_dofs_per_gpu": n_true, "global_true_dofs": n_global, "q1d": p + 1,
                       "scaling_mode": ("strong: one ~10M-dof cylinder cut into N equal z-slabs" if args.scaling == "strong"
                                        else "weak: one z-slab of the cylinder per GPU, same element count per GPU"),
                       "parallelism": f"element partition x{world}, halo (P / P^T) and global sums over "
                                      + ("the peer transport (direct xGMI stores; RCCL as the fall-back)" if (world > 1 and ctx.peer_ready())
                                         else "RCCL")},
            "rehearsal": rehearsal, "pre_warm_steps": args.pre_warm, "halo": halo_info, "n_ranks_legs": nranks, "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "pcg": pcg, "p4": p4, "complex": cplx, "complex_aniso": cplx_aniso, "h1": h1, "eigenmode": eig, "cpw": cpw, "cpw_iso": cpw_iso, "spheres": sph, "magnetostatic": mag, "h_levels": hlev, "tets_mfma": tets,
            "setup_s": t_setup,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
