"""CPU baseline and parity leg (rank 0, N = 1): the oracle timed on the host cores and the device results against it.  Part of bench.py (split in round 6; `python bench.py` is the entry point)."""
import json  # noqa: F401
import os  # noqa: F401
import sys  # noqa: F401
import time  # noqa: F401

import numpy as np  # noqa: F401

from .common import HBM_PEAK_GBS, ROOT, _rel, host_cores, oracle_hex_data  # noqa: F401


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
