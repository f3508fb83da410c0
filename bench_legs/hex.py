"""Secondary legs on the hexahedral bench cylinder: order 4, the one-pass complex applies, H1 (config 4 shape), the h-level hierarchy, magnetostatics.  Part of bench.py (split in round 6; `python bench.py` is the entry point)."""
import json  # noqa: F401
import os  # noqa: F401
import sys  # noqa: F401
import time  # noqa: F401

import numpy as np  # noqa: F401

from .common import HBM_PEAK_GBS, ROOT, _rel, host_cores, oracle_hex_data  # noqa: F401


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
