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

from bench_legs.common import HBM_PEAK_GBS, _rel, host_cores, oracle_hex_data  # noqa: E402,F401
from bench_legs.cpu import cpu_leg  # noqa: E402,F401
from bench_legs.eigen import eigen_leg  # noqa: E402,F401
from bench_legs.hex import complex_leg, h1_leg, hlevels_leg, magnetostatic_leg, p4_leg  # noqa: E402,F401
from bench_legs.ranks import nranks_legs, partition_report  # noqa: E402,F401
from bench_legs.tets import cpw_iso_leg, cpw_leg, spheres_leg, tets_leg  # noqa: E402,F401
from bench_legs.traffic import measure_traffic  # noqa: E402,F401


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
    pmc_file = os.path.join(ROOT, "profiles", "r06_apply_pmc.json")  # (fall-back when the in-run measurement below is not available)
    if not os.path.exists(pmc_file):
        pmc_file = os.path.join(ROOT, "profiles", "r05_apply_pmc.json")
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
                          "pa::et_run_gather_kernel_t<false, false> (E^T over shared-dof runs)", "kernel_ms": kernel_ms,
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
