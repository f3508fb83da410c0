"""A / B of the Chebyshev step fused into the E^T run gather (round 6) on the bench problem: PCG + p-multigrid (plain and
auxiliary-space smoothers), the same solve with PALACE_AMD_FUSED_STEP=0 and 1, alternating, fixed iteration count."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from palace_amd import linalg
from palace_amd.fem.partition import SlabProblem

ctx = linalg.Context()
prob = SlabProblem(ctx, 0, 1, 3, float(os.environ.get("DOFS", "10e6")), levels=True)
its = int(os.environ.get("ITS", "50"))
for hip in (False, True):
    res = {}
    for rnd in range(2):
        for f in ("0", "1"):
            os.environ["PALACE_AMD_FUSED_STEP"] = f
            K, b, x = prob.pcg_gmg_solver(max_it=its, hiptmair=hip, coarse="chebyshev" if not hip else "cg")
            fused = [prob.last_gmg.fused_step(l) for l in range(1, len(prob.spaces))] if not hip else None
            K.mult(b, x); torch.cuda.synchronize()
            t0 = time.perf_counter(); K.mult(b, x); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            st = K.stats()
            res.setdefault(f, []).append(st["iterations"] / dt)
            print(f"hiptmair={hip} fused={f} {fused} round {rnd}: {st['iterations'] / dt:7.1f} it/s  final rel res {st['final_res'] / st['initial_res']:.6e}", flush=True)
            prob._keep.clear()
    print(f"hiptmair={hip}: fused off {max(res['0']):.1f} it/s, on {max(res['1']):.1f} it/s ({100 * (max(res['1']) / max(res['0']) - 1):+.1f} %)")
