"""A / B of the affine-batch form of the streaming H(curl) kernel (round 6) on the bench mesh: the same operators built with
PALACE_AMD_STREAM_AFFINE=0 and 1 in one process, alternating, HIP-event timed (ParOperator curl-curl = the headline step, K + M
on the three p-levels = what the smoothers of the PCG loop apply)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palace_amd import ceed, linalg
from palace_amd.fem.partition import SlabProblem

ctx = linalg.Context()
dofs = float(os.environ.get("DOFS", "10e6"))
out = {}
probs = {}
for aff in ("0", "1"):
    os.environ["PALACE_AMD_STREAM_AFFINE"] = aff
    prob = SlabProblem(ctx, 0, 1, 3, dofs, levels=True)
    mass = ceed.coefficient_context(3, attr_mat=[0], mat_coeff=[np.array([2.08])])
    fine = ceed.curlcurlmass_operator(prob.geom, prob.spaces[-1], mass, ceed.coefficient_context(3))
    ops = {"K(par)": prob.curlcurl_par_operator(), "K+M p3": fine}
    for s in prob.spaces[:-1]:
        ops[f"K+M p{s.p}"] = fine.coarsen(prob.geom, s)
    probs[aff] = (prob, ops, fine)
    print("affine", aff, "stream_affine", fine.stream_affine(), flush=True)
def tm(op, n, reps=50):
    x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.zeros(n, dtype=torch.float64, device="cuda")
    for _ in range(5): op.mult(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): op.mult(x, y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, y
for rnd in range(3):
    for name in probs["0"][1]:
        row = []
        ys = []
        for aff in ("0", "1"):
            prob, ops, _ = probs[aff]
            op = ops[name]
            n = prob.spaces[-1].ndofs if name in ("K(par)", "K+M p3") else [s for s in prob.spaces if f"p{s.p}" in name][0].ndofs
            us, y = tm(op, n)
            row.append(us); ys.append(y)
        print(f"round {rnd} {name:8s} affine off {row[0]:7.1f} us   on {row[1]:7.1f} us   ({100 * (row[1] / row[0] - 1):+.1f} %)", flush=True)
